"""Fingerprint of what EXECUTES: sha256 (16 hex digits) over the `.hip_fatbin` section of libcenternet_uda_hip.so --
the gfx950 code objects hipcc embedded -- so that a committed counter profile can say whether it was collected from the
kernels a bench run executes.  (Round 3 hashed the raw source text: a comment-only commit after the profile collection
made the driver's bench line disown its own `roofline.traffic`.)  Comments and whitespace do not reach the code object;
the library is built without -g and its device code uses neither __LINE__ nor assert().

    python3 profiles/fingerprint.py        ->  prints the fingerprint of the in-tree library
"""
import hashlib
import os
import struct

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'centernet-uda_amd', 'libcenternet_uda_hip.so')


def _elf_sections(blob):
    assert blob[:4] == b'\x7fELF' and blob[4] == 2 and blob[5] == 1, 'not a little-endian ELF64 file'
    shoff, = struct.unpack_from('<Q', blob, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from('<HHH', blob, 0x3a)
    heads = [struct.unpack_from('<IIQQQQIIQQ', blob, shoff + i * shentsize) for i in range(shnum)]
    stroff = heads[shstrndx][4]
    for name_off, _type, _flags, _addr, off, size, *_ in heads:
        end = blob.index(b'\0', stroff + name_off)
        yield blob[stroff + name_off:end].decode(), off, size


def code_fingerprint(path=LIB):
    blob = open(path, 'rb').read()
    for name, off, size in _elf_sections(blob):
        if name == '.hip_fatbin':
            return hashlib.sha256(blob[off:off + size]).hexdigest()[:16]
    raise RuntimeError('%s has no .hip_fatbin section' % path)


if __name__ == '__main__':
    print(code_fingerprint())
