"""Register / LDS budget of every kernel of one source file as hipcc compiles it for gfx950 (code-object metadata):
    python3 profiles/kernel_resources.py centernet-uda_amd/csrc/conv.hip [substring ...] [-DFLAG ...]
-> agpr, vgpr, LDS bytes, spills, waves per SIMD the unified 512-register file allows (arch + acc, 8-register granules)."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(src, flags=()):
    with tempfile.TemporaryDirectory() as tmp:
        co, elf = os.path.join(tmp, 'a.co'), os.path.join(tmp, 'a.elf')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off',
                               '--cuda-device-only', '-c', src, '-o', co, '-I' + os.path.join(ROOT, 'include')] + list(flags))
        subprocess.check_call([LLVM + '/clang-offload-bundler', '--unbundle', '--type=o', '--input=' + co,
                               '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + elf])
        notes = subprocess.run([LLVM + '/llvm-readelf', '--notes', elf], capture_output=True, text=True).stdout
    rows = []
    for blk in notes.split('- .agpr_count:')[1:]:
        f = dict(re.findall(r'\.(\w+):\s+(\S+)', '.agpr_count:' + blk))
        name = subprocess.run(['c++filt', f['name']], capture_output=True, text=True).stdout.strip()
        name = name.replace('cnuda::(anonymous namespace)::', '').replace('void ', '')
        depth, cut = 0, len(name)
        for i, ch in enumerate(name):
            depth += ch == '<'
            depth -= ch == '>'
            if ch == '(' and depth == 0:
                cut = i
                break
        a, v = int(f['agpr_count']), int(f['vgpr_count'])
        total = (v + 7) // 8 * 8 + (a + 7) // 8 * 8 if a else (v + 7) // 8 * 8
        rows.append((name[:cut], a, v, int(f['group_segment_fixed_size']), int(f.get('vgpr_spill_count', 0)),
                     min(8, 512 // max(total, 1))))
    return sorted(rows)


if __name__ == '__main__':
    flags = [a for a in sys.argv[2:] if a.startswith('-')]
    subs = [a for a in sys.argv[2:] if not a.startswith('-')]
    for name, a, v, lds, spill, waves in resources(sys.argv[1], flags):
        if subs and not any(s in name for s in subs):
            continue
        print('%3d agpr %3d vgpr %6d B LDS %3d spills %d waves/SIMD  %s' % (a, v, lds, spill, waves, name))
