"""Which gradient fan-in sums of one benched step still run as a pass of their own (cnuda_add)?  Prints the shape and the
Python call site of every add that hip_runtime.fanout.add_into / ops.add launches during one step."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
import torch                                   # noqa: E402
import hip_runtime as hr                       # noqa: E402

dev = torch.device('cuda', 0)
plugin = bench.build_plugin(dev, parallel=False, uda_name='entropy', backend_name='dla34')
batch = bench.synthetic_batch(16, 512, 42, dev, rotated=False)
for _ in range(2):
    plugin.step(batch)
seen = collections.Counter()
L = hr.lib()
orig = L.cnuda_add


def spy(a, b, out, n, st):
    frames = [f for f in traceback.extract_stack()[:-1] if 'centernet-uda_amd' in f.filename]
    seen[(int(n), ' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in frames[-3:]))] += 1
    return orig(a, b, out, n, st)


L.cnuda_add = spy
# where was the fork made whose total needs the pass?
from hip_runtime import fanout                 # noqa: E402
_fork, _add_into = fanout.fork, fanout.add_into
sites = collections.Counter()


def fork_spy(x, n=2):
    outs = _fork(x, n)
    slot = fanout.slot_of(outs[0])
    if slot is not None:
        fr = [f for f in traceback.extract_stack()[:-1] if 'centernet-uda_amd' in f.filename]
        slot_site[id(slot)] = '%s  (fork of %d, %s)' % (' <- '.join('%s:%d' % (os.path.basename(f.filename), f.lineno) for f in fr[-2:]), n, tuple(x.shape))
    return outs


def add_into_spy(slot, g):
    before = sum(seen.values())
    r = _add_into(slot, g)
    if sum(seen.values()) > before:
        sites[slot_site.get(id(slot.root()), slot_site.get(id(slot), '?'))] += 1
    return r


slot_site = {}
fanout.fork = fork_spy
fanout.add_into = add_into_spy
import backends.dla as bdla                    # noqa: E402
import libs.DCNv2.dcn_v2 as bdcn               # noqa: E402
for mod in (bdla, bdcn, hr.ops if hasattr(hr, 'ops') else bdla):
    if hasattr(mod, 'fork'):
        mod.fork = fork_spy
plugin.step(batch)
torch.cuda.synchronize()
for (n, where), k in sorted(seen.items(), key=lambda kv: -kv[0][0]):
    print('%3d x %12d elements  %s' % (k, n, where))
for site, k in sites.most_common():
    print('%3d  %s' % (k, site))
