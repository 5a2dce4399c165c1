"""Which python lines of one benched step launch kernels that are NOT this library's (ATen elementwise / fill /
copy kernels, runtime copyBuffer / fillBuffer): torch.profiler with stacks, grouped by the innermost frame inside
the repo.    python3 profiles/aten_sources.py [--uda entropy]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
plugin = bench.build_plugin(dev, parallel=False)
batch = bench.synthetic_batch(16, 512, 42, dev)
for _ in range(3):
    plugin.step(bench.fresh(batch))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    plugin.step(bench.fresh(batch))
    torch.cuda.synchronize()
ops = collections.Counter()
where = collections.defaultdict(collections.Counter)
for e in prof.events():
    if not e.name.startswith('aten::') or not e.kernels:
        continue
    ks = [k.name for k in e.kernels if 'cnuda' not in k.name]
    if not ks:
        continue
    frame = next((f for f in e.stack if ROOT in f or 'centernet-uda_amd' in f), e.stack[0] if e.stack else '?')
    frame = frame.replace(ROOT + '/', '')
    for k in ks:
        short = k.split('<')[0].split('(')[0][-40:] + ('<add>' if 'add' in k.lower() else '<fill>' if 'Fill' in k else '')
        ops[short] += 1
        where[short][e.name + ' @ ' + frame] += 1
for k, n in ops.most_common():
    print('%4d  %s' % (n, k))
    for w, c in where[k].most_common(12):
        print('        %4d  %s' % (c, w))
