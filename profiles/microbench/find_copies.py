"""Where do the small device copies / ATen kernels of one training step come from?  torch.profiler with Python stacks
over one warm step of the bench's own plugin; prints every ATen op that launched device work, by its innermost frame
inside this repository.  python profiles/microbench/find_copies.py"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device('cuda', 0)
plugin = bench.build_plugin(dev, parallel=False)
batch = bench.synthetic_batch(16, 512, 42, dev)
for _ in range(3):
    plugin.step(batch)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    plugin.step(batch)
    torch.cuda.synchronize()
by = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::'):
        continue
    if not ev.kernels:
        continue
    frames = [f for f in (ev.stack or []) if 'centernet-uda_amd' in f or 'bench.py' in f]
    where = frames[0].split('centernet-uda_amd/')[-1] if frames else (ev.stack[0] if ev.stack else '?')
    for k in ev.kernels:
        by[(ev.name, k.name[:50], where)] += 1
for (op, kern, where), n in sorted(by.items(), key=lambda kv: -kv[1]):
    print('%4d  %-28s %-50s %s' % (n, op, kern, where))
