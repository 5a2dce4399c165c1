"""A/B in ONE process on ONE box: the benched step with the stem's BatchNorm + ReLU applied by level0's convolution while it
stages its input (hip_runtime.nn.BatchNorm2d.defer_apply, the default of backends.dla.DLA where the kernel exists) and with
BatchNorm's own apply pass, interleaved."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda', 0)
plugin = bench.build_plugin(dev, parallel=False, uda_name='entropy')
batch = bench.synthetic_batch(16, 512, 42, dev)
bns = [m for m in plugin.backend.modules() if getattr(m, 'defer_apply', False)]
print('BatchNorm modules with a deferred apply:', len(bns), flush=True)
for flag in (True, False):
    for m in bns:
        m.defer_apply = flag
    for _ in range(3):
        plugin.step(batch)
torch.cuda.synchronize()
for rnd in range(3):
    for flag in (True, False):
        for m in bns:
            m.defer_apply = flag
        plugin.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            plugin.step(batch)
        torch.cuda.synchronize()
        print('round %d  apply on load %-5s  %.3f ms/step' % (rnd, flag, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
