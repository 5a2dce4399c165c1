"""A/B in ONE process on ONE box: the benched step with the BatchNorm statistics taken in the GEMM / DCN epilogues
(hip_runtime.ops.EPILOGUE_STATS = True, the default) and with BatchNorm's own statistics pass, interleaved."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from hip_runtime import ops

dev = torch.device('cuda', 0)
plugin = bench.build_plugin(dev, parallel=False, uda_name='entropy')
batch = bench.synthetic_batch(16, 512, 42, dev)
for flag in (True, False):
    ops.EPILOGUE_STATS = flag
    for _ in range(3):
        plugin.step(batch)
torch.cuda.synchronize()
for rnd in range(3):
    for flag in (True, False):
        ops.EPILOGUE_STATS = flag
        plugin.step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            plugin.step(batch)
        torch.cuda.synchronize()
        print('round %d  epilogue statistics %-5s  %.3f ms/step' % (rnd, flag, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
