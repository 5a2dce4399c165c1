"""A/B in ONE process on ONE box: the benched step (and the same step with DCN offsets of sigma = 1 / 2 px) with the
data-gradient walk's LDS window 2, 3 and 4 cells wider than the undeformed footprint (cnuda_dcn_set_scatter_margin)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import hip_runtime as hr

dev = torch.device('cuda', 0)
for sigma in (None, 1.0, 2.0):
    plugin = bench.build_plugin(dev, parallel=False, uda_name='entropy')
    batch = bench.synthetic_batch(16, 512, 42, dev)
    if sigma is not None:
        bench.set_dcn_offset_std(plugin, batch, sigma)
    for m in (2, 4):
        hr.lib().cnuda_dcn_set_scatter_margin(m)
        for _ in range(3):
            plugin.step(batch)
    torch.cuda.synchronize()
    for rnd in range(2):
        for m in (2, 3, 4):
            hr.lib().cnuda_dcn_set_scatter_margin(m)
            plugin.step(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                plugin.step(batch)
            torch.cuda.synchronize()
            print('offsets %-8s round %d  margin %d  %.3f ms/step' % ('default' if sigma is None else 'sigma=%g' % sigma, rnd, m,
                                                                       (time.perf_counter() - t0) / 8 * 1e3), flush=True)
    del plugin, batch
    torch.cuda.empty_cache()
hr.lib().cnuda_dcn_set_scatter_margin(0)
