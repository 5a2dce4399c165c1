"""Input gradient and weight gradient of one convolution are independent (both read grad_y): on ONE stream each kernel's last,
partly filled round of workgroups is exposed (few-round launches on the 64 x 64 ... 16 x 16 maps); on TWO streams the tail of one
can run beside the other.  Wall time of the pair per layer shape, one stream against two (B = 32).
    python profiles/microbench/two_stream_bwd.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hip_runtime as hr  # noqa: E402
from hip_runtime import ptr, check, lib  # noqa: E402

L = lib()
side = torch.cuda.Stream()
tot1 = tot2 = 0.0
for (C, Co, S, k, s) in [(64, 64, 128, 3, 1), (128, 128, 64, 3, 1), (256, 256, 32, 3, 1), (512, 512, 16, 3, 1), (64, 128, 128, 3, 2),
                         (128, 256, 64, 3, 2), (256, 512, 32, 3, 2), (256, 128, 64, 1, 1), (512, 256, 32, 1, 1), (64, 27, 128, 3, 1),
                         (128, 27, 64, 3, 1)]:
    B, p = 32, k // 2
    x = torch.randn(B, C, S, S, device='cuda')
    w = torch.randn(Co, C, k, k, device='cuda') * 0.05
    So = (S + 2 * p - k) // s + 1
    gy = torch.randn(B, Co, So, So, device='cuda')
    gx, gw = torch.empty_like(x), torch.empty_like(w)
    g = (B, C, S, S, Co, k, k, s, s, p, p)
    nbytes = L.cnuda_conv2d_workspace_bytes(*g)
    ws1 = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    ws2 = torch.empty(nbytes, dtype=torch.uint8, device='cuda')

    def pair(two):
        main = torch.cuda.current_stream()
        check(L.cnuda_conv2d_backward_data_add(ptr(gy), ptr(w), None, None, ptr(gx), *g, ptr(ws1), nbytes, main.cuda_stream), 'dgrad')
        if two:
            side.wait_stream(main)      # (grad_y is ready on the main stream; here it always is)
            check(L.cnuda_conv2d_backward_weight(ptr(x), ptr(gy), ptr(gw), None, *g, ptr(ws2), nbytes, side.cuda_stream), 'wgrad')
            main.wait_stream(side)
        else:
            check(L.cnuda_conv2d_backward_weight(ptr(x), ptr(gy), ptr(gw), None, *g, ptr(ws2), nbytes, main.cuda_stream), 'wgrad')

    res = []
    for two in (False, True):
        for _ in range(5):
            pair(two)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                pair(two)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)       # us per pair
        res.append(sorted(ts)[2])
    tot1 += res[0]
    tot2 += res[1]
    print('%4d -> %4d %dx%d/s%d at %3d^2: one stream %7.1f us, two streams %7.1f us (%+.1f %%)'
          % (C, Co, k, k, s, S, res[0], res[1], 100 * (res[1] - res[0]) / res[0]))
print('sum: one stream %.0f us, two streams %.0f us (%+.1f %%)' % (tot1, tot2, 100 * (tot2 - tot1) / tot1))
