"""Forward / input-gradient error of long-K convolutions against an fp64 evaluation, relative to the tensor's maximum:
the f32 MFMA's k-ordered chain (product), the same chain folded every 8 chunks (abl/lib_fold8.so, -DIG_FOLD=8), split-K.
    [ABL_LIB=abl/lib_fold8.so] python profiles/microbench/accum_error.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import hip_runtime as hr  # noqa: E402
if os.environ.get('ABL_LIB'):
    hr.LIB_PATH = os.environ['ABL_LIB']
from hip_runtime import ops  # noqa: E402

torch.manual_seed(1)
print('library:', os.environ.get('ABL_LIB', 'product'))
for (B, C, S, Co) in [(32, 512, 16, 512), (32, 256, 32, 256), (32, 128, 64, 128), (32, 64, 128, 64)]:
    x = torch.rand(B, C, S, S, device='cuda') + 0.1            # all-positive operands: every partial sum grows (worst case of a chain)
    w = torch.rand(Co, C, 3, 3, device='cuda') * 0.1
    gy = torch.rand(B, Co, S, S, device='cuda')
    want_y = F.conv2d(x.double(), w.double(), None, 1, 1)
    want_gx = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), 1, 1)
    for max_tiles in (0, 100000):
        with hr.splitk(max_tiles):
            xx = x.clone().requires_grad_(True)
            y = ops.conv2d(xx, w, None, 1, 1)
            y.backward(gy)
        ey = ((y.double() - want_y).abs().max() / want_y.abs().max()).item()
        eg = ((xx.grad.double() - want_gx).abs().max() / want_gx.abs().max()).item()
        mean_y = ((y.double() - want_y).mean() / want_y.abs().max()).item()
        print('%4d -> %4d 3x3 at %3d^2 (K = %d)  split-K %-3s  forward max %.2e (mean %+.1e)  input gradient max %.2e'
              % (C, Co, S, 9 * C, 'on' if max_tiles else 'off', ey, mean_y, eg))
