"""matrix mode 1: the input gradient of 16 -> 32, 3x3 / stride 2 at 128 x 128 for a few batch sizes (which kernel, does it run)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.nn.functional as F
import hip_runtime as hr
from hip_runtime import ops
from test_zz_kernel_coverage import short
hr.set_matrix_mode(int(os.environ.get('MODE', '1')))
for B in (int(a) for a in sys.argv[1:]):
    x = torch.randn(B, 16, 128, 128, device='cuda', requires_grad=True)
    w = torch.randn(32, 16, 3, 3, device='cuda', requires_grad=True)
    with hr.launch_log() as log:
        y = ops.conv2d(x, w, None, 2, 1)
        gy = torch.randn_like(y)
        y.backward(gy)
        torch.cuda.synchronize()
    ref = torch.autograd.grad(F.conv2d(x.detach().cpu().requires_grad_(True), w.detach().cpu(), None, 2, 1), [], allow_unused=True) if False else None
    xc = x.detach().cpu().requires_grad_(True)
    F.conv2d(xc, w.detach().cpu(), None, 2, 1).backward(gy.cpu())
    print('B', B, 'max err', float((x.grad.cpu() - xc.grad).abs().max()), sorted(short(n) for n in log.names), flush=True)

# the module path: pack token (cached packed weights) + a fan-in addend on x's gradient, a few steps
from hip_runtime import nn as hnn
from hip_runtime.fanout import fork
conv = hnn.Conv2d(16, 32, 3, stride=2, padding=1, bias=False).cuda()
for B in (int(a) for a in sys.argv[1:]):
    for it in range(3):
        x = torch.randn(B, 16, 128, 128, device='cuda', requires_grad=True)
        a, b = fork(x, 2)
        y = conv(a)
        z = ops.max_pool2d(b, 2)
        (y.sum() + z.sum()).backward()
        torch.cuda.synchronize()
        print('module path B', B, 'iteration', it, 'ok', float(x.grad.abs().max()), flush=True)
