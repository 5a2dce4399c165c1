import sys, os
sys.path.insert(0, 'centernet-uda_amd'); sys.path.insert(0, '.')
import torch, torch.nn.functional as F
import hip_runtime as hr
from hip_runtime import ops
torch.manual_seed(0)
for (B,C,H,W,Co) in [(3,64,16,16,256),(2,64,8,8,256),(3,64,16,16,128),(1,64,16,16,256),(3,128,16,16,256)]:
    x = torch.randn(B,C,H,W)
    w = torch.randn(Co,C,3,3)/ (C*9)**0.5
    gy = torch.randn(B,Co,H,W)
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, 1, 1).backward(gy)
    for mt in (128, 0):
        xd = x.cuda().requires_grad_(True); wd = w.cuda().requires_grad_(True)
        with hr.splitk(mt), hr.launch_log() as log:
            y = ops.conv2d(xd, wd, None, 1, 1)
            y.backward(gy.cuda())
        err = (xd.grad.cpu() - xr.grad).abs().max().item()
        bad = ((xd.grad.cpu() - xr.grad).abs() > 1e-4).nonzero()
        print((B,C,H,W,Co), 'max_tiles', mt, 'dgrad err', err, 'bad count', len(bad), bad[:6].tolist(), [n.split('(')[0][-60:] for n in log.names if 'Dgrad' in n])
