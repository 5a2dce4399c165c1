"""matrix mode 1 + pack cache + the optimizer's batched refresh (cnuda_pack_refresh): a two-layer net with the strided
16 -> 32 convolution whose input gradient aborted inside the 128 x 128 test net's training step (round 6)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
import torch
from torch import nn
import hip_runtime as hr
from hip_runtime import nn as hnn, optim
hr.set_matrix_mode(int(os.environ.get('MODE', '1')))


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = hnn.Conv2d(16, 16, 3, padding=1, bias=False)
        self.b = hnn.Conv2d(16, 32, 3, stride=2, padding=1, bias=False)
        self.c = hnn.Conv2d(32, 64, 3, stride=2, padding=1, bias=False)

    def forward(self, x):
        return self.c(self.b(self.a(x)))


net = Net().cuda()
opt = optim.Adam(net.parameters(), lr=1e-4)
for it in range(4):
    x = torch.randn(8, 16, 128, 128, device='cuda')
    loss = net(x).square().mean()
    opt.zero_grad()
    loss.backward()
    torch.cuda.synchronize()
    print('iteration', it, 'backward ok', float(loss), flush=True)
    opt.step()
    torch.cuda.synchronize()
    print('iteration', it, 'step ok', flush=True)
