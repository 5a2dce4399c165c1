"""deformable_group = 2 (libs/DCNv2/testcpu.py:169-180 uses it; DLA-34 does not): the plain global-atomics path, timed beside the
deformable_group = 1 kernels at one layer shape (64 -> 64 at 128 x 128, B = 32): forward, backward, wall time by events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from libs.DCNv2.dcn_v2 import DCN  # noqa: E402

torch.manual_seed(0)
B, C, Co, S = 32, 64, 64, 128
for dg in (1, 2):
    m = DCN(C, Co, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=dg).cuda()
    with torch.no_grad():
        m.conv_offset_mask.weight.normal_(0, 0.02)
        m.conv_offset_mask.bias.normal_(0, 0.3)
    x = torch.randn(B, C, S, S, device='cuda', requires_grad=True)
    g = torch.randn(B, Co, S, S, device='cuda')
    for _ in range(2):
        m(x).backward(g)
    torch.cuda.synchronize()
    tf, tb = [], []
    for _ in range(5):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        y = m(x)
        e[1].record()
        y.backward(g)
        e[2].record()
        torch.cuda.synchronize()
        tf.append(e[0].elapsed_time(e[1]))
        tb.append(e[1].elapsed_time(e[2]))
    print('deformable_group %d: %d -> %d at %dx%d, B = %d: forward %.2f ms, backward %.2f ms (median of 5; offset convolution included)'
          % (dg, C, Co, S, S, B, sorted(tf)[2], sorted(tb)[2]))
