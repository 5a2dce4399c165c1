"""dcnq_kernel time against the number of workgroups (256 tiles per 128 x 128 image): how many workgroups a CU runs at once.
    CNUDA_DCNQ=1 python3 profiles/microbench/dcnq_scaling.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hip_runtime as hr  # noqa: E402
from libs.DCNv2.dcn_v2 import DCN  # noqa: E402
torch.manual_seed(0)
C = Co = 64
S = 128
m = DCN(C, Co, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).cuda()
for B in (1, 2, 3, 4, 8, 16, 32):
    x = torch.randn(B, C, S, S, device='cuda', requires_grad=True)
    g = torch.randn(B, Co, S, S, device='cuda')
    for _ in range(2):
        m(x).backward(g)
    torch.cuda.synchronize()
    hr.prof_begin()
    for _ in range(5):
        m(x).backward(g)
    torch.cuda.synchronize()
    out = hr.prof_end()
    for k, v in out.items():
        if 'dcnq' in k or 'bwd_data' in k:
            print('B=%2d tiles=%5d  %-30s %8.1f us' % (B, B * S * S // 64, k[:30], 1e3 * v['ms'] / v['launches']))
