"""dcnq_kernel / the three-kernel DCN backward with the caches cold: a 2 GB tensor is rewritten between the forward and the
backward of the layer, as the rest of a training step does.    CNUDA_DCNQ=1|0 python3 profiles/microbench/dcnq_cold.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hip_runtime as hr  # noqa: E402
from libs.DCNv2.dcn_v2 import DCN  # noqa: E402
torch.manual_seed(0)
junk = torch.zeros(512 * 1024 * 1024, device='cuda')
for (B, C, S, Co) in [(32, 64, 128, 64), (32, 128, 64, 64)]:
    m = DCN(C, Co, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).cuda()
    x = torch.randn(B, C, S, S, device='cuda', requires_grad=True)
    g = torch.randn(B, Co, S, S, device='cuda')
    for cold in (False, True):
        for _ in range(2):
            m(x).backward(g)
        torch.cuda.synchronize()
        hr.prof_begin()
        for _ in range(3):
            y = m(x)
            if cold:
                junk.add_(1.0)
            y.backward(g)
        torch.cuda.synchronize()
        out = hr.prof_end()
        for k, v in sorted(out.items(), key=lambda kv: -kv[1]['ms']):
            if any(s in k for s in ('dcnq', 'dcn_bwd_data', 'shortk')):
                print('B=%d C=%d %dx%d cold=%s  %-44s %8.1f us' % (B, C, S, S, cold, k[:44], 1e3 * v['ms'] / v['launches']))
