"""One DCN layer forward + backward (B = 32), wall time per iteration by events: the data-gradient chain (column-gradient
GEMM + walk) over the whole batch against image chunks (CNUDA_DCN_BWD_CHUNK, read once per process: run per setting).
    CNUDA_DCN_BWD_CHUNK=4 python profiles/microbench/dcn_bwd_chunk_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hip_runtime as hr  # noqa: E402
from libs.DCNv2.dcn_v2 import DCN  # noqa: E402

torch.manual_seed(0)
for (B, C, S, Co) in [(32, 64, 128, 64), (32, 128, 64, 64)]:
    m = DCN(C, Co, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).cuda()
    with torch.no_grad():
        m.conv_offset_mask.weight.normal_(0, 0.02)
        m.conv_offset_mask.bias.normal_(0, 0.3)
    x = torch.randn(B, C, S, S, device='cuda', requires_grad=True)
    g = torch.randn(B, Co, S, S, device='cuda')
    for _ in range(3):
        m(x).backward(g)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m(x).backward(g)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    hr.prof_begin()
    for _ in range(3):
        m(x).backward(g)
    torch.cuda.synchronize()
    out = hr.prof_end()
    walk = sum(v['ms'] for k, v in out.items() if 'dcn_bwd_data' in k) / 3
    gemm = sum(v['ms'] for k, v in out.items() if 'shortk' in k or ('igemm_fwd' in k and 'ConvFwdBuf' in k)) / 3
    print('chunk %s  B=%d C=%d %dx%d: layer fwd+bwd %.3f ms (median of 5)   walk %.3f ms   column-gradient GEMM %.3f ms'
          % (os.environ.get('CNUDA_DCN_BWD_CHUNK', '0'), B, C, S, S, sorted(ts)[2], walk, gemm))
