"""DCN forward alone (the `_ext` entry point: no tape), with and without the column side output, at the layer shapes of
the benched step; torch events around ITERS back-to-back calls.  Environment switches (CNUDA_DCNW=0 ...) per process.
    python3 profiles/dcn_fwd_only.py [--offsets small|zero]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import hip_runtime as hr  # noqa: E402
from hip_runtime import check, lib, ptr, stream, workspace  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--offsets', default='small', choices=['zero', 'small'])
ap.add_argument('--iters', type=int, default=10)
args = ap.parse_args()
torch.manual_seed(0)
dev = torch.device('cuda', 0)
L = lib()
print('CNUDA_DCNW=%s offsets=%s' % (os.environ.get('CNUDA_DCNW', '(unset)'), args.offsets))
for (B, C, S, Co) in [(32, 64, 128, 64), (32, 128, 64, 64), (32, 128, 64, 128), (16, 64, 128, 64)]:
    x = torch.randn(B, C, S, S, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) / (9 * C) ** 0.5
    b = torch.randn(Co, device=dev)
    off = torch.zeros(B, 18, S, S, device=dev) if args.offsets == 'zero' else (0.3 + 0.5 * torch.randn(B, 18, S, S, device=dev))
    m = torch.sigmoid(torch.randn(B, 9, S, S, device=dev))
    out = torch.empty(B, Co, S, S, device=dev)
    cols = torch.empty(B, 9 * C, S * S, device=dev)
    dims = (B, C, S, S, Co, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    nbytes = L.cnuda_dcn_v2_workspace_bytes(*dims)
    ws = workspace(nbytes, dev)
    for label, cp in (('no columns', None), ('+ columns ', cols)):
        def call():
            check(L.cnuda_dcn_v2_forward_cols(ptr(x), ptr(w), ptr(b), ptr(off), ptr(m), ptr(out), ptr(cp), *dims,
                                              ptr(ws), ws.numel(), stream()))
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        print('  B=%d %d->%d @%d  %s %8.1f us  %6.1f TF' % (B, C, Co, S, label, us, 2.0 * B * S * S * Co * C * 9 / us / 1e6))
