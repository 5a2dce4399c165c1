"""decode_detection alone (BASELINE's second metric): B=16, K=150, noise heat maps as bench.py's decode_latency
builds them; driver of profiles/collect_decode_stats.sh.   python profiles/decode_only.py [C] [H]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from backends.decode import decode_detection  # noqa: E402

B, K = 16, 150
if os.environ.get('CNUDA_DECODE_STAGE2_THREADS'):
    import hip_runtime as hr
    hr.lib().cnuda_decode_set_stage2_threads(int(os.environ['CNUDA_DECODE_STAGE2_THREADS']))
if os.environ.get('CNUDA_DECODE_BANDS'):      # A/B of the row bands of stage 1 (cnuda_decode_set_max_bands)
    import hip_runtime as hr
    hr.lib().cnuda_decode_set_max_bands(int(os.environ['CNUDA_DECODE_BANDS']))
cases = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(6, 128), (80, 128), (6, 160), (80, 160)]
for C, H in cases:
    g = torch.Generator(device='cpu').manual_seed(7 + C + H)
    heat = torch.sigmoid(torch.randn(B, C, H, H, generator=g) - 2.19).clamp(1e-4, 1 - 1e-4).cuda()
    wh, reg = (torch.rand(B, 2, H, H, generator=g) * 40).cuda(), torch.rand(B, 2, H, H, generator=g).cuda()
    for _ in range(20):
        decode_detection(heat, wh, reg, K=K)
    torch.cuda.synchronize()
    groups = []
    for _ in range(5):                 # median of five groups (host hiccups land in one group, not in the figure)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(24):
            decode_detection(heat, wh, reg, K=K)
        e1.record()
        torch.cuda.synchronize()
        groups.append(e0.elapsed_time(e1) * 1e3 / 24)
    print('C=%d %dx%d: %.1f us per call' % (C, H, H, sorted(groups)[2]))
