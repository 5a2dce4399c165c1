"""The inference wrapper alone: export.CenterNet on a random-init DLA-34 + DCNv2 backend, eval
forward + decode on a resident batch of 16 x 512 x 512; driver of profiles/collect_infer_stats.sh.
   python profiles/infer_only.py [batch] [size] [calls]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 10
device = torch.device('cuda:0')
plugin = bench.build_plugin(device, parallel=False, uda_name='entropy', backend_name='dla34')
from export import CenterNet  # noqa: E402

model = CenterNet(plugin.backend, bench.MAX_OBJS).eval()
x = torch.randn(batch, 3, size, size, device=device)
for _ in range(3):
    model(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(calls):
    model(x)
e1.record()
torch.cuda.synchronize()
print('inference: %.3f ms per batch of %d (%dx%d)' % (e0.elapsed_time(e1) / calls, batch, size, size))
