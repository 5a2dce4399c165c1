"""Which offset regime the benched step runs its 16 DCN layers in: per layer, the standard deviation / largest magnitude of
the sampling offsets (channels 0..17 of conv_offset_mask's output, libs/DCNv2/dcn_v2.py:119-122) and the share of
(pixel, tap) samples further than 1 / 2 / 3 px from their undeformed position, on the bench's own model and batch after
`--steps` optimizer steps.  OFFSET_STD=<px> re-initialises every conv_offset_mask like bench.py's `dcn_offsets` leg."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda', 0)
plugin = bench.build_plugin(dev, parallel=False, uda_name='entropy')
batch = bench.synthetic_batch(16, 512, 42, dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
if os.environ.get('OFFSET_STD'):
    bench.set_dcn_offset_std(plugin, batch, float(os.environ['OFFSET_STD']))
for _ in range(steps):
    plugin.step(batch)
rows = []


def hook(name):
    def f(mod, inp, out):
        off = out[:, :18].detach()
        a = off.abs()
        rows.append((name, tuple(out.shape), off.std().item(), a.max().item(), (a > 1).float().mean().item(),
                     (a > 2).float().mean().item(), (a > 3).float().mean().item()))
    return f


for n, m in plugin.backend.named_modules():
    if n.endswith('conv_offset_mask'):
        m.register_forward_hook(hook(n))
plugin.step(batch)
torch.cuda.synchronize()
print('%-44s %-22s %8s %8s %8s %8s %8s' % ('layer', 'shape', 'std px', 'max px', '>1px', '>2px', '>3px'))
for r in rows:
    print('%-44s %-22s %8.4f %8.3f %8.4f %8.4f %8.4f' % r)
