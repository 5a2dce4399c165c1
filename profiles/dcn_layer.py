"""One DCN layer (forward + backward through the product's autograd path) at the two layer shapes that carry most of
the DCN time of the benched step: 64 -> 64 at 128 x 128 and 128 -> 64 at 64 x 64, B = 32 (16 source + 16 target).
Driver of profiles/collect_pmc_dcn.sh and of the per-layer timings quoted in DESIGN.md.
    python profiles/dcn_layer.py [--offsets small|zero|sigma1] [--iters N] [--time]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import hip_runtime as hr  # noqa: E402
if os.environ.get('ABL_LIB'):          # another build of the library (profiles/microbench/build_variant.sh)
    hr.LIB_PATH = os.environ['ABL_LIB']
from libs.DCNv2.dcn_v2 import DCN  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--offsets', default='small', choices=['zero', 'small', 'sigma1', 'sigma1.4', 'sigma2'])
ap.add_argument('--margin', type=int, default=0, help='cnuda_dcn_set_scatter_margin for the run (0: default)')
ap.add_argument('--walk-tile', type=int, default=0, help='cnuda_dcn_set_walk_tile for the run (0: default)')
ap.add_argument('--iters', type=int, default=1)
ap.add_argument('--time', action='store_true')
args = ap.parse_args()
torch.manual_seed(0)
if args.margin:
    hr.lib().cnuda_dcn_set_scatter_margin(args.margin)
if args.walk_tile:
    hr.lib().cnuda_dcn_set_walk_tile(args.walk_tile)
SHAPES = [(32, 64, 128, 64), (32, 128, 64, 64)]
if os.environ.get('DCN_LAYER_SHAPES') == 'maps640':         # configs[4]: the 160- / 80-wide maps of a 640 x 640 input
    SHAPES = [(32, 64, 160, 64), (32, 128, 80, 64)]
if os.environ.get('DCN_LAYER_SHAPES') == 'small_maps':      # the layers that take the sample + GEMM pair / the gathering loader
    SHAPES = [(32, 256, 32, 128), (32, 256, 32, 256), (32, 512, 16, 256), (32, 128, 64, 128)]
for (B, C, S, Co) in SHAPES:
    m = DCN(C, Co, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).cuda()
    with torch.no_grad():
        if args.offsets == 'small':          # offsets ~ N(0.3, 0.5 px): the regime after the first optimizer steps
            m.conv_offset_mask.weight.normal_(0, 0.02)
            m.conv_offset_mask.bias.normal_(0, 0.3)
        elif args.offsets.startswith('sigma'):
            m.conv_offset_mask.weight.normal_(0, 0.042 * float(args.offsets[5:]) * (64.0 / C) ** 0.5)
    x = torch.randn(B, C, S, S, device='cuda', requires_grad=True)
    g = torch.randn(B, Co, S, S, device='cuda')
    for _ in range(2):
        m(x).backward(g)
    torch.cuda.synchronize()
    if args.time:
        hr.prof_begin()
    for _ in range(args.iters):
        m(x).backward(g)
    torch.cuda.synchronize()
    if args.time:
        out = hr.prof_end()
        print('B=%d C=%d %dx%d Co=%d offsets=%s' % (B, C, S, S, Co, args.offsets))
        for k, v in sorted(out.items(), key=lambda kv: -kv[1]['ms']):
            print('   %-66s %8.1f us  %6.1f TF  %6.0f GB/s' % (k[:66], 1e3 * v['ms'] / v['launches'],
                  v['flops'] / v['ms'] / 1e9 if v['flops'] else 0, v['bytes'] / v['ms'] / 1e6 if v['bytes'] else 0))
