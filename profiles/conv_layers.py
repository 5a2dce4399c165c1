"""Per-layer kernel times of the convolution shapes of the benched step (DLA-34 at 512 x 512, 32 images through the
trunk): forward, input gradient and weight gradient of each shape, timed with the in-library hipEvents
(hip_runtime.prof_begin / prof_end: the launcher names the kernel it picked).

    python3 profiles/conv_layers.py [--iters 5] [--batch 32]

Environment switches are read by the library once per process (CNUDA_HCONV=0: im2col-style kernels for the 3x3 /
stride-1 layers), so A/B runs are separate invocations.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'centernet-uda_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

# (C, Co, S, k, stride): the 3x3 / 1x1 layers of DLA-34 + heads + DCN offset convolutions, by map size
SHAPES = [
    (3, 16, 512, 7, 1), (16, 16, 512, 3, 1), (16, 32, 512, 3, 2), (32, 64, 256, 3, 2),     # stem, level0, level1, level2's first
    (64, 64, 128, 3, 1), (64, 27, 128, 3, 1), (64, 256, 128, 3, 1),
    (128, 128, 64, 3, 1), (128, 27, 64, 3, 1), (64, 128, 128, 3, 2),
    (256, 256, 32, 3, 1), (256, 27, 32, 3, 1), (128, 256, 64, 3, 2),
    (512, 512, 16, 3, 1), (512, 27, 16, 3, 1), (256, 512, 32, 3, 2),
    (128, 64, 128, 1, 1), (256, 128, 64, 1, 1), (256, 6, 128, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--maps640', action='store_true', help='the same layers on the maps of a 640 x 640 input (configs[4])')
    args = ap.parse_args()
    import hip_runtime as hr
    if os.environ.get('ABL_LIB'):          # another build of the library (profiles/microbench/build_variant.sh)
        hr.LIB_PATH = os.environ['ABL_LIB']
    from hip_runtime import ops
    dev = torch.device('cuda', 0)
    print('CNUDA_HCONV=%s' % os.environ.get('CNUDA_HCONV', '(unset)'))
    print('%-28s %-58s %9s %8s' % ('layer', 'kernel', 'us', 'TFLOP/s'))
    tot = {}
    for C, Co, S, k, s in SHAPES:
        B = args.batch
        if args.maps640:
            S = S * 5 // 4
        x = torch.randn(B, C, S, S, device=dev, requires_grad=True)
        w = (torch.randn(Co, C, k, k, device=dev) / (C * k * k) ** 0.5).requires_grad_(True)
        So = (S + 2 * (k // 2) - k) // s + 1
        gy = torch.randn(B, Co, So, So, device=dev)
        for _ in range(2):
            ops.conv2d(x, w, None, s, k // 2).backward(gy)
        torch.cuda.synchronize()
        hr.prof_begin()
        for _ in range(args.iters):
            ops.conv2d(x, w, None, s, k // 2).backward(gy)
        torch.cuda.synchronize()
        res = hr.prof_end(by_shape=True)
        for (name, shape), d in sorted(res.items(), key=lambda kv: kv[0][1][0]):
            us = d['ms'] / d['launches'] * 1e3 * (d['launches'] / args.iters)
            tf = d['flops'] / (d['ms'] * 1e-3) / 1e12
            print('%-28s %-58s %9.1f %8.1f' % ('%s %d->%d %dx%d@%d/s%d' % (shape[0], C, Co, k, k, S, s), name[:58], us, tf))
            tot[shape[0]] = tot.get(shape[0], 0.0) + us
    print('totals (us): ' + ', '.join('%s %.0f' % kv for kv in sorted(tot.items())))


if __name__ == '__main__':
    main()
