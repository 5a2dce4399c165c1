"""Independent known answers for modulated deformable convolution with FRACTIONAL offsets.

Nothing here shares code with the oracle or with the HIP kernels: the expected values come from plain
convolutions with box-composed kernels or from closed forms.  Each check takes the implementation under test
as `fwd(input, weight, bias, offset, mask, *geom)` / `bwd(input, weight, bias, offset, mask, grad_out, *geom)`
(the native `_ext` signatures, libs/DCNv2/src/dcn_v2.h:10-47) on CPU tensors, so the same vectors pin the CPU
oracle (tests/test_oracle_dcn.py) and the MI355X kernels (tests/test_gpu_dcn.py).

Reference rules being pinned: bilinear sampling with zero outside the plane
(libs/DCNv2/src/cpu/dcn_v2_im2col_cpu.cpp:27-56), the validity window -1 < h < H, -1 < w < W (:160-166),
the scatter weights of col2im (:58-82, 198-257) and the coordinate weights (:84-125, 259-329).
"""
import numpy as np
import torch
import torch.nn.functional as F

BOX_OFFSETS = [(0.5, 0.0), (0.0, 0.5), (0.5, 0.5), (-0.5, 0.5), (0.25, -0.75)]


def _uniform_offset(N, H, W, dh, dw, dtype):
    off = torch.zeros(N, 18, H, W, dtype=dtype)
    off[:, 0::2] = dh
    off[:, 1::2] = dw
    return off


def check_uniform_fractional_offset_is_box_blur_then_conv(fwd, dh, dw, dtype, tol, size=(2, 3, 7, 9, 4)):
    """A constant offset (dh, dw) on every tap samples the bilinearly shifted, zero-extended image, i.e. the
    zero-padded image convolved with the 2x2 kernel [[(1-fh)(1-fw), (1-fh)fw], [fh(1-fw), fh fw]] at an integer
    shift -- half-pixel offsets are plain 2- and 4-neighbour means.  Expected values: F.conv2d only."""
    torch.manual_seed(11)
    N, C, H, W, O = size
    x = torch.randn(N, C, H, W, dtype=dtype)
    w = torch.randn(O, C, 3, 3, dtype=dtype) / (C * 9) ** 0.5
    b = torch.randn(O, dtype=dtype)
    tapmask = torch.rand(9, dtype=dtype) + 0.25                     # modulation, constant per tap
    mask = tapmask.view(1, 9, 1, 1).expand(N, 9, H, W).contiguous()
    out = fwd(x, w, b, _uniform_offset(N, H, W, dh, dw, dtype), mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    ih, iw = int(np.floor(dh)), int(np.floor(dw))
    fh, fw = dh - ih, dw - iw
    P = 4                                                            # zero margin, wide enough for every shift
    xp = F.pad(x.double(), (P, P, P, P))
    box = torch.tensor([[(1 - fh) * (1 - fw), (1 - fh) * fw], [fh * (1 - fw), fh * fw]], dtype=torch.float64)
    # shifted[y, x] = sum_ab box[a, b] * xp[y + a, x + b]: the image sampled at (y + fh, x + fw)
    shifted = F.conv2d(xp.reshape(N * C, 1, H + 2 * P, W + 2 * P), box.view(1, 1, 2, 2))
    shifted = shifted.reshape(N, C, H + 2 * P - 1, W + 2 * P - 1)
    # tap (i, j) of output (y, x) reads the image at (y - 1 + i + ih + fh, x - 1 + j + iw + fw)
    y0, x0 = P - 1 + ih, P - 1 + iw
    ref = F.conv2d(shifted[:, :, y0:y0 + H + 2, x0:x0 + W + 2], (w * tapmask.view(1, 1, 3, 3)).double(), b.double())
    assert (out.double() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


def check_validity_window_is_open(fwd, bwd, dtype, tol):
    """1x1 kernel, no padding: out[y, x] = w * mask * bilinear(I, y + dh, x + dw).  Samples in (-1, 0) and
    (H-1, H) blend the edge pixel with the zero outside; a sample at exactly -1 or exactly H (or beyond) is 0."""
    H, W = 4, 5
    I = torch.arange(1, H * W + 1, dtype=dtype).reshape(1, 1, H, W)
    w = torch.full((1, 1, 1, 1), 2.0, dtype=dtype)
    b = torch.zeros(1, dtype=dtype)
    mask = torch.full((1, 1, H, W), 0.5, dtype=dtype)               # w * mask = 1
    off = torch.zeros(1, 2, H, W, dtype=dtype)
    exp = I.clone()                                                  # zero offset: identity
    cases = [  # (y, x, dh, dw, expected)
        (0, 1, -0.25, 0.0, 0.75 * I[0, 0, 0, 1]),                    # h = -0.25: 0.75 * row 0 + 0.25 * outside
        (0, 2, -1.0, 0.0, 0.0),                                      # h = -1 exactly: outside the open window
        (0, 3, -0.875, 0.0, 0.125 * I[0, 0, 0, 3]),                  # just inside (exactly representable)
        (3, 0, 0.75, 0.0, 0.25 * I[0, 0, 3, 0]),                     # h = H - 0.25
        (3, 1, 1.0, 0.0, 0.0),                                       # h = H exactly
        (3, 2, 0.875, 0.0, 0.125 * I[0, 0, 3, 2]),
        (1, 0, 0.0, -0.5, 0.5 * I[0, 0, 1, 0]),                      # w = -0.5
        (1, 4, 0.0, 0.5, 0.5 * I[0, 0, 1, 4]),                       # w = W - 0.5
        (2, 4, 0.0, 1.0, 0.0),                                       # w = W exactly
        (2, 0, 0.0, -1.0, 0.0),                                      # w = -1 exactly
        (0, 0, -0.5, -0.5, 0.25 * I[0, 0, 0, 0]),                    # corner: three of four neighbours outside
        (3, 4, 0.5, 0.5, 0.25 * I[0, 0, 3, 4]),
        (2, 2, 5.0, 0.0, 0.0), (1, 1, 0.0, -7.5, 0.0),               # far outside
        (1, 2, 0.5, 0.5, 0.25 * (I[0, 0, 1, 2] + I[0, 0, 1, 3] + I[0, 0, 2, 2] + I[0, 0, 2, 3])),
        (2, 1, -1.0, 1.0, I[0, 0, 1, 2]),                            # integer offsets: an exact pixel read
    ]
    for y, x, dh, dw, e in cases:
        off[0, 0, y, x], off[0, 1, y, x], exp[0, 0, y, x] = dh, dw, float(e)
    geom = (1, 1, 1, 1, 0, 0, 1, 1, 1)
    out = fwd(I, w, b, off, mask, *geom)
    assert (out - exp).abs().max().item() <= tol * 20
    # gradients at the window's edge: nothing flows from samples at -1 / H / beyond
    gi, go, gm, gw, gb = bwd(I, w, b, off, mask, torch.ones_like(out), *geom)
    for y, x, dh, dw, e in cases:
        if float(e) == 0.0:
            assert go[0, :, y, x].abs().max().item() == 0 and gm[0, 0, y, x].item() == 0, (y, x)
    # h = -0.25 at (0, 1): d out / d dh = w*mask * (I[0,1] - 0) = I[0,1]; d out / d mask = w * sample
    assert abs(go[0, 0, 0, 1].item() - I[0, 0, 0, 1].item()) <= tol * 20
    assert abs(gm[0, 0, 0, 1].item() - 2.0 * 0.75 * I[0, 0, 0, 1].item()) <= tol * 20
    # w = W - 0.5 at (1, 4): d out / d dw = (0 - I[1,4]) * w*mask
    assert abs(go[0, 1, 1, 4].item() + I[0, 0, 1, 4].item()) <= tol * 20
    # grad_input: every in-plane corner weight of every valid sample, accumulated (closed form)
    egi = torch.zeros(H, W, dtype=torch.float64)
    for y in range(H):
        for x in range(W):
            h, ww = y + off[0, 0, y, x].item(), x + off[0, 1, y, x].item()
            if not (-1 < h < H and -1 < ww < W):
                continue
            h0, w0 = int(np.floor(h)), int(np.floor(ww))
            for yy, wy in ((h0, 1 - (h - h0)), (h0 + 1, h - h0)):
                for xx, wx in ((w0, 1 - (ww - w0)), (w0 + 1, ww - w0)):
                    if 0 <= yy < H and 0 <= xx < W:
                        egi[yy, xx] += wy * wx                       # w * mask * gout = 1
    assert (gi[0, 0].double() - egi).abs().max().item() <= tol * 20


def check_linear_ramp_has_constant_coordinate_gradient(fwd, bwd, dtype, tol, size=(2, 3, 12, 11, 2)):
    """On I[c](y, x) = a_c*y + b_c*x + k_c bilinear interpolation is exact, so for samples whose four
    neighbours are inside the plane: sample = a h + b w + k, d/dh = a, d/dw = b (closed forms).
    Pins col2im_coord's offset / mask gradients, the weight gradient, and col2im's mass and first moments."""
    torch.manual_seed(13)
    N, C, H, W, O = size
    a = torch.linspace(-1.3, 0.7, C, dtype=dtype)
    bb = torch.linspace(0.9, -0.4, C, dtype=dtype)
    k = torch.linspace(-2.0, 5.0, C, dtype=dtype)
    ys = torch.arange(H, dtype=dtype).view(1, 1, H, 1)
    xs = torch.arange(W, dtype=dtype).view(1, 1, 1, W)
    x = (a.view(1, C, 1, 1) * ys + bb.view(1, C, 1, 1) * xs + k.view(1, C, 1, 1)).expand(N, C, H, W).contiguous()
    w = torch.randn(O, C, 3, 3, dtype=dtype) / (C * 9) ** 0.5
    bias = torch.randn(O, dtype=dtype)
    # offsets in (-0.9, 0.9) px; only output pixels at least 2 px from the border get a gradient (all taps interior)
    off = (torch.rand(N, 18, H, W, dtype=dtype) * 1.8 - 0.9)
    mask = torch.rand(N, 9, H, W, dtype=dtype) * 0.8 + 0.1
    gout = torch.zeros(N, O, H, W, dtype=dtype)
    gout[:, :, 2:H - 2, 2:W - 2] = torch.randn(N, O, H - 4, W - 4, dtype=dtype)
    geom = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    out = fwd(x, w, bias, off, mask, *geom).double()
    gi, go, gm, gw, gb = [t.double() for t in bwd(x, w, bias, off, mask, gout, *geom)]
    # closed forms, evaluated in float64
    a, bb, k, ys, xs, w, bias, off, mask, gout = [t.double() for t in (a, bb, k, ys, xs, w, bias, off, mask, gout)]
    tap_i = torch.arange(3, dtype=torch.float64).repeat_interleave(3).view(1, 9, 1, 1)
    tap_j = torch.arange(3, dtype=torch.float64).repeat(3).view(1, 9, 1, 1)
    hpos = ys - 1 + tap_i + off[:, 0::2]                               # [N, 9, H, W]
    wpos = xs - 1 + tap_j + off[:, 1::2]
    samp = (a.view(1, C, 1, 1, 1) * hpos.unsqueeze(1) + bb.view(1, C, 1, 1, 1) * wpos.unsqueeze(1)
            + k.view(1, C, 1, 1, 1))                                   # [N, C, 9, H, W]
    col = samp * mask.unsqueeze(1)
    wf = w.view(O, C, 9)
    exp_out = torch.einsum('oct,ncthw->nohw', wf, col) + bias.view(1, O, 1, 1)
    inner = (slice(None), slice(None), slice(2, H - 2), slice(2, W - 2))
    assert (out[inner] - exp_out[inner]).abs().max().item() <= tol * exp_out[inner].abs().max().item()
    dcol = torch.einsum('oct,nohw->ncthw', wf, gout)                   # d loss / d column
    exp_gm = (dcol * samp).sum(1)
    exp_goh = (dcol * a.view(1, C, 1, 1, 1)).sum(1) * mask
    exp_gow = (dcol * bb.view(1, C, 1, 1, 1)).sum(1) * mask
    exp_go = torch.stack([exp_goh, exp_gow], 2).reshape(N, 18, H, W)   # channel 2*tap = dh, 2*tap + 1 = dw
    assert (gm - exp_gm).abs().max().item() <= tol * max(1.0, exp_gm.abs().max().item())
    assert (go - exp_go).abs().max().item() <= tol * max(1.0, exp_go.abs().max().item())
    exp_gw = torch.einsum('nohw,ncthw->oct', gout, col).view(O, C, 3, 3)
    assert (gw - exp_gw).abs().max().item() <= tol * max(1.0, exp_gw.abs().max().item())
    assert (gb - gout.sum((0, 2, 3))).abs().max().item() <= tol * max(1.0, gout.abs().sum().item())
    # col2im: the four bilinear weights of an interior sample sum to 1 and have first moments (h, w):
    # total mass and first moments of grad_input per (n, c) follow in closed form
    top = dcol * mask.unsqueeze(1)                                     # [N, C, 9, H, W]
    mass_scale = max(1.0, top.abs().sum().item() / (N * C))
    assert (gi.sum((2, 3)) - top.sum((2, 3, 4))).abs().max().item() <= tol * mass_scale
    m_h, m_w = (gi * ys).sum((2, 3)), (gi * xs).sum((2, 3))
    assert (m_h - (top * hpos.unsqueeze(1)).sum((2, 3, 4))).abs().max().item() <= tol * mass_scale * H
    assert (m_w - (top * wpos.unsqueeze(1)).sum((2, 3, 4))).abs().max().item() <= tol * mass_scale * W


def check_col2im_scatters_the_four_bilinear_weights(bwd, dtype, tol):
    """One output pixel, one tap (1x1 kernel), offset (0.3, 0.6): grad_input has exactly four non-zeros,
    (1-0.3)(1-0.6), (1-0.3)0.6, 0.3(1-0.6), 0.3*0.6 times w*mask*gout, at (y, x), (y, x+1), (y+1, x), (y+1, x+1)."""
    H, W = 6, 6
    torch.manual_seed(17)
    x = torch.randn(1, 1, H, W, dtype=dtype)
    w = torch.full((1, 1, 1, 1), 3.0, dtype=dtype)
    mask = torch.full((1, 1, H, W), 0.5, dtype=dtype)
    off = torch.zeros(1, 2, H, W, dtype=dtype)
    off[0, 0, 2, 3], off[0, 1, 2, 3] = 0.3, 0.6
    gout = torch.zeros(1, 1, H, W, dtype=dtype)
    gout[0, 0, 2, 3] = 2.0
    geom = (1, 1, 1, 1, 0, 0, 1, 1, 1)
    gi = bwd(x, w, torch.zeros(1, dtype=dtype), off, mask, gout, *geom)[0]
    exp = torch.zeros_like(gi)
    s = 3.0 * 0.5 * 2.0
    exp[0, 0, 2, 3], exp[0, 0, 2, 4], exp[0, 0, 3, 3], exp[0, 0, 3, 4] = s * 0.7 * 0.4, s * 0.7 * 0.6, s * 0.3 * 0.4, s * 0.3 * 0.6
    assert (gi - exp).abs().max().item() <= tol
    # at the open edge: a sample at h = H - 0.5 keeps only the in-plane row
    off.zero_(); gout.zero_()
    off[0, 0, 5, 1] = 0.5
    gout[0, 0, 5, 1] = 1.0
    gi = bwd(x, w, torch.zeros(1, dtype=dtype), off, mask, gout, *geom)[0]
    exp.zero_()
    exp[0, 0, 5, 1] = 3.0 * 0.5 * 0.5
    assert (gi - exp).abs().max().item() <= tol
