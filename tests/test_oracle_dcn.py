"""Pins oracle/dcn.py.  The reference's native DCNv2 cannot be built in this
image (<TH/TH.h> absent), so the pins are the reference's own known-answer
tests restated (libs/DCNv2/testcpu.py:32-67 zero-offset identity; :69-97
gradcheck), an fp64 finite-difference check of the same C code,
cross-checks against plain convolution where DCN degenerates to it, and
INDEPENDENT known answers for fractional offsets (tests/dcn_known_answers.py:
half-pixel offsets = box blur + convolution, the open validity window at -1 / H,
closed-form coordinate / mask / weight gradients on a linear-ramp image, the four
bilinear scatter weights of col2im) -- none of which share code with the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dcn as od


def _conv_identity(weight):
    weight.zero_()
    o, i, h, w = weight.shape
    for p in range(min(o, i)):
        weight[p, p, h // 2, w // 2] = 1.0


def test_zero_offset_identity_testcpu_32_67():
    # N, inC, inH, inW = 2, 2, 4, 4; outC = 2; 3x3 (testcpu.py:14-17)
    torch.manual_seed(0)
    x = torch.randn(2, 2, 4, 4)
    weight = torch.empty(2, 2, 3, 3)
    _conv_identity(weight)
    bias = torch.zeros(2)
    offset = torch.zeros(2, 18, 4, 4)
    mask = torch.sigmoid(torch.zeros(2, 9, 4, 4))      # = 0.5
    out = od.dcn_v2_forward(x, weight, bias, offset, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    assert (x - 2 * out).abs().max().item() < 1e-10    # the reference's threshold (:61)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_gradcheck_fp64_with_reference_tolerances(seed):
    # testcpu.py:69-97 shapes and value ranges, but in double (the precision
    # the upstream README says gradcheck holds in) on the very same C loops.
    torch.manual_seed(seed)
    N, C, H, W, O = 2, 2, 4, 4, 2
    x = (torch.rand(N, C, H, W, dtype=torch.float64) * 0.01).requires_grad_(True)
    offset = torch.randn(N, 18, H, W, dtype=torch.float64) * 2
    # keep sample points away from the bilinear kinks at integer coordinates
    frac = offset - torch.floor(offset)
    offset = (torch.floor(offset) + frac.clamp(0.05, 0.95)).requires_grad_(True)
    mask = torch.sigmoid(torch.rand(N, 9, H, W, dtype=torch.float64)).requires_grad_(True)
    weight = torch.randn(O, C, 3, 3, dtype=torch.float64).requires_grad_(True)
    bias = torch.rand(O, dtype=torch.float64).requires_grad_(True)
    assert torch.autograd.gradcheck(od.dcn_v2_conv, (x, offset, mask, weight, bias, 1, 1, 1, 1),
                                    eps=1e-6, atol=1e-6, rtol=1e-4)


def test_fp32_backward_matches_fp64():
    torch.manual_seed(3)
    N, C, H, W, O = 2, 8, 12, 12, 4
    args64 = [torch.randn(N, C, H, W, dtype=torch.float64), torch.randn(N, 18, H, W, dtype=torch.float64) * 2,
              torch.sigmoid(torch.randn(N, 9, H, W, dtype=torch.float64)),
              torch.randn(O, C, 3, 3, dtype=torch.float64) * 0.2, torch.randn(O, dtype=torch.float64)]
    gout = torch.randn(N, O, H, W, dtype=torch.float64)
    res = []
    for dt in (torch.float64, torch.float32):
        a = [t.detach().to(dt).clone().requires_grad_(True) for t in args64]
        y = od.dcn_v2_conv(*a, 1, 1, 1, 1)
        y.backward(gout.to(dt))
        res.append([y.detach().double()] + [t.grad.double() for t in a])
    for r64, r32 in zip(*res):
        assert (r64 - r32).abs().max().item() <= 2e-5 * max(1.0, r64.abs().max().item())


@pytest.mark.parametrize('stride,pad,dil', [(1, 1, 1), (2, 1, 1), (1, 2, 2)])
def test_zero_offset_equals_masked_conv(stride, pad, dil):
    torch.manual_seed(4)
    N, C, H, W, O = 2, 4, 9, 10, 6
    x = torch.randn(N, C, H, W)
    w = torch.randn(O, C, 3, 3)
    b = torch.randn(O)
    Ho = (H + 2 * pad - (dil * 2 + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * 2 + 1)) // stride + 1
    out = od.dcn_v2_forward(x, w, b, torch.zeros(N, 18, Ho, Wo), torch.ones(N, 9, Ho, Wo),
                            3, 3, stride, stride, pad, pad, dil, dil, 1)
    ref = F.conv2d(x, w, b, stride, pad, dil)
    assert (out - ref).abs().max().item() < 1e-4


def test_integer_offset_is_a_shift_and_oob_reads_zero():
    torch.manual_seed(5)
    N, C, H, W, O = 1, 3, 8, 8, 2
    x = torch.randn(N, C, H, W)
    w = torch.randn(O, C, 3, 3)
    b = torch.zeros(O)
    offset = torch.zeros(N, 18, H, W)
    offset[:, 0::2] = 1.0      # dh = +1 on every tap
    offset[:, 1::2] = -2.0     # dw = -2
    out = od.dcn_v2_forward(x, w, b, offset, torch.ones(N, 9, H, W), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    shifted = torch.zeros(N, C, H + 8, W + 8)
    shifted[:, :, 4:4 + H, 4:4 + W] = x
    # sample at (y+1, x-2): conv over a shifted zero-padded plane
    ref = F.conv2d(shifted[:, :, 4 + 1 - 1:4 + 1 - 1 + H + 2, 4 - 2 - 1:4 - 2 - 1 + W + 2], w, b)
    assert (out - ref).abs().max().item() < 1e-4


def test_deformable_groups_2_shapes_example_dconv():
    # testcpu.py:169-180 (example_dconv) uses deformable_groups=2; smaller here
    torch.manual_seed(6)
    x = torch.randn(2, 8, 10, 10, requires_grad=True)
    offset = torch.randn(2, 2 * 18, 10, 10, requires_grad=True)
    mask = torch.sigmoid(torch.randn(2, 2 * 9, 10, 10)).requires_grad_(True)
    w = torch.randn(6, 8, 3, 3, requires_grad=True)
    b = torch.randn(6, requires_grad=True)
    y = od.dcn_v2_conv(x, offset, mask, w, b, 1, 1, 1, 2)
    assert y.shape == (2, 6, 10, 10)
    y.sum().backward()
    assert x.grad.shape == x.shape and offset.grad.shape == offset.shape and mask.grad.shape == mask.shape
    # group 1's offsets must only see channels 4..7
    x2 = x.detach().clone(); x2[:, 4:] = 0
    y2 = od.dcn_v2_forward(x2, w.detach(), b.detach(), offset.detach(), mask.detach(), 3, 3, 1, 1, 1, 1, 1, 1, 2)
    off3 = offset.detach().clone(); off3[:, 18:] += 0.7
    y3 = od.dcn_v2_forward(x2, w.detach(), b.detach(), off3, mask.detach(), 3, 3, 1, 1, 1, 1, 1, 1, 2)
    assert (y2 - y3).abs().max().item() == 0


def test_channel_mismatch_raises():
    with pytest.raises(RuntimeError):
        od.dcn_v2_forward(torch.zeros(1, 3, 4, 4), torch.zeros(2, 4, 3, 3), torch.zeros(2),
                          torch.zeros(1, 18, 4, 4), torch.zeros(1, 9, 4, 4), 3, 3, 1, 1, 1, 1, 1, 1, 1)


# ---------------------------------------------------------------------------
# Independent known answers for fractional offsets (closed forms / plain convolutions only; shared with the
# MI355X tests: tests/dcn_known_answers.py)
# ---------------------------------------------------------------------------
import dcn_known_answers as ka  # noqa: E402

_DT = [(torch.float64, 1e-12), (torch.float32, 2e-5)]


@pytest.mark.parametrize('dtype,tol', _DT)
@pytest.mark.parametrize('dh,dw', ka.BOX_OFFSETS)
def test_uniform_fractional_offset_is_a_box_blur_then_conv(dh, dw, dtype, tol):
    ka.check_uniform_fractional_offset_is_box_blur_then_conv(od.dcn_v2_forward, dh, dw, dtype, tol)


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-13), (torch.float32, 1e-6)])
def test_validity_window_is_open_at_minus_one_and_at_H(dtype, tol):
    ka.check_validity_window_is_open(od.dcn_v2_forward, od.dcn_v2_backward, dtype, tol)


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-11), (torch.float32, 3e-5)])
def test_linear_ramp_has_constant_coordinate_gradient(dtype, tol):
    ka.check_linear_ramp_has_constant_coordinate_gradient(od.dcn_v2_forward, od.dcn_v2_backward, dtype, tol)


@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-12), (torch.float32, 1e-6)])
def test_col2im_scatters_the_four_bilinear_weights(dtype, tol):
    ka.check_col2im_scatters_the_four_bilinear_weights(od.dcn_v2_backward, dtype, tol)
