"""The split-operand matrix mode (cnuda_set_matrix_mode(1): every f32 operand cut exactly into three bf16 pieces,
six partial products on the bf16 MFMA, f32 accumulation with periodic folds) must meet the same bar as the
default f32-MFMA mode: 1e-4 of the tensor's scale against the CPU torch primitives the oracle is made of
(tests/test_gpu_ops.py), and it must not carry the bf16 MFMA's accumulation bias into long coherent sums."""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import CONV_CASES, test_conv2d_fwd_bwd as _conv_case

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture
def split_mode():
    import hip_runtime as hr
    before = hr.get_matrix_mode()
    hr.set_matrix_mode(1)
    yield
    hr.set_matrix_mode(before)


@pytest.mark.parametrize('name', sorted(CONV_CASES))
def test_conv2d_fwd_bwd_split_mode(split_mode, name):
    import hip_runtime as hr
    assert hr.get_matrix_mode() == 1
    _conv_case(name)


def test_mode_setter_rejects_unknown_modes():
    import hip_runtime as hr
    before = hr.get_matrix_mode()
    with pytest.raises(RuntimeError, match='mode must be 0'):
        hr.set_matrix_mode(2)
    assert hr.get_matrix_mode() == before


@pytest.mark.parametrize('C,Co,H', [(256, 256, 16), (512, 128, 16)])
def test_long_coherent_sums_carry_no_bias(split_mode, C, Co, H):
    """All-positive operands: every partial sum grows monotonically, which is where an accumulator that
    truncates instead of rounding shows (-3e-6 / -1e-5 of the result at K = 2304 / 4608 without the folds).
    Bar: mean relative error within 2e-7 of zero, worst element within 2e-6 (the f32 chain's own level)."""
    from hip_runtime import ops
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, C, H, H, generator=g)
    w = torch.rand(Co, C, 3, 3, generator=g) * 0.05
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    y = ops.conv2d(x.to(DEV), w.to(DEV), None, 1, 1).double().cpu()
    rel = (y - ref) / ref
    assert abs(rel.mean().item()) < 2e-7, rel.mean().item()
    assert rel.abs().max().item() < 2e-6, rel.abs().max().item()


def test_dcn_backward_split_mode_matches_f32_mode(split_mode):
    """The DCN backward's column-gradient GEMM goes through the convolution entry point and so follows the mode."""
    import hip_runtime as hr
    import _ext
    g = torch.Generator().manual_seed(11)
    B, C, Co, H, W = 2, 64, 64, 12, 10
    x = torch.randn(B, C, H, W, generator=g).to(DEV)
    w = (torch.randn(Co, C, 3, 3, generator=g) / 24).to(DEV)
    b = torch.randn(Co, generator=g).to(DEV)
    off = (torch.randn(B, 18, H, W, generator=g) * 0.7).to(DEV)
    m = torch.sigmoid(torch.randn(B, 9, H, W, generator=g)).to(DEV)
    gy = torch.randn(B, Co, H, W, generator=g).to(DEV)
    args = (x, w, b, off, m, gy, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    got = _ext.dcn_v2_backward(*args)
    hr.set_matrix_mode(0)
    want = _ext.dcn_v2_backward(*args)
    for a, r in zip(got, want):
        scale = max(1.0, r.abs().max().item())
        assert (a - r).abs().max().item() <= 1e-5 * scale
