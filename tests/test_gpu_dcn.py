"""MI355X parity of `_ext.dcn_v2_forward/backward` (HIP) against the CPU oracle,
plus the reference's own known-answer tests run through the HIP path."""
import zlib

import numpy as np
import pytest
import torch

from oracle import dcn as od

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL = 1e-4       # north_star: fp32 within 1e-4 (relative to the tensor's scale)


def _case(seed, B, C, Co, H, W, k=3, s=1, p=1, d=1, dg=1, off_scale=2.0):
    g = torch.Generator().manual_seed(seed)
    Ho = (H + 2 * p - (d * (k - 1) + 1)) // s + 1
    Wo = (W + 2 * p - (d * (k - 1) + 1)) // s + 1
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    off = torch.randn(B, 2 * k * k * dg, Ho, Wo, generator=g) * off_scale
    m = torch.sigmoid(torch.randn(B, k * k * dg, Ho, Wo, generator=g))
    go = torch.randn(B, Co, Ho, Wo, generator=g)
    return (x, w, b, off, m, go), (k, k, s, s, p, p, d, d, dg)


def _close(a, b, tol=TOL):
    a, b = a.double().cpu(), b.double().cpu()
    scale = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert err <= tol * scale, (err, scale)


CASES = {
    'testcpu_shape': dict(B=2, C=2, Co=2, H=4, W=4),                # libs/DCNv2/testcpu.py:14-17
    'g1_mid': dict(B=2, C=8, Co=4, H=12, W=12),
    'dla_64': dict(B=1, C=64, Co=64, H=16, W=16),
    'odd_everything': dict(B=3, C=20, Co=37, H=9, W=11),
    'cout_over_64': dict(B=1, C=32, Co=100, H=10, W=13),
    'stride2': dict(B=2, C=16, Co=8, H=11, W=10, s=2),
    'dilated': dict(B=1, C=16, Co=16, H=12, W=12, p=2, d=2),
    'big_offsets_oob': dict(B=2, C=16, Co=16, H=8, W=8, off_scale=6.0),
    'k1': dict(B=2, C=16, Co=8, H=6, W=7, k=1, p=0),
    'dg2': dict(B=2, C=8, Co=6, H=10, W=10, dg=2),                   # testcpu.py:169-180 uses dg=2
    # round 6: deformable_group > 1 is composed from the deformable_group = 1 kernels per group (offsets / mask read in place
    # through the batch strides of the dg-group tensors): 32 channels per group on the MFMA loaders, 4 groups, a window layer
    'dg2_64ch': dict(B=2, C=64, Co=32, H=16, W=32, dg=2),
    'dg4_odd': dict(B=1, C=32, Co=20, H=9, W=13, dg=4, off_scale=1.0),
    'c512': dict(B=1, C=512, Co=256, H=4, W=4),                      # ida_0.proj_1 shape at 128^2 input
    'width1': dict(B=1, C=4, Co=4, H=5, W=1),                          # no horizontal neighbour: generic kernels
    'width2': dict(B=2, C=16, Co=16, H=5, W=2, off_scale=1.0),
    'wide_rows': dict(B=1, C=20, Co=16, H=10, W=130, off_scale=0.7),  # 64-pixel row tiles, ragged last column tile
    'wide_rows_smooth': dict(B=1, C=16, Co=16, H=6, W=96, off_scale=0.05),
    'stride2_wide': dict(B=1, C=16, Co=8, H=12, W=70, s=2),          # col2im window too large for the LDS -> windowless
}


# The data-gradient walk of the backward has two forms (csrc/dcn.hip): coord_grad + col2im as two kernels, and both
# roles as ONE launch (`dcn_bwd_data_kernel` on `dcn_prep_kernel`'s packed geometry records) -- the form the benched
# step runs on its 128 x 128 / 64 x 64 maps, selected by size.  Every value test below runs both: 'one_launch' lowers
# the size rule to one tile (cnuda_dcn_set_fused_min_tiles), so borders, odd widths, tiles narrower than 64 pixels and
# out-of-bounds samples (`cell = 0x80000000` records) go through that kernel too; the launch log says which ran.
WALKS = {'two_kernels': 2 ** 31 - 1, 'one_launch': 1}


class _walk:
    """forces one form of the data-gradient walk and checks, from the library's launch log, that it ran"""

    def __init__(self, walk, geoms):
        self.walk, self.geoms = walk, geoms          # geoms: [(H, W, k, s, p, d, dg)] of the DCN calls inside the block

    def __enter__(self):
        import hip_runtime as hr
        self.ctx = hr.dcn_fused_min_tiles(WALKS[self.walk])
        self.ctx.__enter__()
        self.log = hr.launch_log()
        self.log.__enter__()

    def __exit__(self, et, ev, tb):
        import dcn_plan
        import hip_runtime as hr
        self.log.__exit__(et, ev, tb)
        names = self.log.names
        self.ctx.__exit__(et, ev, tb)
        if et is None:
            ran = dcn_plan.backward_walk_kernels(names)
            can = [dcn_plan.one_launch_possible(*g) for g in self.geoms]
            if self.walk == 'one_launch' and all(can):
                assert ran == 'one_launch', (ran, names)
            elif self.walk == 'two_kernels' and all(g[1] >= 2 for g in self.geoms):
                assert ran == 'two_kernels', (ran, names)


def _geom_of(x, geom):
    return (x.shape[2], x.shape[3], geom[0], geom[2], geom[4], geom[6], geom[8])


@pytest.mark.parametrize('walk', sorted(WALKS))
@pytest.mark.parametrize('name', sorted(CASES))
def test_forward_backward_vs_oracle(name, walk):
    import _ext
    (x, w, b, off, m, go), geom = _case(zlib.crc32(name.encode()) % 1000, **CASES[name])
    want = od.dcn_v2_forward(x, w, b, off, m, *geom)
    wg = od.dcn_v2_backward(x, w, b, off, m, go, *geom)
    dx, dw_, db, doff, dm, dgo = [t.to(DEV) for t in (x, w, b, off, m, go)]
    out = _ext.dcn_v2_forward(dx, dw_, db, doff, dm, *geom)
    _close(out, want)
    with _walk(walk, [_geom_of(x, geom)]):
        grads = _ext.dcn_v2_backward(dx, dw_, db, doff, dm, dgo, *geom)
    for got, ref, nm in zip(grads, wg, ['input', 'offset', 'mask', 'weight', 'bias']):
        assert got.shape == ref.shape, nm
        _close(got, ref)


def test_the_one_launch_walk_is_reachable_for_the_expected_cases():
    """the cases above that cannot take the one-launch walk are exactly: width 1, and the two strided layers whose LDS
    window exceeds the budget (they run windowless two-kernel / generic paths); deformable groups are composed of
    deformable_group = 1 calls since round 6 and follow the same rule"""
    import dcn_plan
    cannot = sorted(n for n, c in CASES.items()
                    if not dcn_plan.one_launch_possible(c['H'], c['W'], c.get('k', 3), c.get('s', 1), c.get('p', 1),
                                                        c.get('d', 1), c.get('dg', 1)))
    assert cannot == ['stride2', 'stride2_wide', 'width1'], cannot


def test_zero_offset_identity_testcpu_32_67():
    import _ext
    x = torch.randn(2, 2, 4, 4)
    w = torch.zeros(2, 2, 3, 3)
    w[0, 0, 1, 1] = 1
    w[1, 1, 1, 1] = 1
    out = _ext.dcn_v2_forward(x.to(DEV), w.to(DEV), torch.zeros(2, device=DEV), torch.zeros(2, 18, 4, 4, device=DEV),
                              torch.full((2, 9, 4, 4), 0.5, device=DEV), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    assert (x - 2 * out.cpu()).abs().max().item() < 1e-10      # exact, the reference's own threshold


def test_autograd_module_path_and_argument_order():
    from libs.DCNv2.dcn_v2 import dcn_v2_conv
    (x, w, b, off, m, go), geom = _case(5, 2, 16, 8, 9, 9)
    leaves = [t.to(DEV).requires_grad_(True) for t in (x, off, m, w, b)]       # python order: input, offset, mask, weight, bias
    y = dcn_v2_conv(*leaves, 1, 1, 1, 1)
    y.backward(go.to(DEV))
    wg = od.dcn_v2_backward(x, w, b, off, m, go, *geom)                        # gi, goff, gmask, gw, gb
    for leaf, ref in zip(leaves, [wg[0], wg[1], wg[2], wg[3], wg[4]]):
        _close(leaf.grad, ref)


def test_full_size_linearity_64ch_128sq():
    # cfg-size layer (64->64 @128x128, B=4 to bound memory/time): DCN is linear in
    # (input) and in (weight, bias) -- a size-independent property
    import _ext
    (x, w, b, off, m, go), geom = _case(9, 4, 64, 64, 128, 128)
    x, w, b, off, m = [t.to(DEV) for t in (x, w, b, off, m)]
    x2 = torch.randn_like(x)
    zb = torch.zeros_like(b)
    f = lambda inp, bias: _ext.dcn_v2_forward(inp, w, bias, off, m, *geom)
    lhs = f(x + 2 * x2, b)
    rhs = f(x, b) + 2 * f(x2, zb)
    _close(lhs, rhs, 2e-5)


# Independent known answers for fractional offsets (closed forms / plain convolutions, no oracle involved):
# the same vectors that pin the oracle in tests/test_oracle_dcn.py, through the C ABI on the GPU.
def _gpu_fwd(x, w, b, off, m, *geom):
    import _ext
    return _ext.dcn_v2_forward(*[t.to(DEV) for t in (x, w, b, off, m)], *geom).cpu()


def _gpu_bwd(x, w, b, off, m, go, *geom):
    import _ext
    return [t.cpu() for t in _ext.dcn_v2_backward(*[t.to(DEV) for t in (x, w, b, off, m, go)], *geom)]


def _gpu_bwd_walk(walk):
    def run(x, w, b, off, m, go, *geom):
        with _walk(walk, [_geom_of(x, geom)]):
            return _gpu_bwd(x, w, b, off, m, go, *geom)
    return run


@pytest.mark.parametrize('dh,dw', [(0.5, 0.0), (0.0, 0.5), (0.5, 0.5), (-0.5, 0.5), (0.25, -0.75)])
@pytest.mark.parametrize('size', [(2, 3, 7, 9, 4), (2, 64, 32, 40, 64), (1, 128, 16, 16, 128)])
def test_known_answer_half_pixel_offsets_are_box_blurs(dh, dw, size):
    import dcn_known_answers as ka
    ka.check_uniform_fractional_offset_is_box_blur_then_conv(_gpu_fwd, dh, dw, torch.float32, 2e-5, size=size)


@pytest.mark.parametrize('walk', sorted(WALKS))
def test_known_answer_validity_window_open_at_minus_one_and_H(walk):
    import dcn_known_answers as ka
    ka.check_validity_window_is_open(_gpu_fwd, _gpu_bwd_walk(walk), torch.float32, 1e-6)


@pytest.mark.parametrize('walk', sorted(WALKS))
@pytest.mark.parametrize('size', [(2, 3, 12, 11, 2), (2, 64, 40, 36, 64), (1, 16, 20, 70, 32)])
def test_known_answer_linear_ramp_gradients(size, walk):
    import dcn_known_answers as ka
    ka.check_linear_ramp_has_constant_coordinate_gradient(_gpu_fwd, _gpu_bwd_walk(walk), torch.float32, 5e-5, size=size)


@pytest.mark.parametrize('walk', sorted(WALKS))
def test_known_answer_col2im_four_weights(walk):
    import dcn_known_answers as ka
    ka.check_col2im_scatters_the_four_bilinear_weights(_gpu_bwd_walk(walk), torch.float32, 1e-6)


def test_errors():
    import _ext
    z = lambda *s: torch.zeros(*s, device=DEV)
    with pytest.raises(RuntimeError):      # channel mismatch (dcn_v2_cuda.cu:83-84)
        _ext.dcn_v2_forward(z(1, 3, 4, 4), z(2, 4, 3, 3), z(2), z(1, 18, 4, 4), z(1, 9, 4, 4), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):      # kernel shape mismatch (:80-81)
        _ext.dcn_v2_forward(z(1, 4, 4, 4), z(2, 4, 3, 3), z(2), z(1, 18, 4, 4), z(1, 9, 4, 4), 5, 5, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):      # CPU tensors: no fallback
        _ext.dcn_v2_forward(torch.zeros(1, 4, 4, 4), torch.zeros(2, 4, 3, 3), torch.zeros(2),
                            torch.zeros(1, 18, 4, 4), torch.zeros(1, 9, 4, 4), 3, 3, 1, 1, 1, 1, 1, 1, 1)


@pytest.mark.parametrize('B,C,S,Co', [(32, 64, 128, 64), (32, 128, 64, 128)])
def test_dcn_layer_is_reproducible_at_full_size(B, C, S, Co):
    """Forward, weight gradient and offset-branch gradients of a full-size DCN layer are bit-identical from call to
    call; grad_input only differs by the order of col2im's straggler atomics (1e-7).  (It caught a 16-byte epilogue
    that stored through buffer descriptors and lost stores whenever the weight-gradient GEMM ran beside the
    column-gradient GEMM on the DCN backward's second stream: DESIGN.md section 10.)"""
    from libs.DCNv2.dcn_v2 import DCN
    torch.manual_seed(3)
    m = DCN(C, Co, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV)
    with torch.no_grad():
        m.conv_offset_mask.weight.normal_(0, 0.05)
        m.conv_offset_mask.bias.normal_(0, 0.3)
    x0 = torch.randn(B, C, S, S, device=DEV)
    g = torch.randn(B, Co, S, S, device=DEV)
    outs = []
    for _ in range(3):
        x = x0.clone().requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        y = m(x)
        y.backward(g)
        outs.append((y.detach().clone(), x.grad.clone(), m.weight.grad.clone(), m.conv_offset_mask.weight.grad.clone()))
    for o in outs[1:]:
        assert torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][2], o[2])
        for a, b in ((outs[0][1], o[1]), (outs[0][3], o[3])):
            assert (a - b).abs().max().item() <= 1e-5 * a.abs().max().item()


def test_literal_c_abi_entry_points():
    """The two symbols SURVEY 8b names -- `cnuda_dcn_v2_forward` / `cnuda_dcn_v2_backward`, the binding INTEGRATION.md
    section A shows (src/dcn_v2.h:10-54) -- called directly through ctypes with raw device pointers, no shim."""
    import ctypes
    import hip_runtime as hr
    (x, w, b, off, m, go), geom = _case(17, 2, 16, 8, 9, 11)
    want = od.dcn_v2_forward(x, w, b, off, m, *geom)
    wg = od.dcn_v2_backward(x, w, b, off, m, go, *geom)
    dx, dw_, db, doff, dm, dgo = [t.to(DEV) for t in (x, w, b, off, m, go)]
    L = ctypes.CDLL(hr.LIB_PATH)
    L.cnuda_dcn_v2_workspace_bytes.restype = ctypes.c_size_t
    L.cnuda_last_error.restype = ctypes.c_char_p
    B, C, H, W = x.shape
    dims = [ctypes.c_int(v) for v in (B, C, H, W, w.shape[0]) + geom]
    nbytes = L.cnuda_dcn_v2_workspace_bytes(*dims)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = torch.empty_like(want, device=DEV)
    rc = L.cnuda_dcn_v2_forward(P(dx), P(dw_), P(db), P(doff), P(dm), P(out), *dims, P(ws), ctypes.c_size_t(nbytes), st)
    assert rc == 0, L.cnuda_last_error()
    _close(out, want)
    grads = [torch.empty_like(t) for t in (dx, doff, dm, dw_, db)]
    rc = L.cnuda_dcn_v2_backward(P(dx), P(dw_), P(db), P(doff), P(dm), P(dgo), *[P(g) for g in grads], *dims, P(ws),
                                 ctypes.c_size_t(nbytes), st)
    assert rc == 0, L.cnuda_last_error()
    torch.cuda.synchronize()
    for got, ref in zip(grads, wg):
        _close(got, ref)
    # a workspace that is too small is refused, not overrun
    rc = L.cnuda_dcn_v2_forward(P(dx), P(dw_), P(db), P(doff), P(dm), P(out), *dims, P(ws), ctypes.c_size_t(16), st)
    assert rc != 0 and b'workspace' in L.cnuda_last_error()


@pytest.mark.parametrize('sigma,want', [(0.3, 0), (1.0, 1), (2.5, 3)], ids=['sigma0.3', 'sigma1', 'sigma2.5'])
def test_offset_census_picks_the_kernels_and_not_the_values(sigma, want):
    """libs/DCNv2/dcn_v2.py DCN: every CENSUS_EVERY training forwards the layer counts its own offsets beyond +-2 / +-3 px
    (cnuda_dcn_offset_census) and tells the library which regime its next calls run in (cnuda_dcn_set_offset_regime): a
    wide-margin window for the data-gradient walk, the gathering loader for the forward.  Speed only -- output and all
    gradients equal those of regime 0 to summation order, at every offset scale."""
    import hip_runtime as hr
    from libs.DCNv2.dcn_v2 import DCN
    torch.manual_seed(11)
    m = DCN(32, 32, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV).train()
    x = torch.randn(2, 32, 32, 32, device=DEV)
    with torch.no_grad():
        m.conv_offset_mask.weight.normal_(0, 1.0)
        std = m.conv_offset_mask(x)[:, :18].std().item()
        m.conv_offset_mask.weight[:18] *= sigma / std
        m.conv_offset_mask.bias.zero_()
    g = torch.randn(2, 32, 32, 32, device=DEV)

    def run(force_regime):
        m.zero_grad()
        xs = x.clone().requires_grad_(True)
        if force_regime is None:
            m._census_calls = 0                         # census at this forward
        else:
            m._census_calls, m._regime = 1, force_regime  # (not a census call)
        with hr.launch_log() as log:
            y = m(xs)
            y.backward(g)
        return y.detach(), [xs.grad, m.weight.grad.clone(), m.conv_offset_mask.weight.grad.clone()], set(log.names), m._regime

    y0, g0, k0, _ = run(0)
    y1, g1, k1, regime = run(None)
    assert regime == want, (regime, want)
    assert any('dcn_offset_census_kernel' in n for n in k1) and not any('dcn_offset_census_kernel' in n for n in k0)
    assert any('dcnw_fwd_kernel' in n for n in k0)
    assert any('dcnw_fwd_kernel' in n for n in k1) == (not (want & 2)), k1
    assert hr.lib().cnuda_dcn_set_offset_regime(0) == 0          # every call leaves the process-wide setting at 0
    scale = max(1.0, y0.abs().max().item())
    assert (y1 - y0).abs().max().item() <= 1e-5 * scale
    for a, b in zip(g1, g0):
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize('B,C,Co,S', [(8, 64, 64, 64), (16, 64, 64, 128), (4, 128, 64, 32), (8, 512, 256, 16), (2, 32, 32, 24)],
                         ids=['halo_tile_offsets', 'halo_tile_256_offsets', 'split_k4_offsets', 'split_k_offsets', 'small_odd'])
def test_offsets_and_mask_read_out_of_the_offset_convolutions_output(B, C, Co, S):
    """Round 6: `DCN.forward` no longer materialises offset and mask tensors -- the offset convolution's epilogue applies the
    mask's sigmoid (cnuda_conv2d_forward_rowsig: halo-tile, im2col and split-K instances), the deformable convolution reads
    both out of that one tensor (cnuda_dcn_v2_forward_om) and its backward writes one gradient tensor with the mask's part
    already multiplied by m (1 - m) (cnuda_dcn_v2_backward_om).  Same expressions as the split kernels it replaces
    (libs/DCNv2/dcn_v2.py:118-122 of the reference): everything is bit-identical to the split form except grad_input, whose
    straggler atomics may arrive in another order, and no split kernel is launched."""
    import hip_runtime as hr
    from libs.DCNv2 import dcn_v2
    from test_zz_kernel_coverage import short
    torch.manual_seed(3 + C)
    m = dcn_v2.DCN(C, Co, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV)
    with torch.no_grad():
        m.conv_offset_mask.weight.normal_(0, 0.5 / (9 * C) ** 0.5)
        m.conv_offset_mask.bias.normal_(0, 0.3)
    m.train()
    x = torch.randn(B, C, S, S, device=DEV)
    gy = torch.randn(B, Co, S, S, device=DEV)
    res = {}
    for use_om in (True, False):
        prev, dcn_v2.USE_OM = dcn_v2.USE_OM, use_om
        try:
            for p in m.parameters():
                p.grad = None
            xx = x.clone().requires_grad_(True)
            with hr.launch_log() as log:
                y = m(xx)
                y.backward(gy)
            names = sorted(short(n) for n in log.names)
            assert any('split_offset_mask' in n for n in names) == (not use_om), names
            res[use_om] = [y.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in m.parameters()]
        finally:
            dcn_v2.USE_OM = prev
    for i, (a, b) in enumerate(zip(res[True], res[False])):
        if i == 1:
            assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()
        else:
            assert torch.equal(a, b), i


def test_a_forward_hook_on_the_offset_convolution_still_fires():
    """`DCN.forward` calls the offset convolution's kernel directly (row-sigmoid epilogue) -- unless somebody hooked the
    `conv_offset_mask` module (bench.measure_dcn_offsets does; so may a user's probe): then the module is CALLED, its output
    split like the reference does, and the layer's result is the same."""
    from libs.DCNv2 import dcn_v2
    torch.manual_seed(9)
    m = dcn_v2.DCN(32, 32, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV)
    with torch.no_grad():
        m.conv_offset_mask.weight.normal_(0, 0.05)
    x = torch.randn(2, 32, 16, 16, device=DEV)
    with torch.no_grad():
        want = m(x)
        seen = []
        h = m.conv_offset_mask.register_forward_hook(lambda mod, inp, out: seen.append(tuple(out.shape)))
        got = m(x)
        h.remove()
        again = m(x)
    assert seen == [(2, 27, 16, 16)]
    assert torch.equal(got, want) and torch.equal(again, want)


@pytest.mark.parametrize('name', ['dg2', 'dg2_64ch', 'dg4_odd'])
def test_deformable_groups_through_the_autograd_path_with_saved_columns(name):
    """deformable_group > 1 (round 6: composed of deformable_group = 1 calls per group) through the product's autograd
    Function, which keeps the sampled columns of every group between forward and backward: values against the C oracle."""
    from libs.DCNv2.dcn_v2 import dcn_v2_conv
    (x, w, b, off, m, go), geom = _case(zlib.crc32(name.encode()) % 1000, **CASES[name])
    want = od.dcn_v2_forward(x, w, b, off, m, *geom)
    wg = od.dcn_v2_backward(x, w, b, off, m, go, *geom)           # input, offset, mask, weight, bias
    leaves = [t.to(DEV).requires_grad_(True) for t in (x, off, m, w, b)]
    kh, kw, sh, sw, ph, pw, dh, dw, dg = geom
    y = dcn_v2_conv(*leaves, (sh, sw), (ph, pw), (dh, dw), dg)
    assert y.grad_fn.saved_tensors[5] is not None                 # the columns were kept
    _close(y, want)
    y.backward(go.to(DEV))
    for got, ref, nm in zip([leaves[0].grad, leaves[1].grad, leaves[2].grad, leaves[3].grad, leaves[4].grad], wg,
                            ['input', 'offset', 'mask', 'weight', 'bias']):
        assert got.shape == ref.shape, nm
        _close(got, ref)
