"""MI355X: randomised geometry sweeps of the convolution / transposed-convolution / depthwise / DCN entry points
against CPU torch (the primitives the oracle is built from).  Seeds are fixed: the cases are the same on every
run; they exist to catch tile-edge and padding-class mistakes that the hand-picked shapes miss.  1e-4 of scale."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _close(a, b, tol=1e-4, what=''):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, b.abs().max().item())
    assert (a - b).abs().max().item() <= tol * scale, (what, (a - b).abs().max().item(), scale)


def _conv_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        k = int(rs.choice([1, 1, 3, 3, 3, 4, 5, 7]))
        s = int(rs.choice([1, 1, 1, 2, 2, 3]))
        p = int(rs.randint(0, k // 2 + 1))
        B, C, Co = int(rs.randint(1, 4)), int(rs.choice([1, 3, 5, 16, 17, 32, 48, 64, 70, 130])), int(rs.choice([1, 2, 6, 16, 27, 33, 64, 96, 140]))
        H, W = int(rs.randint(1, 23)), int(rs.randint(1, 23))
        if (H + 2 * p - k) // s + 1 < 1 or (W + 2 * p - k) // s + 1 < 1 or H + 2 * p < k or W + 2 * p < k:
            continue
        out.append((B, C, H, W, Co, k, s, p, bool(rs.randint(2)), float(rs.choice([-1.0, -1.0, 0.0, 0.2]))))
    return out


@pytest.mark.parametrize('case', _conv_cases(48, 1234), ids=lambda c: 'B%dC%dH%dW%dCo%dk%ds%dp%d%s%s' % (c[:8] + ('b' if c[8] else '', 'a' if c[9] >= 0 else '')))
@pytest.mark.parametrize('mode', [0, 1])
def test_conv2d_random_geometry(case, mode):
    import hip_runtime as hr
    from hip_runtime import ops
    B, C, H, W, Co, k, s, p, bias, act = case
    g = torch.Generator().manual_seed(hash(case) % 100000)
    x = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True) if bias else None
    y = F.conv2d(x, w, b, s, p)
    if act >= 0:
        y = F.leaky_relu(y, act)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    before = hr.get_matrix_mode()
    hr.set_matrix_mode(mode)
    try:
        dx, dw = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
        db = b.detach().to(DEV).requires_grad_(True) if bias else None
        dy = ops.conv2d(dx, dw, db, s, p, act)
        dy.backward(gy.to(DEV))
    finally:
        hr.set_matrix_mode(before)
    _close(dy, y, what='y')
    _close(dx.grad, x.grad, what='gx')
    _close(dw.grad, w.grad, what='gw')
    if bias:
        _close(db.grad, b.grad, what='gb')


def _convt_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        k, s = int(rs.choice([2, 3, 4, 4])), int(rs.choice([1, 2, 2]))
        p = int(rs.randint(0, (k - 1) // 2 + 1))
        op = int(rs.randint(0, s))
        B, Ci, Co = int(rs.randint(1, 3)), int(rs.choice([3, 16, 20, 64, 130])), int(rs.choice([2, 16, 24, 64, 96]))
        H, W = int(rs.randint(1, 12)), int(rs.randint(1, 12))
        if (H - 1) * s - 2 * p + k + op < 1 or (W - 1) * s - 2 * p + k + op < 1:
            continue
        out.append((B, Ci, Co, H, W, k, s, p, op))
    return out


@pytest.mark.parametrize('case', _convt_cases(16, 77), ids=lambda c: 'B%dCi%dCo%dH%dW%dk%ds%dp%dop%d' % c)
def test_conv_transpose2d_random_geometry(case):
    from hip_runtime import ops
    B, Ci, Co, H, W, k, s, p, op = case
    g = torch.Generator().manual_seed(hash(case) % 100000)
    x = torch.randn(B, Ci, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Ci, Co, k, k, generator=g) / (Ci * k * k / (s * s)) ** 0.5).requires_grad_(True)
    y = F.conv_transpose2d(x, w, None, s, p, op)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dx, dw = x.detach().to(DEV).requires_grad_(True), w.detach().to(DEV).requires_grad_(True)
    dy = ops.conv_transpose2d(dx, dw, s, p, op)
    dy.backward(gy.to(DEV))
    _close(dy, y, what='y')
    _close(dx.grad, x.grad, what='gx')
    _close(dw.grad, w.grad, what='gw')


def _dcn_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        B, C, Co = int(rs.randint(1, 3)), int(rs.choice([2, 8, 16, 24, 64, 80])), int(rs.choice([2, 4, 16, 27, 64, 72]))
        H, W = int(rs.randint(1, 15)), int(rs.randint(1, 15))
        s = int(rs.choice([1, 1, 2]))
        dg = int(rs.choice([1, 1, 1, 2])) if C % 2 == 0 else 1
        out.append((B, C, Co, H, W, s, dg, float(rs.choice([0.3, 1.0, 3.0]))))
    return out


@pytest.mark.parametrize('walk', ['two_kernels', 'one_launch'])      # the two forms of the data-gradient walk (test_gpu_dcn.py)
@pytest.mark.parametrize('case', _dcn_cases(20, 4321), ids=lambda c: 'B%dC%dCo%dH%dW%ds%ddg%dsig%g' % c)
def test_dcn_random_geometry_vs_oracle(case, walk):
    import _ext
    from test_gpu_dcn import _walk
    from oracle import dcn as od
    od.build()
    B, C, Co, H, W, s, dg, sigma = case
    rs = np.random.RandomState(hash(case) % 100000)
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    if Ho < 1 or Wo < 1:
        pytest.skip('empty output')
    x = T(rs.standard_normal((B, C, H, W)).astype(np.float32))
    w = T((rs.standard_normal((Co, C, 3, 3)) / (3 * C ** 0.5)).astype(np.float32))
    b = T(rs.standard_normal(Co).astype(np.float32))
    off = T((rs.standard_normal((B, 18 * dg, Ho, Wo)) * sigma).astype(np.float32))
    m = T(rs.uniform(0, 1, (B, 9 * dg, Ho, Wo)).astype(np.float32))
    gy = T(rs.standard_normal((B, Co, Ho, Wo)).astype(np.float32))
    geom = (3, 3, s, s, 1, 1, 1, 1, dg)
    want_y = od.dcn_v2_forward(x, w, b, off, m, *geom)
    want = od.dcn_v2_backward(x, w, b, off, m, gy, *geom)
    d = [t.to(DEV) for t in (x, w, b, off, m)]
    got_y = _ext.dcn_v2_forward(*d, *geom)
    with _walk(walk, [(H, W, 3, s, 1, 1, dg)]):
        got = _ext.dcn_v2_backward(*d, gy.to(DEV), *geom)
    _close(got_y, want_y, what='y')
    for name, a, r in zip(('gx', 'goff', 'gmask', 'gw', 'gb'), got, want):
        _close(a, r, what=name)


def _bn_cases(n, seed):
    rs = np.random.RandomState(seed)
    return [(int(rs.randint(1, 5)), int(rs.choice([1, 3, 16, 27, 64, 130])), int(rs.randint(1, 20)), int(rs.randint(1, 20)),
             [False, True, 6][int(rs.randint(3))], bool(rs.randint(2))) for _ in range(n)]


@pytest.mark.parametrize('case', _bn_cases(24, 99), ids=lambda c: 'B%dC%dH%dW%dact%sres%d' % (c[0], c[1], c[2], c[3], c[4], c[5]))
def test_batch_norm_random_geometry(case):
    from hip_runtime import ops
    B, C, H, W, act, res = case
    if B * H * W < 2:
        pytest.skip('one value per channel (torch raises too)')
    g = torch.Generator().manual_seed(hash(case) % 100000)
    x = (torch.randn(B, C, H, W, generator=g) * 3 + 1).requires_grad_(True)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (1 + torch.randn(C, generator=g)).requires_grad_(True)
    r = torch.randn(B, C, H, W, generator=g).requires_grad_(True) if res else None
    rm, rv = torch.randn(C, generator=g) * 0.1, 1 + 0.3 * torch.rand(C, generator=g)
    rm_d, rv_d = rm.clone().to(DEV), rv.clone().to(DEV)
    y = F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-5)
    y = y + r if res else y
    y = F.relu6(y) if act == 6 else (F.relu(y) if act else y)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    lx, lg, lb = [t.detach().to(DEV).requires_grad_(True) for t in (x, gamma, beta)]
    lr = r.detach().to(DEV).requires_grad_(True) if res else None
    dy = ops.batch_norm_act(lx, lg, lb, rm_d, rv_d, True, 0.1, 1e-5, lr, act)
    dy.backward(gy.to(DEV))
    _close(dy, y, what='y')
    _close(lx.grad, x.grad, what='gx')
    _close(lg.grad, gamma.grad, what='ggamma')
    _close(lb.grad, beta.grad, what='gbeta')
    if res:
        _close(lr.grad, r.grad, what='gres')
    _close(rm_d, rm, 1e-5, 'running_mean')
    _close(rv_d, rv, 1e-5, 'running_var')


def _decode_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        B, C, H, W = int(rs.randint(1, 4)), int(rs.choice([1, 2, 6, 20, 80])), int(rs.randint(2, 70)), int(rs.randint(2, 70))
        K = int(rs.randint(1, min(H * W, 200) + 1))
        out.append((B, C, H, W, K, bool(rs.randint(2)), bool(rs.randint(2)), int(rs.randint(1 << 30))))
    return out


@pytest.mark.parametrize('case', _decode_cases(24, 555), ids=lambda c: 'B%dC%dH%dW%dK%drot%dreg%d' % c[:7])
def test_decode_random_geometry_vs_oracle(case):
    from backends import decode as hd
    from oracle import decode as od
    B, C, H, W, K, rotated, with_reg, seed = case
    rs = np.random.RandomState(seed)
    heat = np.unique(rs.uniform(1e-4, 1 - 1e-4, 4 * B * C * H * W).astype(np.float32))      # tie-free scores
    assert heat.size >= B * C * H * W
    heat = rs.permutation(heat)[:B * C * H * W].reshape(B, C, H, W)
    wh = rs.uniform(1, 50, (B, 3 if rotated else 2, H, W)).astype(np.float32)
    reg = rs.uniform(0, 1, (B, 2, H, W)).astype(np.float32) if with_reg else None
    want, winds, wcls = od.decode_detection(heat, wh, reg, K=K, rotated=rotated, return_inds=True)
    dev = lambda a: None if a is None else T(a).to(DEV)
    dets, inds = hd._run(dev(heat), dev(wh), dev(reg), K, rotated, 3)
    assert np.array_equal(inds.cpu().numpy(), winds)
    cc = 6 if rotated else 5
    assert np.array_equal(dets[..., cc].cpu().numpy().astype(np.int32), wcls)
    assert np.array_equal(dets[..., cc - 1].cpu().numpy(), want[..., cc - 1])
    np.testing.assert_allclose(dets.cpu().numpy(), want, rtol=1e-5, atol=5e-5)


def _cat_cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    while len(out) < n:
        ns = int(rs.randint(2, 5))
        cs = tuple(int(rs.choice([64, 64, 64, 128, 128, 192, 256])) for _ in range(ns))
        # (mostly rows of a multiple of four pixels: the library declines the rest -- a quarter of the cases keeps any width)
        B, H = int(rs.randint(1, 9)), int(rs.randint(1, 17))
        W = int(rs.randint(1, 13)) if rs.randint(4) == 0 else 4 * int(rs.randint(1, 9))
        Co = int(rs.choice([16, 27, 64, 100, 128, 192, 256]))
        out.append((B, H, W, cs, Co))
    return out


@pytest.mark.parametrize('case', _cat_cases(32, 77), ids=lambda c: 'B%dH%dW%d_%s_Co%d' % (c[0], c[1], c[2], 'x'.join(map(str, c[3])), c[4]))
def test_conv1x1_cat_random_geometry(case):
    """ops.conv1x1_cat (DLA's Root without the concatenation) on random source lists and map sizes: where the library takes the
    sources, output, input gradients and weight gradient equal conv2d(cat(xs)) BIT FOR BIT (the same GEMMs) and are torch's CPU
    values; where it declines (a plan that cuts K, pixels per image no multiple of four), it says so with None."""
    import ctypes
    import hip_runtime as hr
    from hip_runtime import ops
    B, H, W, cs, Co = case
    g = torch.Generator().manual_seed(hash(case) % 100000)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cs]
    w = torch.randn(Co, sum(cs), 1, 1, generator=g) / sum(cs) ** 0.5
    gy = torch.randn(B, Co, H, W, generator=g)
    supported = bool(hr.lib().cnuda_conv2d_cat_supported((ctypes.c_int * len(cs))(*cs), len(cs), B, H, W, Co))
    dx = [t.to(DEV).requires_grad_(True) for t in xs]
    dw = w.to(DEV).requires_grad_(True)
    y = ops.conv1x1_cat(dx, dw)
    assert (y is not None) == supported
    if (H * W) % 4:
        assert not supported
    if not supported:
        return
    y.backward(gy.to(DEV))
    px = [t.to(DEV).requires_grad_(True) for t in xs]
    pw = w.to(DEV).requires_grad_(True)
    py = ops.conv2d(ops.cat_channels(px), pw, None, 1, 0)
    py.backward(gy.to(DEV))
    assert torch.equal(y, py) and torch.equal(dw.grad, pw.grad)
    for a, b in zip(dx, px):
        assert torch.equal(a.grad, b.grad)
    rx = [t.clone().requires_grad_(True) for t in xs]
    rw = w.clone().requires_grad_(True)
    ry = F.conv2d(torch.cat(rx, 1), rw)
    ry.backward(gy)
    _close(y, ry, what='y')
    _close(dw.grad, rw.grad, what='gw')
    for a, r in zip(dx, rx):
        _close(a.grad, r.grad, what='gx')
