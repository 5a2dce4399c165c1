"""MI355X parity of the ResNet backend (configs[0]): the two kernels it adds (full
transposed convolution, windowed max pool) against CPU torch, the network against golden vectors from the
reference's CenterResNet class, and `uda.base.Model.step` at configs[0]'s real size."""
import ast

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import inputs as gin

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _checksums(t):
    t = t.detach().double().reshape(-1).cpu()
    idx = torch.arange(t.numel(), dtype=torch.float64)
    return np.array([t.sum().item(), t.abs().sum().item(), (t * torch.cos(0.01 * idx)).sum().item()])


def _close(a, b, tol=1e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(1.0, np.abs(b).max())
    assert np.abs(a - b).max() <= tol * scale, (np.abs(a - b).max(), scale)


def _close_calibrated(got, ref32, ref64, floor=1e-4, k=8.0, what=''):
    """see tests/test_gpu_dla.py::_close_calibrated: as close to the fp64 value as the reference's own fp32
    result is, within a factor k, never tighter than north_star's 1e-4."""
    got, ref32, ref64 = [np.asarray(t, np.float64) for t in (got, ref32, ref64)]
    scale = max(1.0, np.abs(ref64).max())
    noise = np.abs(ref32 - ref64).max()
    err = np.abs(got - ref64).max()
    assert err <= max(floor * scale, k * noise), (what, err, noise, scale)


@pytest.mark.parametrize('B,Ci,Co,H,W,k,s,p,op', [
    (2, 32, 16, 5, 7, 4, 2, 1, 0),        # CenterResNet's stage shape class
    (1, 512, 256, 8, 8, 4, 2, 1, 0),      # first stage at configs[0]'s size
    (2, 16, 32, 6, 5, 3, 2, 1, 1),        # _get_deconv_cfg kernel 3 (output_padding 1)
    (2, 16, 16, 4, 4, 2, 2, 0, 0),        # kernel 2
    (1, 20, 24, 5, 5, 3, 1, 1, 0),        # stride 1, channels not a multiple of 16
])
def test_conv_transpose2d_matches_torch(B, Ci, Co, H, W, k, s, p, op):
    from hip_runtime import ops
    rs = np.random.RandomState(B * 100 + Ci)
    x = T(rs.standard_normal((B, Ci, H, W)).astype(np.float32))
    w = T((rs.standard_normal((Ci, Co, k, k)) / np.sqrt(Ci * k * k / (s * s))).astype(np.float32))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, None, s, p, op)
    gy = T(rs.standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(gy)
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = ops.conv_transpose2d(xg, wg, s, p, op)
    assert y.shape == yr.shape
    y.backward(gy.to(DEV))
    _close(y.detach().cpu().numpy(), yr.detach().numpy())
    _close(xg.grad.cpu().numpy(), xr.grad.numpy())
    _close(wg.grad.cpu().numpy(), wr.grad.numpy())


@pytest.mark.parametrize('B,C,H,W,k,s,p', [(2, 5, 16, 16, 3, 2, 1), (1, 3, 9, 11, 3, 2, 1), (2, 4, 8, 8, 3, 1, 1),
                                           (1, 2, 7, 7, 2, 1, 0), (2, 64, 128, 128, 3, 2, 1)])
def test_max_pool_window_matches_torch(B, C, H, W, k, s, p):
    from hip_runtime import ops
    rs = np.random.RandomState(H * 10 + k)
    x = T(rs.standard_normal((B, C, H, W)).astype(np.float32))
    x[0, 0, :4, :4] = 1.5            # ties: the first maximum in scan order takes the gradient
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, k, s, p)
    gy = T(rs.standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(gy)
    xg = x.to(DEV).requires_grad_(True)
    y = ops.max_pool2d(xg, k, s, p)
    y.backward(gy.to(DEV))
    assert torch.equal(y.detach().cpu(), yr.detach())
    _close(xg.grad.cpu().numpy(), xr.grad.numpy(), 1e-6)


# End-to-end gradient checksums: ReLU masks are discontinuous, and among the ~10^5 pre-activations of a layer
# one or two lie within fp32 rounding of zero.  Whether such an element flips depends on the summation order
# (measured on MI355X for this fixture: one flipped element in `hm.0` and one in `base.5.1.bn1` relative to the
# CPU result; each shifts its own layer's gradient by ~0.5 % and everything upstream by ~0.1-0.3 %, while the
# forward outputs and losses stay within the calibrated 1e-4 class).  The reference's fp32-vs-fp64 distance
# cannot calibrate a rare event, so the whole-network gradient checks use a 5e-3 floor on the L1 checksum;
# every layer's backward is held to 1e-4 on its own in test_gpu_ops.py / the op tests above.
GRAD_FLOOR = 5e-3


def _model(g):
    from backends import resnet
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    m = resnet.build(18, num_classes=6, pretrained=False)
    m.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes).items()})
    return m.to(DEV)


def test_resnet18_forward_backward_golden(golden):
    g = golden('resnet18_fwd')
    model = _model(g)
    x = T(gin.image_batch(2, 128, 128, 61)).to(DEV)
    model.eval()
    with torch.no_grad():
        out = model(x)
    assert list(out) == ['hm', 'wh', 'reg']
    for k in out:
        _close_calibrated(out[k].cpu().numpy(), g['eval_' + k], g['f64_eval_' + k], what='eval ' + k)
    model.train()
    out = model(x)
    for k in out:
        _close_calibrated(out[k].detach().cpu().numpy(), g['train_' + k], g['f64_train_' + k], what=k)
    scalar = sum((out[k] * torch.cos(torch.arange(out[k].numel(), dtype=torch.float32)
                                     .reshape(out[k].shape) * 0.1).to(DEV)).sum() for k in out)
    scalar.backward()
    _close_calibrated(scalar.item(), g['scalar'], g['f64_scalar'], floor=2e-4, what='scalar')
    params = dict(model.named_parameters())
    for key in g.files:
        if key.startswith('gradsum__'):
            n = key[len('gradsum__'):]
            got, w32, w64 = _checksums(params[n].grad), g[key], g['f64_' + key]
            noise = np.abs(w32 - w64).max()
            assert np.abs(got - w64).max() <= max(GRAD_FLOOR * max(1.0, w64[1]), 16 * noise), (n, got, w64, noise)
    sd = model.state_dict()
    for key in g.files:
        if key.startswith('rm__'):
            n = key[4:]
            _close(sd[n + '.running_mean'].cpu().numpy(), g[key], 1e-5)
            _close(sd[n + '.running_var'].cpu().numpy(), g['rv__' + n], 1e-5)
            assert int(sd[n + '.num_batches_tracked']) == int(g['nbt__' + n])


def test_model_step_resnet18_config0(golden):
    """S1: uda.base.Model.step (uda/base.py:31-56) with the ResNet-18 backend, B=2, 256x256, Adam lr 5e-5."""
    from uda.base import Model
    from hip_runtime import optim
    from losses.centernet import DetectionLoss
    g = golden('resnet18_step')
    model = _model(golden('resnet18_fwd'))
    plugin = Model()
    plugin.backend = model
    plugin.device = torch.device(DEV)
    plugin.optimizer = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0,
                                          periodic=False)
    plugin.init_done()
    plugin.to(DEV)
    plugin.set_phase(True)
    B, S, M = 2, 256, 16
    data = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (5, 3), 2, 71).items()}
    data['input'] = T(gin.image_batch(B, S, S, 72))
    out = plugin.step(data)
    stats = out['stats']
    assert sorted(stats) == ['centernet_loss', 'hm_loss', 'off_loss', 'total_loss', 'wh_loss']
    for k in stats:
        assert not stats[k].is_cuda and not stats[k].requires_grad
        _close_calibrated(stats[k].item(), g['stat_' + k], g['f64_stat_' + k], floor=1e-4, what=k)
    _close(out['source_domain']['hm'].detach().cpu().numpy()[:, :, ::4, ::4], g['hm_after'], 1e-4)     # Q1
    params = dict(model.named_parameters())
    for key in g.files:
        if key.startswith('gradsum__'):
            n = key[len('gradsum__'):]
            got, w32, w64 = _checksums(params[n].grad), g[key], g['f64_' + key]
            noise = np.abs(w32 - w64).max()
            assert np.abs(got - w64).max() <= max(GRAD_FLOOR * max(1.0, w64[1]), 16 * noise), (n, got, w64, noise)
        if key.startswith('param__'):
            n = key[len('param__'):]
            # Adam's first step moves every element by lr * sign(g): elements whose gradient rounds to the other
            # sign differ by 2 * lr; bounded by lr * numel on the L1 checksum
            got, want = _checksums(params[n]), g[key]
            assert abs(got[1] - want[1]) <= 1e-4 * max(1.0, want[1]) + 2 * 5e-5 * 0.02 * params[n].numel(), n
