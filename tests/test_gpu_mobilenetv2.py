"""MI355X parity of the MobileNetV2 backend: the kernels it adds (depthwise 3x3 convolution,
BN + ReLU6) against CPU torch, and the network (with DCN up-sampling stages / with skip connections) against
golden vectors from the reference's CenterMobileNetV2 class."""
import ast

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import inputs as gin
from test_gpu_resnet import GRAD_FLOOR, _checksums, _close, _close_calibrated

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
CASES = {'dcn': (dict(use_dcn=True, use_skip=False), 2, 64, 95), 'skip': (dict(use_dcn=False, use_skip=True), 2, 96, 96)}


@pytest.mark.parametrize('B,C,H,W,k,s', [(2, 32, 16, 16, 3, 1), (2, 96, 17, 13, 3, 2), (1, 960, 4, 4, 3, 1),
                                         (3, 5, 9, 7, 3, 2), (2, 8, 12, 12, 5, 1), (1, 16, 2, 2, 3, 2)])
def test_depthwise_conv2d_matches_torch(B, C, H, W, k, s):
    from hip_runtime import ops
    rs = np.random.RandomState(C * 7 + H)
    x = T(rs.standard_normal((B, C, H, W)).astype(np.float32))
    w = T((rs.standard_normal((C, 1, k, k)) / k).astype(np.float32))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, s, (k - 1) // 2, 1, C)
    gy = T(rs.standard_normal(tuple(yr.shape)).astype(np.float32))
    yr.backward(gy)
    xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = ops.depthwise_conv2d(xg, wg, s, (k - 1) // 2)
    assert y.shape == yr.shape
    y.backward(gy.to(DEV))
    _close(y.detach().cpu().numpy(), yr.detach().numpy())
    _close(xg.grad.cpu().numpy(), xr.grad.numpy())
    _close(wg.grad.cpu().numpy(), wr.grad.numpy())


@pytest.mark.parametrize('res', [False, True])
def test_batch_norm_relu6_matches_torch(res):
    from hip_runtime import ops
    g = torch.Generator().manual_seed(9)
    shape = (2, 12, 9, 10)
    x = (torch.randn(shape, generator=g) * 4 + 2).requires_grad_(True)              # plenty of values beyond 6
    gamma = (1 + 0.2 * torch.randn(12, generator=g)).requires_grad_(True)
    beta = (1.5 + torch.randn(12, generator=g)).requires_grad_(True)
    r = torch.randn(shape, generator=g).requires_grad_(True) if res else None
    rm, rv = torch.zeros(12), torch.ones(12)
    y = F.batch_norm(x, rm.clone(), rv.clone(), gamma, beta, True, 0.1, 1e-5)
    y = F.relu6(y + r if res else y)
    assert (y == 6).any() and (y == 0).any()
    gy = torch.randn(shape, generator=g)
    y.backward(gy)
    lx, lg, lb = [t.detach().to(DEV).requires_grad_(True) for t in (x, gamma, beta)]
    lr = r.detach().to(DEV).requires_grad_(True) if res else None
    dy = ops.batch_norm_act(lx, lg, lb, rm.to(DEV), rv.to(DEV), True, 0.1, 1e-5, lr, 6)
    _close(dy.detach().cpu().numpy(), y.detach().numpy())
    dy.backward(gy.to(DEV))
    for a, b in ((lx, x), (lg, gamma), (lb, beta)) + (((lr, r),) if res else ()):
        _close(a.grad.cpu().numpy(), b.grad.numpy())
    with pytest.raises(ValueError):
        ops.batch_norm_act(lx, lg, lb, rm.to(DEV), rv.to(DEV), True, 0.1, 1e-5, None, 3)


@pytest.mark.parametrize('tag', sorted(CASES))
def test_mobilenetv2_forward_backward_golden(golden, tag):
    from backends import mobilenetv2
    flags, B, S, seed = CASES[tag]
    g = golden('mbv2_' + tag)
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    model = mobilenetv2.build(num_classes=6, pretrained=False, **flags)
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes).items()})
    model = model.to(DEV)
    x = T(gin.image_batch(B, S, S, seed)).to(DEV)
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
    # the scalar sums ~2e4 outputs whose fp32 noise is 1e-3 each (the per-output checks above): the reference's own
    # single fp32 draw (|ref32 - ref64| = 1.9e-3) can sit well below the spread of such a sum -- measured here
    # 5.7e-3 with the f32 MFMA and 1.9e-2 in the split-operand mode, whose outputs are no further from the fp64
    # values than the f32 ones -- hence 16x like the gradient checksums
    _close_calibrated(scalar.item(), g['scalar'], g['f64_scalar'], floor=2e-4, k=16.0, what='scalar')
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
            # base.18.1 sees 8 samples per channel (2x2 maps, B=2) after 52 layers: statistics at north_star's 1e-4
            _close(sd[n + '.running_mean'].cpu().numpy(), g[key], 1e-4)
            _close(sd[n + '.running_var'].cpu().numpy(), g['rv__' + n], 1e-4)
            assert int(sd[n + '.num_batches_tracked']) == int(g['nbt__' + n])


def test_skip_with_dcn_fails_like_the_reference():
    """mobilenetv2.py:96-107 adds the H/16 skip to the H/32 output of the first DCN when both flags are set."""
    from backends import mobilenetv2
    m = mobilenetv2.build(num_classes=2, pretrained=False, use_dcn=True, use_skip=True).to(DEV)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64, device=DEV))
