"""CPU: the functional ResNet oracle (oracle/resnet.py) against golden vectors produced by the reference's
CenterResNet class (tests/golden/resnet18_*.npz; trunk = restated torchvision 0.6, see the oracle's header)."""
import ast

import numpy as np
import torch

import inputs as gin
from oracle import losses as oracle_losses
from oracle import resnet as oracle_resnet

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _checksums(t):
    t = t.detach().double().reshape(-1)
    idx = torch.arange(t.numel(), dtype=torch.float64)
    return np.array([t.sum().item(), t.abs().sum().item(), (t * torch.cos(0.01 * idx)).sum().item()])


def _state(g):
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    state = {k: T(v) for k, v in gin.fill_state(shapes).items()}
    params = [str(n) for n in g['param_names']]
    for n in params:
        state[n].requires_grad_(True)
    return state, params


def _close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (np.abs(a - b).max(), np.abs(b).max())


def test_resnet18_forward_backward_golden(golden):
    g = golden('resnet18_fwd')
    state, params = _state(g)
    x = T(gin.image_batch(2, 128, 128, 61))
    with torch.no_grad():
        ev = oracle_resnet.forward({k: v.clone() for k, v in state.items()}, x, training=False)
    assert list(ev) == [str(h) for h in g['head_order']] == ['hm', 'wh', 'reg']
    for k in ev:
        _close(ev[k].numpy(), g['eval_' + k], 1e-4)          # tolerance of north_star (fp32 heatmaps/heads)
    out = oracle_resnet.forward(state, x, training=True)
    for k in out:
        _close(out[k].detach().numpy(), g['train_' + k], 1e-4)
    scalar = sum((out[k] * torch.cos(torch.arange(out[k].numel(), dtype=torch.float32)
                                     .reshape(out[k].shape) * 0.1)).sum() for k in out)
    scalar.backward()
    _close(scalar.item(), g['scalar'], 1e-4)
    for key in g.files:
        if key.startswith('gradsum__'):
            n = key[len('gradsum__'):]
            w = g[key]
            assert np.abs(_checksums(state[n].grad) - w).max() <= 2e-4 * max(1.0, w[1]), n
        if key.startswith('rm__'):
            n = key[4:]
            _close(state[n + '.running_mean'].numpy(), g[key], 1e-5)
            _close(state[n + '.running_var'].numpy(), g['rv__' + n], 1e-5)
            assert int(state[n + '.num_batches_tracked']) == int(g['nbt__' + n]) == 1


def test_resnet18_step_golden(golden):
    """Model.step (uda/base.py:31-56) at configs[0]'s size: B=2, 256x256, Adam lr 5e-5."""
    g = golden('resnet18_step')
    state, params = _state(golden('resnet18_fwd'))
    B, S, M = 2, 256, 16
    batch = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (5, 3), 2, 71).items()}
    x = T(gin.image_batch(B, S, S, 72))
    opt = torch.optim.Adam([state[n] for n in params], lr=5e-5)
    opt.zero_grad()
    out = oracle_resnet.forward(state, x, training=True)
    loss, stats, prob = oracle_losses.detection_loss(out, batch, hm_weight=1.0, wh_weight=0.1, off_weight=1.0)
    loss.backward()
    opt.step()
    stats['total_loss'] = loss
    for k, v in stats.items():
        want = float(g['stat_' + k])
        assert abs(float(v) - want) <= 2e-4 * max(1e-3, abs(want)), (k, float(v), want)
    for key in g.files:
        if key.startswith('gradsum__'):
            n = key[len('gradsum__'):]
            w = g[key]
            assert np.abs(_checksums(state[n].grad) - w).max() <= 5e-4 * max(1.0, w[1]), n
    _close(prob.detach().numpy()[:, :, ::4, ::4], g['hm_after'], 1e-4)        # Q1: probabilities
