"""CPU: the functional MobileNetV2 oracle (oracle/mobilenetv2.py) against golden vectors produced by the
reference's CenterMobileNetV2 class (tests/golden/mbv2_*.npz; trunk = restated torchvision 0.6, DCN = the CPU
DCN oracle -- see the oracle's header)."""
import ast

import numpy as np
import pytest
import torch

import inputs as gin
from oracle import dcn as oracle_dcn
from oracle import mobilenetv2 as oracle_mb

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
CASES = {'dcn': (dict(use_dcn=True, use_skip=False), 2, 64, 95), 'skip': (dict(use_dcn=False, use_skip=True), 2, 96, 96)}


def _checksums(t):
    t = t.detach().double().reshape(-1)
    idx = torch.arange(t.numel(), dtype=torch.float64)
    return np.array([t.sum().item(), t.abs().sum().item(), (t * torch.cos(0.01 * idx)).sum().item()])


def _close(a, b, tol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max()), (np.abs(a - b).max(), np.abs(b).max())


@pytest.mark.parametrize('tag', sorted(CASES))
def test_mobilenetv2_forward_backward_golden(golden, tag):
    oracle_dcn.build()
    flags, B, S, seed = CASES[tag]
    g = golden('mbv2_' + tag)
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    state = {k: T(v) for k, v in gin.fill_state(shapes).items()}
    for n in (str(n) for n in g['param_names']):
        state[n].requires_grad_(True)
    x = T(gin.image_batch(B, S, S, seed))
    with torch.no_grad():
        ev = oracle_mb.forward({k: v.clone() for k, v in state.items()}, x, training=False, **flags)
    assert list(ev) == [str(h) for h in g['head_order']] == ['hm', 'wh', 'reg']
    for k in ev:
        assert ev[k].shape == (B, {'hm': 6, 'wh': 2, 'reg': 2}[k], S // 4, S // 4)
        _close(ev[k].numpy(), g['eval_' + k], 1e-4)
    out = oracle_mb.forward(state, x, training=True, **flags)
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
