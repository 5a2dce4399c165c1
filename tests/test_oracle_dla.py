"""Pins oracle/dla.py (functional DLA-34 restatement) against outputs of the
reference module (tests/golden/dla_*.npz, step_*.npz)."""
import ast

import numpy as np
import pytest
import torch

import inputs as gin
from oracle import dla as odla
from oracle import losses as ol

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _checksums(t):
    t = t.detach().double().reshape(-1)
    idx = torch.arange(t.numel(), dtype=torch.float64)
    return np.array([t.sum().item(), t.abs().sum().item(), (t * torch.cos(0.01 * idx)).sum().item()])


def _state(g, requires_grad=True, offset_gain=1.0):
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    st = {}
    for k, v in gin.fill_state(shapes, offset_gain).items():
        t = T(v).clone()
        if requires_grad and t.is_floating_point() and 'running_' not in k:
            t.requires_grad_(True)
        st[k] = t
    return st


def _close(a, b, tol=2e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(1.0, np.abs(b).max())
    assert np.abs(a - b).max() <= tol * scale, (np.abs(a - b).max(), scale)


@pytest.mark.parametrize('tag,rotated,B,S,seed', [('axis', False, 2, 64, 41), ('rot', True, 2, 96, 42)])
def test_dla_forward_backward(golden, tag, rotated, B, S, seed):
    g = golden('dla_' + tag)
    heads = [str(h) for h in g['head_order']]
    assert heads == ['hm', 'wh', 'reg']
    x = T(gin.image_batch(B, S, S, seed))
    st = _state(g, requires_grad=False)
    with torch.no_grad():
        out = odla.forward(st, x, heads, training=False)
    for k in heads:
        _close(out[k].numpy(), g['eval_' + k])
    st = _state(g)
    out = odla.forward(st, x, heads, training=True)
    for k in heads:
        _close(out[k].detach().numpy(), g['train_' + k])
    scalar = sum((out[k] * torch.cos(torch.arange(out[k].numel(), dtype=torch.float32)
                                     .reshape(out[k].shape) * 0.1)).sum() for k in heads)
    scalar.backward()
    assert abs(scalar.item() - float(g['scalar'])) <= 2e-4 * max(1, abs(float(g['scalar'])))
    for key in g.files:
        if key.startswith('gradsum__'):
            n = key[len('gradsum__'):]
            got, want = _checksums(st[n].grad), g[key]
            assert np.abs(got - want).max() <= 5e-4 * max(1.0, want[1]), (n, got, want)
    none = sorted(k for k, v in st.items() if v.requires_grad and v.grad is None)
    assert none == sorted(str(s) for s in g['grad_none'])
    for key in g.files:
        if key.startswith('rm__'):
            n = key[4:]
            _close(st[n + '.running_mean'].numpy(), g[key], 1e-5)
            _close(st[n + '.running_var'].numpy(), g['rv__' + n], 1e-5)
            assert int(st[n + '.num_batches_tracked']) == int(g['nbt__' + n])
    # the same restatement in float64 reproduces the reference module evaluated in float64
    st64 = {k: (v.detach().double() if v.is_floating_point() else v.clone()) for k, v in _state(g, False).items()}
    with torch.no_grad():
        out64 = odla.forward(st64, x.double(), heads, training=True)
    for k in heads:
        _close(out64[k].numpy(), g['f64_train_' + k], 1e-9)


@pytest.mark.parametrize('tag,weight', [('entropy', 1e-4), ('maxsq', 0.3)])
def test_uda_step(golden, tag, weight):
    g = golden('step_' + tag)
    gd = golden('dla_axis')
    st = _state(gd)
    B, S, M = 2, 64, 8
    batch = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (3, 2), 2, 51).items()}
    params = [v for k, v in st.items() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=5e-5, weight_decay=1e-4)
    opt.zero_grad()
    out_s = odla.forward(st, T(gin.image_batch(B, S, S, 52)), training=True)
    out_t = odla.forward(st, T(gin.image_batch(B, S, S, 53)), training=True)
    c_loss, stats, prob = ol.detection_loss(out_s, batch, 1.0, 0.1, 1.0, 1.0, False)
    u = (ol.entropy_loss if tag == 'entropy' else ol.max_square_loss)(out_t['hm']) * weight
    c_loss.backward()
    u.backward()
    opt.step()
    key = 'entropy_loss' if tag == 'entropy' else 'max_square_loss'
    assert abs(u.item() - float(g['stat_' + key])) <= 1e-5 * max(1e-6, abs(float(g['stat_' + key])))  # Q4: weighted value is logged
    for k, v in stats.items():
        assert abs(v.item() - float(g['stat_' + k])) <= 2e-4 * max(1, abs(float(g['stat_' + k]))), k
    assert abs((c_loss + u).item() - float(g['stat_total_loss'])) <= 2e-4 * max(1, abs(float(g['stat_total_loss'])))
    _close(prob.detach().numpy(), g['src_hm_after'], 1e-4)
    for fk in g.files:
        if fk.startswith('gradsum__'):
            n = fk[len('gradsum__'):]
            got, want = _checksums(st[n].grad), g[fk]
            assert np.abs(got - want).max() <= 1e-3 * max(1.0, want[1]), (n, got, want)
            gotp, wantp = _checksums(st[n]), g['param__' + n]
            assert np.abs(gotp - wantp).max() <= 1e-5 * max(1.0, wantp[1]), (n, gotp, wantp)
    _close(st['base.base_layer.1.running_mean'].numpy(), g['rm__base.base_layer.1'], 1e-5)
    _close(st['base.base_layer.1.running_var'].numpy(), g['rv__base.base_layer.1'], 1e-5)
    assert int(st['base.base_layer.1.num_batches_tracked']) == 2      # Q6: two BN updates per step


def test_base_step_dla_configs1(golden):
    """S1 with the DLA backend (configs[1]): the oracle pieces in `uda.base.Model`'s order against the imported
    reference class (tests/golden/step_base128.npz: evaluation sequence, then one training step)."""
    from oracle import decode as odec
    g = golden('step_base128')
    B, S, M, n_obj, K = 4, 128, 16, (5, 1, 9, 3), 40
    st = _state(g, offset_gain=0.1)

    def batch(seed):
        d = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, n_obj, 2, seed).items()}
        d['input'] = T(gin.image_batch(B, S, S, seed + 1))
        return d
    ev = batch(93)
    with torch.no_grad():
        out = odla.forward(st, ev['input'], training=False)
        loss, stats, prob = ol.detection_loss(out, ev, 1.0, 0.1, 1.0, 1.0, False)
    for k, v in dict(stats, total_loss=loss).items():
        assert abs(v.item() - float(g['eval_stat_' + k])) <= 1e-5 * max(1.0, abs(float(g['eval_stat_' + k]))), k
    _close(prob.numpy(), g['eval_hm'], 1e-5)
    _close(out['wh'].numpy(), g['eval_wh'], 1e-5)
    _close(out['reg'].numpy(), g['eval_reg'], 1e-5)
    dets = odec.decode_detection(g['eval_hm'], g['eval_wh'], g['eval_reg'], K=K)
    dets[:, :, :4] *= 4
    np.testing.assert_array_equal(dets[:, :, :4], g['det_pred_boxes'])
    np.testing.assert_array_equal(dets[:, :, 4], g['det_pred_scores'])
    np.testing.assert_array_equal(dets[:, :, 5].astype(np.int32), g['det_pred_classes'])
    data = batch(91)
    opt = torch.optim.Adam([v for v in st.values() if v.requires_grad], lr=5e-5, weight_decay=1e-4)
    opt.zero_grad()
    out = odla.forward(st, data['input'], training=True)
    loss, stats, prob = ol.detection_loss(out, data, 1.0, 0.1, 1.0, 1.0, False)
    loss.backward()
    opt.step()
    assert [str(k) for k in g['stat_keys']] == ['centernet_loss', 'hm_loss', 'wh_loss', 'off_loss', 'total_loss']
    for k, v in dict(stats, total_loss=loss).items():
        assert abs(v.item() - float(g['stat_' + k])) <= 1e-4 * max(1.0, abs(float(g['stat_' + k]))), k
    _close(prob.detach().numpy(), g['train_hm'], 1e-4)
    _close(out['wh'].detach().numpy(), g['train_wh'], 1e-4)
    _close(out['reg'].detach().numpy(), g['train_reg'], 1e-4)
    for fk in g.files:
        if fk.startswith('gradsum__'):
            n = fk[len('gradsum__'):]
            got, want = _checksums(st[n].grad), g[fk]
            assert np.abs(got - want).max() <= 1e-3 * max(1.0, want[1]), (n, got, want)
            gotp, wantp = _checksums(st[n]), g['param__' + n]
            assert np.abs(gotp - wantp).max() <= 1e-5 * max(1.0, wantp[1]), (n, gotp, wantp)
    for fk in g.files:
        if fk.startswith('rm__'):
            n = fk[4:]
            _close(st[n + '.running_mean'].numpy(), g[fk], 1e-5)
            _close(st[n + '.running_var'].numpy(), g['rv__' + n], 1e-5)
            assert int(st[n + '.num_batches_tracked']) == int(g['nbt__' + n]) == 1


def test_advent_step(golden):
    """S4 (uda/adversarial_entropy_minimization.py:77-152) on the oracle: rotated model, periodic angle loss,
    discriminator on entropy maps, the sigmoid-source quirk (Q1) and five backward calls."""
    import torch.nn.functional as F
    g = golden('step_advent')
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    dshapes = dict(ast.literal_eval(str(g['dshapes_json'])))
    st = {}
    for k, v in gin.fill_state(shapes).items():
        t = T(v).clone()
        if t.is_floating_point() and 'running_' not in k:
            t.requires_grad_(True)
        st[k] = t
    dp = {k: T(gin.fill_value('discriminator.' + k, tuple(v))).clone().requires_grad_(True) for k, v in dshapes.items()}

    def D(x):
        for i in (0, 2, 4, 6):
            x = F.leaky_relu(F.conv2d(x, dp['%d.weight' % i], dp['%d.bias' % i], 2, 1), 0.2)
        return F.conv2d(x, dp['8.weight'], dp['8.bias'], 2, 1)

    B, S, M, C = 2, 128, 8, 6
    batch = {k: T(v) for k, v in gin.detection_batch(B, C, S // 4, S // 4, M, (4, 2), 3, 71).items()}
    opt = torch.optim.Adam([v for v in st.values() if v.requires_grad], lr=5e-5, weight_decay=1e-4)
    dopt = torch.optim.Adam(list(dp.values()), lr=1e-3, weight_decay=1e-4)
    heads = ('hm', 'wh', 'reg')
    out_s = odla.forward(st, T(gin.image_batch(B, S, S, 72)), heads, training=True)
    out_t = odla.forward(st, T(gin.image_batch(B, S, S, 73)), heads, training=True)
    for v in dp.values():
        v.requires_grad_(False)
    fool = D(ol.entropy_map(out_t['hm']))
    loss, stats, prob = ol.detection_loss(out_s, batch, 1.0, 0.1, 1.0, 1.0, True)
    loss.backward()
    dtf = ol.advent_loss(fool, 0) * 1e-4
    dtf.backward()
    for v in dp.values():
        v.requires_grad_(True)
    ds = ol.advent_loss(D(ol.entropy_map(prob.detach())), 0) / 2.0          # Q1: probabilities, not logits
    ds.backward()
    dt = ol.advent_loss(D(ol.entropy_map(out_t['hm'].detach())), 1) / 2.0
    dt.backward()
    opt.step(); dopt.step()
    got = dict(stats)
    got.update(total_loss=loss + ds + dt + dtf, dis_soruce=ds, dis_target=dt, dis_fool=dtf)
    for k, v in got.items():
        want = float(g['stat_' + k])
        assert abs(float(v) - want) <= 2e-4 * max(1e-3, abs(want)), (k, float(v), want)
    for n, v in dp.items():
        gs, ws = _checksums(v.grad), g['dgradsum__' + n]
        assert np.abs(gs - ws).max() <= 2e-3 * max(1e-6, ws[1]), (n, gs, ws)
        ps, wp = _checksums(v), g['dparam__' + n]
        assert np.abs(ps - wp).max() <= 1e-4 * max(1.0, wp[1]), (n, ps, wp)
