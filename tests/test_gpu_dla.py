"""MI355X parity of the DLA-34 backend and of the UDA step plugins against golden
vectors produced by the reference module (tests/golden/dla_*.npz, step_*.npz)."""
import ast

import numpy as np
import pytest
import torch

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


def _close_rel(got, ref, tol=1e-4, what=''):
    """north_star's bar, taken literally: |got - reference| <= 1e-4 x the largest magnitude of the reference
    tensor (no floor of 1: a loss of 1e-4 is held to 1e-8)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    print('%-28s err %.2e  scale %.2e  err/scale %.2e' % (what, err, scale, err / max(scale, 1e-300)))
    assert err <= tol * scale, (what, err, scale)


def _close_calibrated(got, ref32, ref64, floor=1e-4, k=8.0, what=''):
    """A 50-layer fp32 network in train-mode BatchNorm carries rounding noise well
    above 1e-4 whatever the summation order: the reference's own fp32 result differs
    from the same module evaluated in fp64 by `noise` (1e-3 on the heads, 1e-2 on
    gradients for these tiny inputs).  The HIP result must sit as close to the exact
    (fp64) value as the reference does, within a factor k, and never needs to beat
    the 1e-4 floor of north_star.  k = 8: the fp32 MFMA is one k-ordered fma chain
    per output, which measures 3-5x the per-convolution rounding noise of oneDNN's
    blocked accumulation on the CPU (numerics); every single layer is
    held to 1e-4 against the oracle in test_gpu_ops.py / test_gpu_dcn.py."""
    got, ref32, ref64 = [np.asarray(t, np.float64) for t in (got, ref32, ref64)]
    scale = np.abs(ref64).max()                 # relative to the tensor's own magnitude, also for tiny statistics
    noise = np.abs(ref32 - ref64).max()
    err = np.abs(got - ref64).max()
    print('%-28s err/scale %.2e  reference fp32-vs-fp64 noise/scale %.2e  err/noise %.2f'
          % (what, err / max(scale, 1e-300), noise / max(scale, 1e-300), err / max(noise, 1e-300)))
    assert err <= max(floor * scale, k * noise), (what, err, noise, scale)


def _check_full_grads(g, named, prefix='grad__'):
    """Whole gradient tensors of the step fixtures (make_golden.FULL_GRADS), element by element: 1e-4 of the tensor's
    largest magnitude -- or, where the reference's own float32 run is further than that from its float64 run, ten
    times that distance (printed).  Measured on the MI355X (round 5): the two head output layers sit at 2e-6 / 8e-6 of
    their maximum (reference noise 3e-6 / 7e-6); for the three tensors deep in the network the REFERENCE's float32 run
    is 0.8-1.3e-2 of the maximum away from its float64 run element by element (train-mode BatchNorm over 64..1024
    samples per channel: the gradients are differences of nearly equal sums), the HIP result 2.5-5.2e-2 (2.3-6.2 x:
    the k-ordered f32 MFMA chain carries 3-5 x the rounding noise of oneDNN's blocked sums).
    Three moments of a tensor (the gradsum__ keys) let a sparse defect through -- a dropped 16-byte store, one
    mis-addressed tile is an O(1) error of an element; this does not."""
    seen = 0
    for fk in g.files:
        if not fk.startswith(prefix):
            continue
        n = fk[len(prefix):]
        got = named[n].grad.detach().cpu().numpy().astype(np.float64)
        w32, w64 = g[fk].astype(np.float64), g['f64_' + fk]
        assert got.shape == w64.shape, (n, got.shape, w64.shape)
        scale, noise, err = np.abs(w64).max(), np.abs(w32 - w64).max(), np.abs(got - w64).max()
        print('%-52s %7d elements  err/max %.2e  reference noise/max %.2e  worst element %d'
              % (n, got.size, err / max(scale, 1e-300), noise / max(scale, 1e-300), int(np.abs(got - w64).argmax())))
        assert err <= max(1e-4 * scale, 10 * noise), (n, err, scale, noise)
        seen += 1
    assert seen >= (1 if prefix != 'grad__' else 5), (prefix, seen)


def _model(g, rotated=False):
    from backends import dla
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    m = dla.build(num_classes=6, rotated_boxes=rotated)
    m.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes).items()})
    return m.to(DEV)


@pytest.mark.parametrize('tag,rotated,B,S,seed', [('axis', False, 2, 64, 41), ('rot', True, 2, 96, 42)])
def test_dla_forward_backward_golden(golden, tag, rotated, B, S, seed):
    g = golden('dla_' + tag)
    model = _model(g, rotated)
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
    _close_calibrated(scalar.item(), g['scalar'], g['f64_scalar'], floor=2e-4, what='scalar')
    params = dict(model.named_parameters())
    for key in g.files:
        if key.startswith('gradsum__'):
            n = key[len('gradsum__'):]
            got, w32, w64 = _checksums(params[n].grad), g[key], g['f64_' + key]
            noise = np.abs(w32 - w64).max()
            assert np.abs(got - w64).max() <= max(5e-4 * max(1.0, w64[1]), 16 * noise), (n, got, w64, noise)
    none = sorted(n for n, p in params.items() if p.grad is None)
    assert none == sorted(str(s) for s in g['grad_none'])           # discarded project branches get no gradient
    sd = model.state_dict()
    for key in g.files:
        if key.startswith('rm__'):
            n = key[4:]
            _close(sd[n + '.running_mean'].cpu().numpy(), g[key], 1e-5)
            _close(sd[n + '.running_var'].cpu().numpy(), g['rv__' + n], 1e-5)
            assert int(sd[n + '.num_batches_tracked']) == int(g['nbt__' + n])


class _Cfg(dict):
    __getattr__ = dict.__getitem__


@pytest.mark.parametrize('tag,weight', [('entropy', 1e-4), ('maxsq', 0.3)])
def test_uda_step_golden(golden, tag, weight):
    import uda
    from hip_runtime import optim
    from losses.centernet import DetectionLoss
    g = golden('step_' + tag)
    model = _model(golden('dla_axis'))
    plugin = uda.EntropyMinimization(weight) if tag == 'entropy' else uda.MaxSquaresMinimization(weight)
    plugin.backend = model
    plugin.device = torch.device(DEV)
    plugin.optimizer = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, weight_decay=1e-4)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0,
                                          periodic=False)
    plugin.init_done()
    plugin.to(DEV)
    plugin.set_phase(True)
    B, S, M = 2, 64, 8
    data = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (3, 2), 2, 51).items()}
    data['input'] = T(gin.image_batch(B, S, S, 52))
    data['target_domain_input'] = T(gin.image_batch(B, S, S, 53))
    out = plugin.step(data)
    stats = out['stats']
    for k in stats:
        assert not stats[k].is_cuda and not stats[k].requires_grad
        _close_calibrated(stats[k].item(), g['stat_' + k], g['f64_stat_' + k], floor=1e-4, what=k)
    _close_calibrated(out['source_domain']['hm'].detach().cpu().numpy(), g['src_hm_after'],
                      g['f64_src_hm_after'], what='hm prob')                            # Q1
    params = dict(model.named_parameters())
    for fk in g.files:
        if fk.startswith('gradsum__'):
            n = fk[len('gradsum__'):]
            got, want, w64 = _checksums(params[n].grad), g[fk], g['f64_' + fk]
            noise = np.abs(want - w64).max()
            assert np.abs(got - w64).max() <= max(1e-3 * max(1.0, w64[1]), 16 * noise), (n, got, w64, noise)
            gotp, wantp = _checksums(params[n]), g['param__' + n]
            # Adam's first step moves every element by lr*sign(g): elements whose tiny gradient
            # changes sign under rounding noise move by 2*lr -- allow 5 % of them
            flips = 0.05 * params[n].numel() * 2 * 5e-5
            assert np.abs(gotp - wantp).max() <= 1e-5 * max(1.0, wantp[1]) + flips, (n, gotp, wantp)
    # discarded project branches: no gradient -> Adam (weight decay!) must leave them untouched
    shapes = dict(ast.literal_eval(str(golden('dla_axis')['shapes_json'])))
    n = 'base.level3.project.0.weight'
    assert torch.equal(params[n].detach().cpu(), T(gin.fill_value(n, shapes[n])))
    sd = model.state_dict()
    _close(sd['base.base_layer.1.running_mean'].cpu().numpy(), g['rm__base.base_layer.1'], 1e-5)
    _close(sd['base.base_layer.1.running_var'].cpu().numpy(), g['rv__base.base_layer.1'], 1e-5)
    assert int(sd['base.base_layer.1.num_batches_tracked']) == 2                    # Q6
    # evaluation path: no_grad step + get_detections
    plugin.set_phase(False)
    plugin.cfg = _Cfg(max_detections=20, model=_Cfg(backend=_Cfg(params=_Cfg(rotated_boxes=False))))
    ev = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (3, 2), 2, 51).items()}
    ev['input'] = T(gin.image_batch(B, S, S, 52))
    ev['target_domain_input'] = T(gin.image_batch(B, S, S, 53))
    ev['id'] = torch.arange(B)
    ev['gt_dets'] = torch.rand(B, M, 6)
    ev['gt_areas'] = torch.rand(B, M)
    with torch.no_grad():
        o = plugin.step(ev, is_training=False)
    dets = plugin.get_detections(o, ev)
    assert dets['pred_boxes'].shape == (B, 20, 4) and dets['pred_scores'].shape == (B, 20)
    assert [len(b) for b in dets['gt_boxes']] == [3, 2]


def _plugin_base(model, K=40, rotated=False):
    import uda.base as ub
    from hip_runtime import optim
    from losses.centernet import DetectionLoss
    plugin = ub.Model()
    plugin.cfg = _Cfg(max_detections=K, model=_Cfg(backend=_Cfg(params=_Cfg(rotated_boxes=rotated))))
    plugin.backend = model
    plugin.device = torch.device(DEV)
    plugin.optimizer = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, weight_decay=1e-4)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0,
                                          periodic=False)
    plugin.init_done()
    plugin.to(DEV)
    return plugin


def test_base_step_dla_configs1_plain_1e4(golden):
    """configs[1] (DLA-34 + DCNv2, no UDA) through `uda.base.Model`: the evaluation sequence and one training step
    against the imported reference class (tests/golden/step_base128.npz, B = 4 at 128 x 128, DCN offsets of
    +-0.1 px: a fixture on which the reference's own float32 and float64 runs agree to 3e-6 (eval) / 2e-5 (train),
    so north_star's plain 1e-4 is asserted with no calibration)."""
    from backends import dla
    from detections_check import compare_detections
    g = golden('step_base128')
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    model = dla.build(num_classes=6)
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, 0.1).items()})
    plugin = _plugin_base(model.to(DEV))
    B, S, M, n_obj = 4, 128, 16, (5, 1, 9, 3)

    def batch(seed):
        d = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, n_obj, 2, seed).items()}
        d['input'] = T(gin.image_batch(B, S, S, seed + 1))
        return d
    # -- evaluation (train.py:204-223) ------------------------------------------------------------
    plugin.set_phase(False)
    ev = batch(93)
    ev.update({k: T(v) for k, v in gin.eval_extras(B, M, False, 95).items()})
    with torch.no_grad():
        o = plugin.step(ev, is_training=False)
    for k in ('centernet_loss', 'hm_loss', 'wh_loss', 'off_loss', 'total_loss'):
        _close_rel(o['stats'][k].item(), g['eval_stat_' + k], what='eval ' + k)
    for k in ('hm', 'wh', 'reg'):
        _close_rel(o['source_domain'][k].cpu().numpy(), g['eval_' + k], what='eval ' + k)
    # get_detections on the reference's own head outputs: everything bit-equal (indices, classes, scores, boxes)
    ref_out = {'source_domain': {k: T(g['eval_' + k]).to(DEV) for k in ('hm', 'wh', 'reg')}}
    ev2 = {k: v.clone() for k, v in ev.items()}
    compare_detections(plugin.get_detections(ref_out, ev2), g, 'det_')
    # ... and on the product's own outputs: scores within 1e-4; where the reference's ranking is decided by more
    # than that, the same detection in the same slot
    dets = plugin.get_detections(o, ev)
    ws, wb, wc = g['det_pred_scores'], g['det_pred_boxes'], g['det_pred_classes']
    assert np.abs(dets['pred_scores'] - ws).max() <= 1e-4
    gap = np.minimum(np.abs(np.diff(ws, axis=1, prepend=2.0)), np.abs(np.diff(ws, axis=1, append=-1.0)))
    sure = gap > 2e-4
    assert sure.mean() > 0.5
    assert np.array_equal(dets['pred_classes'][sure], wc[sure])
    assert np.abs(dets['pred_boxes'][sure] - wb[sure]).max() <= 1e-4 * np.abs(wb).max()
    compare_detections(dets, g, 'det_', exact_order=False)                 # ground-truth side: exact
    # -- one training step (uda/base.py:31-56) -----------------------------------------------------
    plugin.set_phase(True)
    data = batch(91)
    out = plugin.step(data)
    assert list(out['stats']) == [str(k) for k in g['stat_keys']]
    for k, v in out['stats'].items():
        assert not v.is_cuda and not v.requires_grad
        _close_rel(v.item(), g['stat_' + k], what='train ' + k)
    for k in ('hm', 'wh', 'reg'):
        _close_rel(out['source_domain'][k].detach().cpu().numpy(), g['train_' + k], what='train ' + k)   # hm: Q1
    np.testing.assert_array_equal(data['wh'].cpu().numpy(), g['wh_target_after'])                        # Q2
    params = dict(model.named_parameters())
    for fk in g.files:
        if fk.startswith('gradsum__'):
            n = fk[len('gradsum__'):]
            got, w32, w64 = _checksums(params[n].grad), g[fk], g['f64_' + fk]
            noise = np.abs(w32 - w64).max()
            err = np.abs(got - w64).max()
            print('%-58s err/|g|_1 %.2e  reference noise/|g|_1 %.2e' % (n, err / max(w64[1], 1e-300), noise / max(w64[1], 1e-300)))
            # gradient check sums: 1e-4 of the gradient's l1 mass where the reference's own float32 run gets that
            # close to float64, else 8x the reference's own distance
            assert err <= max(1e-4 * w64[1], 8 * noise), (n, got, w64, noise)
            gotp, wantp = _checksums(params[n]), g['param__' + n]
            flips = 0.05 * params[n].numel() * 2 * 5e-5                     # Adam sign flips of ~zero gradients
            assert np.abs(gotp - wantp).max() <= 1e-5 * max(1.0, wantp[1]) + flips, (n, gotp, wantp)
    _check_full_grads(g, params)
    sd = model.state_dict()
    for fk in g.files:
        if fk.startswith('rm__'):
            n = fk[4:]
            _close_rel(sd[n + '.running_mean'].cpu().numpy(), g[fk], 1e-4, what='running_mean ' + n)
            _close_rel(sd[n + '.running_var'].cpu().numpy(), g['rv__' + n], 1e-4, what='running_var ' + n)
            assert int(sd[n + '.num_batches_tracked']) == 1


@pytest.mark.parametrize('name', sorted(gin.GETDET_CASES))
def test_get_detections_golden(golden, name):
    """P4: `Model.get_detections` (uda/base.py:73-139) with the HIP decode against the dict of the imported
    reference class: x down_ratio, reg_mask row selection, rotated column split, keypoint branch."""
    import types
    import uda.base as ub
    from detections_check import compare_detections
    g = golden('getdet_' + name)
    src, batch, K, rotated = gin.getdet_inputs(name)
    m = ub.Model()
    m.cfg = _Cfg(max_detections=K, model=_Cfg(backend=_Cfg(params=_Cfg(rotated_boxes=rotated))))
    m.backend = types.SimpleNamespace(down_ratio=4)
    dets = m.get_detections({'source_domain': {k: T(v).to(DEV) for k, v in src.items()}},
                            {k: T(v).to(DEV) for k, v in batch.items()})
    compare_detections(dets, g)


def test_advent_step_golden(golden):
    """cfg5 semantics on the GPU plugin: rotated DLA-34, periodic angle loss, ADVENT discriminator
    (uda/adversarial_entropy_minimization.py:77-152), five backward calls, two fused-Adam optimizers."""
    import uda
    from hip_runtime import optim
    from losses.centernet import DetectionLoss
    g = golden('step_advent')
    from backends import dla
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    dshapes = dict(ast.literal_eval(str(g['dshapes_json'])))
    model = dla.build(num_classes=6, rotated_boxes=True)
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes).items()})
    plugin = uda.AdversarialEntropyMinimization(
        1e-4, optimizer=_Cfg(name='Adam', params=_Cfg(lr=1e-3, weight_decay=1e-4)))
    plugin.cfg = _Cfg(max_detections=20, model=_Cfg(backend=_Cfg(params=_Cfg(rotated_boxes=True, num_classes=6))))
    plugin.backend = model
    plugin.device = torch.device(DEV)
    plugin.optimizer = optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-4)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0,
                                          periodic=True)
    plugin.init_done()
    assert sorted(plugin.discriminator.state_dict()) == sorted(dshapes)         # checkpoint keys 0,2,4,6,8
    plugin.discriminator.load_state_dict(
        {k: T(gin.fill_value('discriminator.' + k, tuple(v))) for k, v in dshapes.items()})
    plugin.to(DEV)
    plugin.set_phase(True)
    B, S, M = 2, 128, 8
    data = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (4, 2), 3, 71).items()}
    data['input'] = T(gin.image_batch(B, S, S, 72))
    data['target_domain_input'] = T(gin.image_batch(B, S, S, 73))
    out = plugin.step(data)
    stats = out['stats']
    assert set(stats) == {'centernet_loss', 'hm_loss', 'wh_loss', 'off_loss', 'total_loss', 'dis_soruce',
                          'dis_target', 'dis_fool'}
    for k in stats:
        _close_calibrated(stats[k].item(), g['stat_' + k], g['f64_stat_' + k], floor=1e-4, what=k)
    _close_calibrated(out['source_domain']['hm'].detach().cpu().numpy(), g['src_hm_after'], g['f64_src_hm_after'],
                      what='hm prob')
    np.testing.assert_allclose(data['wh'].cpu().numpy(), g['wh_target_after'], rtol=1e-6, atol=1e-6)   # Q2
    dparams = dict(plugin.discriminator.named_parameters())
    for n in dshapes:
        got, w32, w64 = _checksums(dparams[n].grad), g['dgradsum__' + n], g['f64_dgradsum__' + n]
        noise = np.abs(w32 - w64).max()
        # the discriminator sees entropy maps that are allowed to differ by 1e-4 (fp32 budget of the backbone);
        # LeakyReLU sign flips turn that into a few 1e-3 of its gradient sums
        assert np.abs(got - w64).max() <= max(5e-3 * max(1e-6, w64[1]), 16 * noise), (n, got, w64, noise)
    for p in plugin.discriminator.parameters():
        assert p.requires_grad                                      # unfrozen again after the generator phase
    assert out['source_generator'].shape == (B, 1, 1, 1)


UDA128 = {'entropy': ('EntropyMinimization', (1e-4,), False, 111),
          'maxsq': ('MaxSquaresMinimization', (0.3,), False, 121),
          'advent': ('AdversarialEntropyMinimization', (1e-4,), True, 131)}


@pytest.mark.parametrize('batch_domains', [True, False], ids=['batched', 'sequential'])
@pytest.mark.parametrize('tag', sorted(UDA128))
def test_uda_step128_plain_1e4(golden, tag, batch_domains):
    """S2 (the benchmarked step), S3, S4 at north_star's plain tolerance: one training step of the product's plugin
    against the reference's OWN plugin class (uda/entropy_minimization.py:11-43, uda/max_squares_minimization.py:22-50,
    uda/adversarial_entropy_minimization.py:77-152; tests/golden/step_<tag>128.npz, B = 4 + 4 at 128 x 128, DCN offsets
    of +-0.1 px -- the reference's own float32 and float64 runs agree to ~2e-5 there).  Stats, both domains' head
    outputs and the running statistics: |got - ref| <= 1e-4 |ref|max, no calibration; gradient check sums:
    max(1e-4 of the gradient's l1 mass, 8 x the reference's own float32-vs-float64 distance).  Once through the
    default batched two-domain path, once with the reference's literal call sequence (batch_domains = False)."""
    import uda
    from backends import dla
    from hip_runtime import optim
    from losses.centernet import DetectionLoss
    g = golden('step_%s128' % tag)
    cls, args, rotated, seed = UDA128[tag]
    B, S, M, C, n_obj = 4, 128, 16, 6, (5, 1, 9, 3)
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    model = dla.build(num_classes=C, rotated_boxes=rotated)
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, 0.1).items()})
    plugin = getattr(uda, cls)(*args)
    plugin.batch_domains = batch_domains
    plugin.cfg = _Cfg(max_detections=40, model=_Cfg(backend=_Cfg(params=_Cfg(rotated_boxes=rotated, num_classes=C))))
    plugin.backend = model
    plugin.device = torch.device(DEV)
    plugin.optimizer = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, weight_decay=1e-4)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0,
                                          periodic=rotated)
    plugin.init_done()
    if tag == 'advent':
        dshapes = dict(ast.literal_eval(str(g['dshapes_json'])))
        assert sorted(plugin.discriminator.state_dict()) == sorted(dshapes)
        plugin.discriminator.load_state_dict(
            {k: T(gin.fill_value('discriminator.' + k, tuple(v))) for k, v in dshapes.items()})
    plugin.to(DEV)
    plugin.set_phase(True)
    data = {k: T(v) for k, v in gin.detection_batch(B, C, S // 4, S // 4, M, n_obj, 3 if rotated else 2, seed).items()}
    data['input'] = T(gin.image_batch(B, S, S, seed + 1))
    data['target_domain_input'] = T(gin.image_batch(B, S, S, seed + 2))
    out = plugin.step(data)
    assert list(out['stats']) == [str(k) for k in g['stat_keys']]
    for k, v in out['stats'].items():
        assert not v.is_cuda and not v.requires_grad
        _close_rel(v.item(), g['stat_' + k], what='stat ' + k)
    for k in ('hm', 'wh', 'reg'):
        _close_rel(out['source_domain'][k].detach().cpu().numpy(), g['src_' + k], what='source ' + k)   # hm: Q1
        _close_rel(out['target_domain'][k].detach().cpu().numpy(), g['tgt_' + k], what='target ' + k)
    np.testing.assert_array_equal(data['wh'].cpu().numpy(), g['wh_target_after'])                       # Q2
    params = dict(model.named_parameters())

    def check_gradsums(prefix, named, pprefix):
        for fk in g.files:
            if not fk.startswith(prefix):
                continue
            n = fk[len(prefix):]
            got, w32, w64 = _checksums(named[n].grad), g[fk], g['f64_' + fk]
            noise, err = np.abs(w32 - w64).max(), np.abs(_checksums(named[n].grad) - w64).max()
            print('%-58s err/|g|_1 %.2e  reference noise/|g|_1 %.2e'
                  % (n, err / max(w64[1], 1e-300), noise / max(w64[1], 1e-300)))
            assert err <= max(1e-4 * w64[1], 8 * noise), (n, got, w64, noise)
            gotp, wantp = _checksums(named[n]), g[pprefix + n]
            flips = 0.05 * named[n].numel() * 2 * (5e-5 if pprefix == 'param__' else 1e-3)   # Adam sign flips
            assert np.abs(gotp - wantp).max() <= 1e-5 * max(1.0, wantp[1]) + flips, (n, gotp, wantp)
    check_gradsums('gradsum__', params, 'param__')
    _check_full_grads(g, params)
    if tag == 'advent':
        check_gradsums('dgradsum__', dict(plugin.discriminator.named_parameters()), 'dparam__')
        _check_full_grads(g, dict(plugin.discriminator.named_parameters()), 'dgrad__')
        _close_rel(out['source_generator'].detach().cpu().numpy(), g['source_generator'], what='source_generator')
        assert all(p.requires_grad for p in plugin.discriminator.parameters())
    sd = model.state_dict()
    for fk in g.files:
        if fk.startswith('rm__'):
            n = fk[4:]
            _close_rel(sd[n + '.running_mean'].cpu().numpy(), g[fk], what='running_mean ' + n)
            _close_rel(sd[n + '.running_var'].cpu().numpy(), g['rv__' + n], what='running_var ' + n)
            assert int(sd[n + '.num_batches_tracked']) == int(g['nbt__' + n]) == 2                   # Q6
