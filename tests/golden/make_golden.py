#!/usr/bin/env python
"""Generates tests/golden/*.npz by IMPORTING the reference (read-only, from
/root/reference) in the build container.  Only inputs' recipes (inputs.py) and
the reference's OUTPUT VALUES are written; no reference source travels.

    python tests/golden/make_golden.py            # needs /root/reference

What runs as-is from the reference: utils.tensor, backends.decode,
losses.{centernet,entropy,max_square,advent(.crit)}, utils.image.entropy_map
(imported through a stub for the unrelated cv2/imgaug imports at the top of that
file -- see _load_entropy_map), backends.dla.
What cannot: the native `_ext` (libs/DCNv2/src needs <TH/TH.h>, absent from
this image -> unbuildable here).  backends.dla is therefore imported with
`sys.modules['_ext']` bound to this repo's CPU oracle (oracle/dcn.py), and
`torchsummary` (imported but unused, backends/dla.py:15) bound to an empty
module; `DLA.load_pretrained_model` (an HTTP download, dla.py:297-309) is
replaced by a no-op.  The DLA fixtures thus pin the reference's *network
wiring, BN/conv/upsample arithmetic and head layout* around the oracle's DCN;
the DCN arithmetic itself is pinned by tests/test_oracle_dcn.py.
"""
import os
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('CENTERNET_UDA_REFERENCE', '/root/reference')
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
sys.path.insert(2, HERE)

import torch  # noqa: E402

import inputs as gin  # noqa: E402
from oracle import dcn as oracle_dcn  # noqa: E402

warnings.filterwarnings('ignore')
torch.manual_seed(0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print('wrote %-28s %7.1f KiB' % (name + '.npz', os.path.getsize(path) / 1024))


# ---------------------------------------------------------------------------
def make_decode():
    from backends.decode import decode_detection, _nms, _topk
    for name in gin.DECODE_CASES:
        d = gin.decode_inputs(name)
        heat = T(d['heat'])
        nmsd = _nms(heat.clone())
        scores, inds, clses, ys, xs = _topk(nmsd, K=d['K'])
        # tie-free check (torch.topk order among equals is unspecified)
        s = scores.numpy()
        assert (s[:, :-1] > s[:, 1:]).all(), 'fixture %s has tied scores' % name
        assert s.min() > 0, 'fixture %s selected suppressed (zero) entries' % name
        dets = decode_detection(heat.clone(), T(d['wh']).clone(),
                                reg=None if d['reg'] is None else T(d['reg']).clone(),
                                K=d['K'], rotated=d['rotated'])
        nz = nmsd.numpy() != 0
        extra = {}
        if name in ('small', 'noreg'):          # keypoint branch (decode.py:69-74), with and without `reg`
            d2, kps = decode_detection(heat.clone(), T(d['wh']).clone(),
                                       reg=None if d['reg'] is None else T(d['reg']).clone(),
                                       kps=T(gin.decode_kps_inputs(name)).clone(), K=d['K'], rotated=d['rotated'])
            assert torch.equal(d2, dets)
            extra['kps'] = kps.numpy()
        save('decode_' + name, dets=dets.numpy(), inds=inds.numpy(), clses=clses.numpy(),
             nms_nonzero_count=np.array(nz.sum()), nms_sum=np.array(nmsd.double().sum().item()), **extra)


# ---------------------------------------------------------------------------
def make_losses():
    from losses.centernet import DetectionLoss
    from losses.entropy import EntropyLoss
    from losses.max_square import MaxSquareLoss
    from losses.advent import AdventLoss
    for name in gin.LOSS_CASES:
        out_np, batch_np, w = gin.loss_inputs(name)
        out = {k: T(v).clone().requires_grad_(True) for k, v in out_np.items()}
        leaves = dict(out)
        # the loss mutates its inputs in place (Q1/Q2): hand it non-leaf views
        out_in = {k: v * 1.0 for k, v in out.items()}
        batch = {k: T(v).clone() for k, v in batch_np.items()}
        crit = DetectionLoss(**w)
        loss, stats = crit(out_in, batch)
        loss.backward()
        save('losses_det_' + name,
             loss=loss.item(), **{'stat_' + k: v.item() for k, v in stats.items()},
             hm_after=out_in['hm'].detach().numpy(),            # Q1: sigmoid-clamped in place
             wh_target_after=batch['wh'].numpy(), reg_target_after=batch['reg'].numpy(),  # Q2
             **{'grad_' + k: v.grad.numpy() for k, v in leaves.items()})
    for name in gin.KPS_CASES:                   # DetectionLoss with the keypoint term (KPSL1Loss)
        out_np, batch_np, w = gin.kps_inputs(name)
        out = {k: T(v).clone().requires_grad_(True) for k, v in out_np.items()}
        leaves = dict(out)
        out_in = {k: v * 1.0 for k, v in out.items()}
        batch = {k: T(v).clone() for k, v in batch_np.items()}
        loss, stats = DetectionLoss(**w)(out_in, batch)
        loss.backward()
        save('losses_kps_' + name, loss=loss.item(), **{'stat_' + k: v.item() for k, v in stats.items()},
             kps_target_after=batch['kps'].numpy(), **{'grad_' + k: v.grad.numpy() for k, v in leaves.items()})
    # UDA losses on raw logits
    rs = np.random.RandomState(31)
    hm = (rs.standard_normal((2, 6, 16, 16)) * 2.0 - 1.0).astype(np.float32)
    res = {'hm': hm}
    for tag, mod, key in (('entropy', EntropyLoss(), 'entropy_loss'),
                          ('maxsq', MaxSquareLoss(), 'max_square_loss')):
        x = T(hm).clone().requires_grad_(True)
        l, st = mod({'hm': x}, None)
        l.backward()
        res[tag + '_loss'] = l.item()
        res[tag + '_grad'] = x.grad.numpy()
        assert abs(st[key].item() - l.item()) == 0
    from utils_image_entropy import entropy_map
    x = T(hm).clone().requires_grad_(True)
    em = entropy_map(x)
    em.sum().backward()
    res['entropy_map'] = em.detach().numpy()
    res['entropy_map_grad_of_sum'] = x.grad.numpy()
    # AdventLoss.forward calls y_pred.get_device() (-1 on CPU) and fails; its
    # arithmetic is the member `crit` (advent.py:8) against a filled label.
    adv = AdventLoss()
    logits = (rs.standard_normal((2, 1, 5, 5)) * 1.5).astype(np.float32)
    res['advent_logits'] = logits
    for label in (0, 1):
        y = T(logits).clone().requires_grad_(True)
        l = adv.crit(y, torch.full_like(y, float(label)))
        l.backward()
        res['advent_loss_%d' % label] = l.item()
        res['advent_grad_%d' % label] = y.grad.numpy()
    save('losses_uda', **res)


def _load_entropy_map():
    """utils/image.py imports cv2/imgaug at module scope (absent here); only
    its pure-torch `entropy_map` (:121-124) is on the hot path.  Import the
    module with empty stand-ins for those two unrelated packages."""
    for missing in ('cv2', 'imgaug', 'imgaug.augmenters'):
        if missing not in sys.modules:
            try:
                __import__(missing)
            except Exception:
                sys.modules[missing] = types.ModuleType(missing)
    import importlib
    m = importlib.import_module('utils.image')
    shim = types.ModuleType('utils_image_entropy')
    shim.entropy_map = m.entropy_map
    sys.modules['utils_image_entropy'] = shim


# ---------------------------------------------------------------------------
def _import_reference_dla():
    ext = types.ModuleType('_ext')
    ext.dcn_v2_forward = oracle_dcn.dcn_v2_forward
    ext.dcn_v2_backward = oracle_dcn.dcn_v2_backward
    sys.modules['_ext'] = ext
    ts = types.ModuleType('torchsummary')
    ts.summary = lambda *a, **k: None
    sys.modules['torchsummary'] = ts
    from backends import dla
    dla.DLA.load_pretrained_model = lambda self, *a, **k: None
    return dla


def _checksums(t):
    t = t.detach().double().reshape(-1)
    idx = torch.arange(t.numel(), dtype=torch.float64)
    return np.array([t.sum().item(), t.abs().sum().item(), (t * torch.cos(0.01 * idx)).sum().item()])


GRAD_PROBES = [
    'base.base_layer.0.weight', 'base.level2.tree1.conv1.weight', 'base.level3.tree2.root.conv.weight',
    'base.level5.project.0.weight', 'base.level4.tree1.tree2.bn2.weight',
    'dla_up.ida_0.proj_1.conv.weight', 'dla_up.ida_0.proj_1.conv.conv_offset_mask.weight',
    'dla_up.ida_2.node_3.conv.bias', 'dla_up.ida_1.up_2.weight', 'ida_up.proj_2.conv.conv_offset_mask.bias',
    'ida_up.node_2.actf.0.bias', 'hm.0.weight', 'hm.2.bias', 'wh.2.weight', 'reg.0.bias',
]


# Whole gradient tensors (small ones) stored by the *128 step fixtures: a sparse defect -- a dropped store, one wrong tile --
# passes three moments of a tensor; it does not pass an element-wise comparison.  One detection-head output layer, one DCN
# offset / mask convolution (its gradient flows through the deformable sampling's coordinate path), a strided 3x3 of the
# base, a depthwise transposed convolution of IDAUp, and a BatchNorm scale deep in the tree.
FULL_GRADS = [
    'hm.2.weight', 'wh.2.weight', 'ida_up.node_1.conv.conv_offset_mask.weight', 'base.level2.tree1.conv1.weight',
    'dla_up.ida_1.up_2.weight', 'base.level4.tree1.tree2.bn2.weight',
]


def _dla_case(dla, rotated, B, S, seed, dtype):
    """Reference DLASeg at `dtype` (float32 = the reference's arithmetic; float64 =
    the same module evaluated exactly, used by the tests to express tolerances
    relative to the reference's own rounding noise)."""
    model = dla.build(num_classes=6, rotated_boxes=rotated)
    sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items()}
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes).items()})
    model = model.to(dtype)
    x = T(gin.image_batch(B, S, S, seed)).to(dtype)
    res = {}
    model.eval()
    with torch.no_grad():
        out = model(x)
    for k in out:
        res['eval_' + k] = out[k].numpy()
    model.train()
    out = model(x)
    res['head_order'] = np.array(list(out.keys()))
    for k in out:
        res['train_' + k] = out[k].detach().numpy()
    scalar = sum((out[k] * torch.cos(torch.arange(out[k].numel(), dtype=dtype)
                                     .reshape(out[k].shape) * 0.1)).sum() for k in out)
    scalar.backward()
    res['scalar'] = scalar.item()
    params = dict(model.named_parameters())
    for n in GRAD_PROBES:
        res['gradsum__' + n] = _checksums(params[n].grad)
    res['grad_none'] = np.array(sorted(n for n, p in params.items() if p.grad is None))
    sd2 = model.state_dict()
    for n in ('base.base_layer.1', 'base.level3.project.1', 'base.level4.project.1',
              'dla_up.ida_0.node_1.actf.0', 'ida_up.node_2.actf.0'):
        res['rm__' + n] = sd2[n + '.running_mean'].numpy()
        res['rv__' + n] = sd2[n + '.running_var'].numpy()
        res['nbt__' + n] = sd2[n + '.num_batches_tracked'].numpy()
    meta = {'state_names': np.array(sorted(shapes)), 'n_params': sum(p.numel() for p in model.parameters()),
            'shapes_json': np.array(repr(sorted((k, v) for k, v in shapes.items())))}
    return res, meta


def make_dla():
    dla = _import_reference_dla()
    for tag, rotated, B, S, seed in DLA_CASES:
        r32, meta = _dla_case(dla, rotated, B, S, seed, torch.float32)
        r64, _ = _dla_case(dla, rotated, B, S, seed, torch.float64)
        out = dict(meta)
        out.update(r32)
        for k, v in r64.items():
            if k.startswith(('eval_', 'train_', 'gradsum__', 'scalar')):
                out['f64_' + k] = v
        save('dla_' + tag, **out)
    return dla


DLA_CASES = (('axis', False, 2, 64, 41), ('rot', True, 2, 96, 42))


def _step_case(dla, tag, dtype):
    """The reference's own plugin class (uda/entropy_minimization.py:5-43, uda/max_squares_minimization.py:5-50)
    driving the imported DLA-34, DetectionLoss and torch.optim.Adam (train.py:88-90)."""
    from losses.centernet import DetectionLoss
    uda = _import_reference_uda()
    plugin = uda.EntropyMinimization(1e-4) if tag == 'entropy' else uda.MaxSquaresMinimization(0.3)
    B, S, M = 2, 64, 8
    model = dla.build(num_classes=6)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes).items()})
    model = model.to(dtype)
    plugin.backend = model
    plugin.device = torch.device('cpu')
    plugin.optimizer = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, weight_decay=1e-4)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0, periodic=False)
    plugin.init_done()
    plugin.to('cpu')
    plugin.set_phase(True)
    batch = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (3, 2), 2, 51).items()}
    for k in ('hm', 'wh', 'reg'):
        batch[k] = batch[k].to(dtype)
    batch['input'] = T(gin.image_batch(B, S, S, 52)).to(dtype)
    batch['target_domain_input'] = T(gin.image_batch(B, S, S, 53)).to(dtype)
    out = plugin.step(batch)
    res = {'stat_' + k: v.item() for k, v in out['stats'].items()}
    params = dict(model.named_parameters())
    for n in GRAD_PROBES:
        if n in params and params[n].grad is not None:
            res['gradsum__' + n] = _checksums(params[n].grad)
            res['param__' + n] = _checksums(params[n])
    sd = model.state_dict()
    res['rm__base.base_layer.1'] = sd['base.base_layer.1.running_mean'].numpy()
    res['rv__base.base_layer.1'] = sd['base.base_layer.1.running_var'].numpy()
    res['nbt__base.base_layer.1'] = sd['base.base_layer.1.num_batches_tracked'].numpy()
    res['src_hm_after'] = out['source_domain']['hm'].detach().numpy()
    return res


def make_step(dla):
    """One `EntropyMinimization.step` (uda/entropy_minimization.py:11-43) and one
    `MaxSquaresMinimization.step` of the imported reference classes, in float32 and,
    for tolerance calibration, float64."""
    for tag in ('entropy', 'maxsq'):
        r32 = _step_case(dla, tag, torch.float32)
        r64 = _step_case(dla, tag, torch.float64)
        out = dict(r32)
        for k, v in r64.items():
            if k.startswith(('stat_', 'gradsum__', 'src_hm_after')):
                out['f64_' + k] = v
        save('step_' + tag, **out)


def _advent_case(dla, dtype):
    """`AdversarialEntropyMinimization.step` (uda/adversarial_entropy_minimization.py:77-152) re-enacted with
    the imported reference pieces on a rotated-box model with the periodic angle loss (cfg5 semantics).
    Not importable as a class here: the module needs hydra, and AdventLoss.forward calls `.to(-1)` on CPU
    (losses/advent.py:14) -- its arithmetic is the member `crit` against a filled label.  The discriminator is
    built from the layer list of get_fc_discriminator (:51-68)."""
    from losses.centernet import DetectionLoss
    from losses.advent import AdventLoss
    from utils_image_entropy import entropy_map
    from torch import nn
    B, S, M, C = 2, 128, 8, 6
    model = dla.build(num_classes=C, rotated_boxes=True)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes).items()})
    ndf = 64
    D = nn.Sequential(
        nn.Conv2d(C, ndf, 4, 2, 1), nn.LeakyReLU(0.2, inplace=True),
        nn.Conv2d(ndf, ndf * 2, 4, 2, 1), nn.LeakyReLU(0.2, inplace=True),
        nn.Conv2d(ndf * 2, ndf * 4, 4, 2, 1), nn.LeakyReLU(0.2, inplace=True),
        nn.Conv2d(ndf * 4, ndf * 8, 4, 2, 1), nn.LeakyReLU(0.2, inplace=True),
        nn.Conv2d(ndf * 8, 1, 4, 2, 1))
    dshapes = {k: tuple(v.shape) for k, v in D.state_dict().items()}
    D.load_state_dict({k: T(gin.fill_value('discriminator.' + k, v)) for k, v in dshapes.items()})
    model, D = model.to(dtype), D.to(dtype)
    model.train(); D.train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-4)
    dopt = torch.optim.Adam(D.parameters(), lr=1e-3, weight_decay=1e-4)
    crit = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0, periodic=True)
    adv = AdventLoss()
    bce = lambda y, label: adv.crit(y, torch.full_like(y, float(label)))
    batch = {k: T(v) for k, v in gin.detection_batch(B, C, S // 4, S // 4, M, (4, 2), 3, 71).items()}
    for k in ('hm', 'wh', 'reg'):
        batch[k] = batch[k].to(dtype)
    x_s, x_t = T(gin.image_batch(B, S, S, 72)).to(dtype), T(gin.image_batch(B, S, S, 73)).to(dtype)
    w_adv = 1e-4
    opt.zero_grad(); dopt.zero_grad()
    for p_ in D.parameters():
        p_.requires_grad = False
    out_s, out_t = model(x_s), model(x_t)
    fool = D(entropy_map(out_t['hm']))
    loss, stats = crit(out_s, batch)
    loss.backward()
    dtf = bce(fool, 0)
    dtf *= w_adv
    dtf.backward()
    for p_ in D.parameters():
        p_.requires_grad = True
    source, target = out_s['hm'].detach(), out_t['hm'].detach()
    ds = bce(D(entropy_map(source)), 0)
    ds /= 2.0
    ds.backward()
    dt = bce(D(entropy_map(target)), 1)
    dt /= 2.0
    dt.backward()
    opt.step(); dopt.step()
    res = {'stat_' + k: v.item() for k, v in stats.items()}
    res.update(stat_total_loss=(loss + ds + dt + dtf).item(), stat_dis_soruce=ds.item(), stat_dis_target=dt.item(),
               stat_dis_fool=dtf.item())
    for n, p_ in D.named_parameters():
        res['dgradsum__' + n] = _checksums(p_.grad)
        res['dparam__' + n] = _checksums(p_)
    params = dict(model.named_parameters())
    for n in GRAD_PROBES:
        if params[n].grad is not None:
            res['gradsum__' + n] = _checksums(params[n].grad)
    res['src_hm_after'] = out_s['hm'].detach().numpy()
    res['wh_target_after'] = batch['wh'].numpy()
    res['dshapes_json'] = np.array(repr(sorted(dshapes.items())))
    res['shapes_json'] = np.array(repr(sorted(shapes.items())))
    return res


def make_advent(dla):
    r32 = _advent_case(dla, torch.float32)
    r64 = _advent_case(dla, torch.float64)
    out = dict(r32)
    for k, v in r64.items():
        if k.startswith(('stat_', 'gradsum__', 'dgradsum__', 'src_hm_after')):
            out['f64_' + k] = v
    save('step_advent', **out)


# ---------------------------------------------------------------------------
def _import_reference_uda():
    """The reference's own step classes (uda/base.py, uda/entropy_minimization.py, uda/max_squares_minimization.py).
    `utils/helper.py` imports hydra / omegaconf at module scope (configuration packages, absent here, touched only
    by instantiate_augmenters and by ADVENT's custom-optimizer lookup); they are bound to empty modules the same
    way torchsummary / cv2 / imgaug are.  No arithmetic on the path lives in them."""
    for m in ('hydra', 'hydra.utils', 'omegaconf', 'omegaconf.listconfig'):
        if m not in sys.modules:
            try:
                __import__(m)
            except Exception:
                sys.modules[m] = types.ModuleType(m)
    if not hasattr(sys.modules['omegaconf.listconfig'], 'ListConfig'):
        sys.modules['omegaconf.listconfig'].ListConfig = list
    if not hasattr(sys.modules['hydra'], 'utils'):
        sys.modules['hydra'].utils = sys.modules['hydra.utils']
    _load_entropy_map()                                  # cv2 / imgaug stand-ins for utils/image.py
    import uda
    import uda.base
    return uda


def _ns(**kw):
    return types.SimpleNamespace(**kw)


def _cfg(K, rotated):
    return _ns(max_detections=K, model=_ns(backend=_ns(params=_ns(rotated_boxes=rotated))))


def _store_detections(res, prefix, dets):
    """uda/base.py:121-137's dict -> flat arrays (per-image ground-truth lists concatenated, with their lengths)."""
    for k in ('pred_boxes', 'pred_classes', 'pred_scores'):
        res[prefix + k] = np.asarray(dets[k])
    if 'pred_kps' in dets:
        res[prefix + 'pred_kps'] = np.asarray(dets['pred_kps'])
    res[prefix + 'gt_counts'] = np.array([len(b) for b in dets['gt_boxes']])
    for k in ('gt_boxes', 'gt_classes', 'gt_areas') + (('gt_kps',) if 'gt_kps' in dets else ()):
        res[prefix + k] = np.concatenate([np.asarray(v) for v in dets[k]], 0)
        res[prefix + k + '_dtype'] = np.array(str(np.asarray(dets[k][0]).dtype))
    res[prefix + 'gt_ids'] = np.asarray(dets['gt_ids'])
    res[prefix + 'pred_classes_dtype'] = np.array(str(dets['pred_classes'].dtype))


def make_getdet():
    """`Model.get_detections` (uda/base.py:73-139) of the imported reference class on synthetic head outputs:
    the x down_ratio scaling, the reg_mask == 1 row selection, the 4/5 vs 5/6 column split for rotated boxes,
    the keypoint branch."""
    uda = _import_reference_uda()
    for name in gin.GETDET_CASES:
        src_np, batch_np, K, rotated = gin.getdet_inputs(name)
        m = uda.base.Model()
        m.cfg = _cfg(K, rotated)
        m.backend = _ns(down_ratio=4)
        res = {}
        dets = m.get_detections({'source_domain': {k: T(v).clone() for k, v in src_np.items()}},
                                {k: T(v).clone() for k, v in batch_np.items()})
        _store_detections(res, '', dets)
        save('getdet_' + name, **res)


BASE_STEP = dict(B=4, S=128, M=16, n_obj=(5, 1, 9, 3), offset_gain=0.1, K=40)


def _base_step_case(uda, dla, dtype):
    """configs[1]: `uda.base.Model` of the imported reference driving DLA-34 -- the evaluation sequence of
    train.py:204-223 (set_phase(False), `step(data, is_training=False)` under no_grad, `get_detections`), then
    one training `step` (uda/base.py:31-56).  Well-conditioned variant of the parameter fill
    (inputs.fill_value, offset_gain = 0.1) at B = 4, 128 x 128 so that float32 agrees with float64 to ~1e-5."""
    from losses.centernet import DetectionLoss
    c = BASE_STEP
    B, S, M = c['B'], c['S'], c['M']
    model = dla.build(num_classes=6)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, c['offset_gain']).items()})
    model = model.to(dtype)
    plugin = uda.base.Model()
    plugin.cfg = _cfg(c['K'], False)
    plugin.backend = model
    plugin.device = torch.device('cpu')
    plugin.optimizer = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, weight_decay=1e-4)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0, periodic=False)
    plugin.init_done()
    plugin.to('cpu')

    def batch(seed):
        d = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, c['n_obj'], 2, seed).items()}
        for k in ('hm', 'wh', 'reg'):
            d[k] = d[k].to(dtype)
        d['input'] = T(gin.image_batch(B, S, S, seed + 1)).to(dtype)
        return d
    # evaluation sequence (train.py:204-223) on the freshly filled model.  It runs BEFORE the training step: after
    # one Adam step the parameters whose gradient is analytically zero (convolution biases in front of a BatchNorm)
    # have moved by +-lr on rounding noise alone, and eval-mode outputs of float32 and float64 runs of the
    # reference itself then differ by 7e-3 -- nothing a 1e-4 comparison could be held to.
    res = {}
    plugin.set_phase(False)
    ev = batch(93)
    ev.update({k: T(v) for k, v in gin.eval_extras(B, M, False, 95).items()})
    ev['gt_dets'] = ev['gt_dets'].to(dtype)
    with torch.no_grad():
        o = plugin.step(ev, is_training=False)
    for k, v in o['stats'].items():
        res['eval_stat_' + k] = v.item()
    for k in ('hm', 'wh', 'reg'):
        res['eval_' + k] = o['source_domain'][k].numpy()
    dets = plugin.get_detections(o, ev)
    _store_detections(res, 'det_', dets)
    plugin.set_phase(True)
    data = batch(91)
    out = plugin.step(data)
    res.update({'stat_' + k: v.item() for k, v in out['stats'].items()})
    res['stat_keys'] = np.array(list(out['stats']))
    for k in ('hm', 'wh', 'reg'):
        res['train_' + k] = out['source_domain'][k].detach().numpy()       # hm: clamped probabilities (Q1)
    res['wh_target_after'] = data['wh'].numpy()                            # Q2
    params = dict(model.named_parameters())
    for n in GRAD_PROBES:
        if params[n].grad is not None:
            res['gradsum__' + n] = _checksums(params[n].grad)
            res['param__' + n] = _checksums(params[n])
    for n in FULL_GRADS:
        res['grad__' + n] = params[n].grad.detach().numpy().copy()
    sd = model.state_dict()
    for n in ('base.base_layer.1', 'base.level5.tree2.bn2', 'ida_up.node_2.actf.0'):
        res['rm__' + n] = sd[n + '.running_mean'].numpy()
        res['rv__' + n] = sd[n + '.running_var'].numpy()
        res['nbt__' + n] = sd[n + '.num_batches_tracked'].numpy()
    res['shapes_json'] = np.array(repr(sorted(shapes.items())))
    return res


def make_base_step(dla):
    uda = _import_reference_uda()
    r32 = _base_step_case(uda, dla, torch.float32)
    r64 = _base_step_case(uda, dla, torch.float64)
    out = dict(r32)
    for k, v in r64.items():
        if k.startswith(('stat_', 'eval_', 'train_', 'gradsum__', 'grad__')) and k != 'stat_keys':
            out['f64_' + k] = v
    save('step_base128', **out)


UDA_STEP128 = dict(B=4, S=128, M=16, n_obj=(5, 1, 9, 3), offset_gain=0.1, C=6)
UDA_STEP128_CASES = {
    # tag: (reference class name, constructor arguments, rotated boxes + periodic angle loss, batch seed)
    'entropy': ('EntropyMinimization', (1e-4,), False, 111),
    'maxsq': ('MaxSquaresMinimization', (0.3,), False, 121),
    'advent': ('AdversarialEntropyMinimization', (1e-4,), True, 131),
}


def _uda_step128_case(uda, dla, tag, dtype):
    """The benchmarked step and its two siblings at plain tolerance: the reference's OWN plugin classes
    (uda/entropy_minimization.py:11-43, uda/max_squares_minimization.py:22-50,
    uda/adversarial_entropy_minimization.py:77-152) driving the imported DLA-34 at B = 4 + 4, 128 x 128, with the
    well-conditioned parameter fill of `step_base128` (offset_gain = 0.1: DCN offsets of about +-0.1 px).
    ADVENT: the class runs as imported (its discriminator from get_fc_discriminator, its default Adam); only
    `AdventLoss.forward` cannot execute on a CPU tensor (`y_t.to(y_pred.get_device())` is `.to(-1)`,
    losses/advent.py:14) -- it is replaced by the same arithmetic, the member `crit` against a filled label."""
    from losses.centernet import DetectionLoss
    c = UDA_STEP128
    cls, args, rotated, seed = UDA_STEP128_CASES[tag]
    B, S, M, C = c['B'], c['S'], c['M'], c['C']
    model = dla.build(num_classes=C, rotated_boxes=rotated)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, c['offset_gain']).items()})
    model = model.to(dtype)
    plugin = getattr(uda, cls)(*args)
    plugin.cfg = _ns(max_detections=40, model=_ns(backend=_ns(params=_ns(rotated_boxes=rotated, num_classes=C))))
    plugin.backend = model
    plugin.device = torch.device('cpu')
    plugin.optimizer = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, weight_decay=1e-4)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0,
                                          periodic=rotated)
    plugin.init_done()
    res = {}
    if tag == 'advent':
        D = plugin.discriminator
        dshapes = {k: tuple(v.shape) for k, v in D.state_dict().items()}
        D.load_state_dict({k: T(gin.fill_value('discriminator.' + k, v)) for k, v in dshapes.items()})
        D.to(dtype)
        crit = plugin.adversarial_loss.crit

        def advent_forward(y_pred, y_true):
            loss = crit(y_pred, torch.full_like(y_pred, float(y_true)))
            return loss, {'advent_loss': loss}
        plugin.adversarial_loss.forward = advent_forward
        res['dshapes_json'] = np.array(repr(sorted(dshapes.items())))
    plugin.to('cpu')
    plugin.set_phase(True)
    batch = {k: T(v) for k, v in gin.detection_batch(B, C, S // 4, S // 4, M, c['n_obj'], 3 if rotated else 2,
                                                     seed).items()}
    for k in ('hm', 'wh', 'reg'):
        batch[k] = batch[k].to(dtype)
    batch['input'] = T(gin.image_batch(B, S, S, seed + 1)).to(dtype)
    batch['target_domain_input'] = T(gin.image_batch(B, S, S, seed + 2)).to(dtype)
    out = plugin.step(batch)
    res.update({'stat_' + k: v.item() for k, v in out['stats'].items()})
    res['stat_keys'] = np.array(list(out['stats']))
    for k in ('hm', 'wh', 'reg'):
        res['src_' + k] = out['source_domain'][k].detach().numpy()          # hm: clamped probabilities (Q1)
        res['tgt_' + k] = out['target_domain'][k].detach().numpy()          # hm: raw logits (no loss rebinds them)
    res['wh_target_after'] = batch['wh'].numpy()                            # Q2
    params = dict(model.named_parameters())
    for n in GRAD_PROBES:
        if n in params and params[n].grad is not None:
            res['gradsum__' + n] = _checksums(params[n].grad)
            res['param__' + n] = _checksums(params[n])
    for n in FULL_GRADS:
        res['grad__' + n] = params[n].grad.detach().numpy().copy()
    if tag == 'advent':
        for n, p_ in plugin.discriminator.named_parameters():
            res['dgradsum__' + n] = _checksums(p_.grad)
            res['dparam__' + n] = _checksums(p_)
        res['dgrad__0.weight'] = dict(plugin.discriminator.named_parameters())['0.weight'].grad.detach().numpy().copy()
        res['source_generator'] = out['source_generator'].detach().numpy()
    sd = model.state_dict()
    for n in ('base.base_layer.1', 'base.level5.tree2.bn2', 'ida_up.node_2.actf.0'):
        res['rm__' + n] = sd[n + '.running_mean'].numpy()
        res['rv__' + n] = sd[n + '.running_var'].numpy()
        res['nbt__' + n] = sd[n + '.num_batches_tracked'].numpy()
    res['shapes_json'] = np.array(repr(sorted(shapes.items())))
    return res


def make_uda_step128(dla, tags=None):
    uda = _import_reference_uda()
    for tag in (tags or UDA_STEP128_CASES):
        r32 = _uda_step128_case(uda, dla, tag, torch.float32)
        r64 = _uda_step128_case(uda, dla, tag, torch.float64)
        out = dict(r32)
        for k, v in r64.items():
            if k.startswith(('stat_', 'src_', 'tgt_', 'gradsum__', 'dgradsum__', 'grad__', 'dgrad__', 'source_generator',
                             'rm__', 'rv__')) \
                    and k != 'stat_keys':
                out['f64_' + k] = v
        save('step_%s128' % tag, **out)


# ---------------------------------------------------------------------------
RESNET_GRAD_PROBES = [
    'base.0.weight', 'base.1.weight', 'base.4.1.conv2.weight', 'base.5.0.downsample.0.weight',
    'base.5.0.downsample.1.bias', 'base.6.1.bn2.weight', 'base.7.0.conv1.weight', 'base.7.1.conv2.weight',
    'deconv_layers.0.weight', 'deconv_layers.4.weight', 'deconv_layers.6.weight', 'deconv_layers.7.bias',
    'hm.0.weight', 'hm.2.bias', 'wh.2.weight', 'reg.0.bias',
]
RESNET_CASES = (('fwd', 2, 128, 61),)          # tag, B, S, seed  (eval + train forward, synthetic backward)
RESNET_STEP = (2, 256, 16)                     # configs[0]: B=2, 256x256, M objects slots


def _import_reference_resnet():
    """backends/resnet.py with `torch.hub.load` (a GitHub download of pytorch/vision v0.6.0, resnet.py:27-28)
    bound to the oracle's restated torchvision trunk -- see oracle/resnet.py's header."""
    from oracle import resnet as oracle_resnet
    torch.hub.load = lambda repo, name, pretrained=False, **kw: oracle_resnet.torchvision_resnet(
        int(name.replace('resnet', '')))
    from backends import resnet
    return resnet


def _resnet_model(resnet, dtype):
    model = resnet.build(18, num_classes=6, pretrained=False)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes).items()})
    return model.to(dtype), shapes


def _resnet_case(resnet, B, S, seed, dtype):
    model, shapes = _resnet_model(resnet, dtype)
    x = T(gin.image_batch(B, S, S, seed)).to(dtype)
    res = {}
    model.eval()
    with torch.no_grad():
        out = model(x)
    for k in out:
        res['eval_' + k] = out[k].numpy()
    model.train()
    out = model(x)
    res['head_order'] = np.array(list(out.keys()))
    for k in out:
        res['train_' + k] = out[k].detach().numpy()
    scalar = sum((out[k] * torch.cos(torch.arange(out[k].numel(), dtype=dtype)
                                     .reshape(out[k].shape) * 0.1)).sum() for k in out)
    scalar.backward()
    res['scalar'] = scalar.item()
    params = dict(model.named_parameters())
    for n in RESNET_GRAD_PROBES:
        res['gradsum__' + n] = _checksums(params[n].grad)
    sd2 = model.state_dict()
    for n in ('base.1', 'base.5.0.downsample.1', 'base.7.1.bn2', 'deconv_layers.4'):
        res['rm__' + n] = sd2[n + '.running_mean'].numpy()
        res['rv__' + n] = sd2[n + '.running_var'].numpy()
        res['nbt__' + n] = sd2[n + '.num_batches_tracked'].numpy()
    meta = {'state_names': np.array(list(shapes)), 'param_names': np.array([n for n, _ in model.named_parameters()]),
            'n_params': sum(p.numel() for p in model.parameters()),
            'shapes_json': np.array(repr(sorted((k, v) for k, v in shapes.items())))}
    return res, meta


def _resnet_step(resnet, dtype):
    """`Model.step` (uda/base.py:31-56) at configs[0]'s real size, re-enacted with the imported backend and
    DetectionLoss (uda.base itself imports hydra through utils.helper)."""
    from losses.centernet import DetectionLoss
    B, S, M = RESNET_STEP
    model, _ = _resnet_model(resnet, dtype)
    model.train()
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5)
    crit = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0, periodic=False)
    batch = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (5, 3), 2, 71).items()}
    for k in ('hm', 'wh', 'reg'):
        batch[k] = batch[k].to(dtype)
    batch['input'] = T(gin.image_batch(B, S, S, 72)).to(dtype)
    opt.zero_grad()
    out = model(batch['input'])
    loss, stats = crit(out, batch)
    loss.backward()
    opt.step()
    stats['total_loss'] = loss
    res = {'stat_' + k: v.item() for k, v in stats.items()}
    params = dict(model.named_parameters())
    for n in RESNET_GRAD_PROBES:
        res['gradsum__' + n] = _checksums(params[n].grad)
        res['param__' + n] = _checksums(params[n])
    res['hm_after'] = out['hm'].detach().numpy()
    return res


def make_resnet():
    resnet = _import_reference_resnet()
    for tag, B, S, seed in RESNET_CASES:
        r32, meta = _resnet_case(resnet, B, S, seed, torch.float32)
        r64, _ = _resnet_case(resnet, B, S, seed, torch.float64)
        out = dict(meta)
        out.update(r32)
        for k, v in r64.items():
            if k.startswith(('eval_', 'train_', 'gradsum__', 'scalar')):
                out['f64_' + k] = v
        save('resnet18_' + tag, **out)
    r32 = _resnet_step(resnet, torch.float32)
    r64 = _resnet_step(resnet, torch.float64)
    out = dict(r32)
    for k, v in r64.items():
        if k.startswith(('stat_', 'gradsum__')):
            out['f64_' + k] = v
    out['hm_after'] = out['hm_after'][:, :, ::4, ::4].copy()       # keep the file small
    save('resnet18_step', **out)


# ---------------------------------------------------------------------------
MBV2_CASES = [('dcn', dict(use_dcn=True, use_skip=False), 2, 64, 95), ('skip', dict(use_dcn=False, use_skip=True), 2, 96, 96)]
MBV2_GRAD_PROBES = {
    'dcn': ['base.0.0.weight', 'base.1.conv.0.0.weight', 'base.3.conv.1.0.weight', 'base.7.conv.2.weight',
            'base.14.conv.3.bias', 'base.18.0.weight', 'deconv_layers.0.weight', 'deconv_layers.0.conv_offset_mask.weight',
            'deconv_layers.6.bias', 'deconv_layers.9.weight', 'deconv_layers.16.weight', 'hm.0.weight', 'wh.2.bias'],
    'skip': ['base.0.0.weight', 'base.6.conv.1.0.weight', 'base.13.conv.2.weight', 'base.17.conv.3.weight',
             'deconv_layers.0.weight', 'deconv_layers.3.weight', 'deconv_layers.7.weight', 'skip_0.weight', 'skip_3.bias',
             'reg.0.weight', 'hm.2.bias'],
}


def _import_reference_mobilenetv2():
    """backends/mobilenetv2.py with `torch.hub.load` (a GitHub download of pytorch/vision v0.6.0, :31-34) bound to
    the oracle's restated torchvision trunk, and the native `_ext` bound to the CPU DCN oracle (as for dla)."""
    from oracle import mobilenetv2 as oracle_mb
    ext = types.ModuleType('_ext')
    ext.dcn_v2_forward = oracle_dcn.dcn_v2_forward
    ext.dcn_v2_backward = oracle_dcn.dcn_v2_backward
    sys.modules['_ext'] = ext
    torch.hub.load = lambda repo, name, pretrained=False, **kw: oracle_mb.torchvision_mobilenet_v2()
    from backends import mobilenetv2
    return mobilenetv2


def _mbv2_case(mb, tag, flags, B, S, seed, dtype):
    model = mb.build(num_classes=6, pretrained=False, **flags)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes).items()})   # conv_offset_mask filled non-zero (Q7)
    model = model.to(dtype)
    x = T(gin.image_batch(B, S, S, seed)).to(dtype)
    res = {}
    model.eval()
    with torch.no_grad():
        out = model(x)
    for k in out:
        res['eval_' + k] = out[k].numpy()
    model.train()
    out = model(x)
    res['head_order'] = np.array(list(out.keys()))
    for k in out:
        res['train_' + k] = out[k].detach().numpy()
    scalar = sum((out[k] * torch.cos(torch.arange(out[k].numel(), dtype=dtype)
                                     .reshape(out[k].shape) * 0.1)).sum() for k in out)
    scalar.backward()
    res['scalar'] = scalar.item()
    params = dict(model.named_parameters())
    for n in MBV2_GRAD_PROBES[tag]:
        res['gradsum__' + n] = _checksums(params[n].grad)
    sd2 = model.state_dict()
    for n in ('base.0.1', 'base.5.conv.1.1', 'base.18.1', 'deconv_layers.1'):
        res['rm__' + n] = sd2[n + '.running_mean'].numpy()
        res['rv__' + n] = sd2[n + '.running_var'].numpy()
        res['nbt__' + n] = sd2[n + '.num_batches_tracked'].numpy()
    meta = {'state_names': np.array(list(shapes)), 'param_names': np.array([n for n, _ in model.named_parameters()]),
            'n_params': sum(p.numel() for p in model.parameters()),
            'shapes_json': np.array(repr(sorted((k, v) for k, v in shapes.items())))}
    return res, meta


def make_mobilenetv2():
    mb = _import_reference_mobilenetv2()
    for tag, flags, B, S, seed in MBV2_CASES:
        r32, meta = _mbv2_case(mb, tag, flags, B, S, seed, torch.float32)
        r64, _ = _mbv2_case(mb, tag, flags, B, S, seed, torch.float64)
        out = dict(meta)
        out.update(r32)
        for k, v in r64.items():
            if k.startswith(('eval_', 'train_', 'gradsum__', 'scalar')):
                out['f64_' + k] = v
        save('mbv2_' + tag, **out)


# ---------------------------------------------------------------------------
def make_targets():
    """datasets/coco.py:191-221 re-enacted with the reference's own utils/image.py functions (the dataset class
    itself needs pycocotools / imgaug / cv2; utils/image.py is imported with empty stand-ins for its unused
    cv2 / imgaug imports, see _load_entropy_map)."""
    _load_entropy_map()
    import importlib
    img = importlib.import_module('utils.image')
    out = {}
    for name, (C, H, W, M, n, seed) in gin.TARGET_CASES.items():
        boxes, classes = gin.target_boxes(name)
        hm = np.zeros((C, H, W), dtype=np.float32)
        wh = np.zeros((M, 2), dtype=np.float32)
        reg = np.zeros((M, 2), dtype=np.float32)
        ind = np.zeros((M), dtype=np.int64)
        reg_mask = np.zeros((M), dtype=np.uint8)
        gt_det = np.zeros((M, 6), dtype=np.float32)
        gt_areas = np.zeros((M), dtype=np.float32)
        for k in range(n):
            bbox = np.array(boxes[k])
            cls_id = int(classes[k])
            bbox[[0, 2]] = np.clip(bbox[[0, 2]], 0, W - 1)
            bbox[[1, 3]] = np.clip(bbox[[1, 3]], 0, H - 1)
            h, w = bbox[3] - bbox[1], bbox[2] - bbox[0]
            if h > 0 and w > 0:
                radius = img.gaussian_radius((np.ceil(h), np.ceil(w)))
                radius = max(0, int(radius))
                ct = np.array([(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2], dtype=np.float32)
                ct_int = ct.astype(np.int32)
                img.draw_umich_gaussian(hm[cls_id], ct_int, radius)
                wh[k] = 1. * w, 1. * h
                ind[k] = ct_int[1] * W + ct_int[0]
                reg[k] = ct - ct_int
                reg_mask[k] = 1
                gt_det[k] = ([ct[0] - w / 2, ct[1] - h / 2, ct[0] + w / 2, ct[1] + h / 2, 1, cls_id])
                gt_areas[k] = w * h
        for key, v in dict(hm=hm, wh=wh, reg=reg, ind=ind, reg_mask=reg_mask, gt_dets=gt_det, gt_areas=gt_areas).items():
            out['%s__%s' % (name, key)] = v
    save('targets', **out)


if __name__ == '__main__':
    oracle_dcn.build()
    which = set(sys.argv[1:]) or {'decode', 'losses', 'dla', 'step', 'advent', 'resnet', 'targets', 'mobilenetv2',
                                  'getdet', 'base', 'uda128'}
    if 'decode' in which:
        make_decode()
    if 'resnet' in which:
        make_resnet()
    if 'targets' in which:
        make_targets()
    if 'mobilenetv2' in which:
        make_mobilenetv2()
    if which & {'losses', 'advent', 'uda128'}:
        _load_entropy_map()
    if 'losses' in which:
        make_losses()
    if 'getdet' in which:
        make_getdet()
    if which & {'dla', 'step', 'advent', 'base', 'uda128'}:
        d = make_dla() if 'dla' in which else _import_reference_dla()
        if 'base' in which:
            make_base_step(d)
        if 'step' in which:
            make_step(d)
        if 'advent' in which:
            make_advent(d)
        if 'uda128' in which:
            make_uda_step128(d)
