"""Pins oracle/losses.py against the reference values in tests/golden/losses_*.npz."""
import numpy as np
import pytest
import torch

import inputs as gin
from oracle import losses as ol

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize('name', sorted(gin.LOSS_CASES))
def test_detection_loss(golden, name):
    out_np, batch_np, w = gin.loss_inputs(name)
    g = golden('losses_det_' + name)
    out = {k: T(v).clone().requires_grad_(True) for k, v in out_np.items()}
    batch = {k: T(v).clone() for k, v in batch_np.items()}
    loss, stats, prob = ol.detection_loss(out, batch, **w)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * max(1, abs(float(g['loss'])))
    for k, v in stats.items():
        assert abs(float(v) - float(g['stat_' + k])) <= 1e-5 * max(1, abs(float(g['stat_' + k]))), k
    np.testing.assert_allclose(prob.detach().numpy(), g['hm_after'], rtol=1e-6, atol=1e-7)
    for k in out:
        np.testing.assert_allclose(out[k].grad.numpy(), g['grad_' + k], rtol=1e-4, atol=1e-7)
    # oracle leaves inputs untouched; the reference's in-place masking result (Q2):
    m = batch_np['reg_mask'][:, :, None].astype(np.float32)
    np.testing.assert_allclose(batch_np['reg'] * m, g['reg_target_after'], rtol=0, atol=0)
    if name == 'plain' or name == 'nopos' or name == 'periodic':
        np.testing.assert_allclose(batch_np['wh'] * m, g['wh_target_after'], rtol=0, atol=0)


@pytest.mark.parametrize('name', sorted(gin.KPS_CASES))
def test_keypoint_detection_loss(golden, name):
    """DetectionLoss with the KPSL1Loss term (losses/centernet.py:136-189, :44-49)."""
    out_np, batch_np, w = gin.kps_inputs(name)
    g = golden('losses_kps_' + name)
    out = {k: T(v).clone().requires_grad_(True) for k, v in out_np.items()}
    batch = {k: T(v).clone() for k, v in batch_np.items()}
    det, stats, _ = ol.detection_loss(out, batch, hm_weight=w['hm_weight'], wh_weight=w['wh_weight'],
                                      off_weight=w['off_weight'])
    kp = ol.kps_l1(out['kps'], batch['kp_reg_mask'], batch['ind'], batch['kps'], w['kp_weight'], w['kp_indices'],
                   w['kp_distance_weight'], w['kp_distance_weight_l1'])
    loss = det + kp
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * max(1, abs(float(g['loss'])))
    assert abs(kp.item() - float(g['stat_kp_loss'])) <= 1e-5 * max(1, abs(float(g['stat_kp_loss'])))
    assert float(g['stat_centernet_loss']) == float(g['loss'])      # the reference's `loss += kp_loss` is in place
    for k in out:
        np.testing.assert_allclose(out[k].grad.numpy(), g['grad_' + k], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(batch_np['kps'] * batch_np['kp_reg_mask'], g['kps_target_after'], rtol=0, atol=0)


def test_uda_losses(golden):
    g = golden('losses_uda')
    for tag, fn in (('entropy', ol.entropy_loss), ('maxsq', ol.max_square_loss)):
        x = T(g['hm']).clone().requires_grad_(True)
        l = fn(x)
        l.backward()
        assert abs(l.item() - float(g[tag + '_loss'])) <= 1e-6 * max(1, abs(float(g[tag + '_loss'])))
        np.testing.assert_allclose(x.grad.numpy(), g[tag + '_grad'], rtol=1e-4, atol=1e-10)
    x = T(g['hm']).clone().requires_grad_(True)
    em = ol.entropy_map(x)
    em.sum().backward()
    np.testing.assert_allclose(em.detach().numpy(), g['entropy_map'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(x.grad.numpy(), g['entropy_map_grad_of_sum'], rtol=1e-4, atol=1e-7)
    for label in (0, 1):
        y = T(g['advent_logits']).clone().requires_grad_(True)
        l = ol.advent_loss(y, label)
        l.backward()
        assert abs(l.item() - float(g['advent_loss_%d' % label])) < 1e-6
        np.testing.assert_allclose(y.grad.numpy(), g['advent_grad_%d' % label], rtol=1e-5, atol=1e-9)
