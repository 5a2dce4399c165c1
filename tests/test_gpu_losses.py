"""MI355X parity of the loss plugins against golden vectors from the reference
and the CPU oracle, including the reference's in-place side effects (Q1, Q2)."""
import numpy as np
import pytest
import torch

import inputs as gin
from oracle import losses as ol

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _rel(a, b, tol):
    assert abs(float(a) - float(b)) <= tol * max(1.0, abs(float(b))), (float(a), float(b))


@pytest.mark.parametrize('name', sorted(gin.LOSS_CASES))
def test_detection_loss_golden(golden, name):
    from losses.centernet import DetectionLoss
    out_np, batch_np, w = gin.loss_inputs(name)
    g = golden('losses_det_' + name)
    leaves = {k: T(v).to(DEV).requires_grad_(True) for k, v in out_np.items()}
    out = dict(leaves)
    batch = {k: T(v).clone().to(DEV) for k, v in batch_np.items()}
    crit = DetectionLoss(**w)
    loss, stats = crit(out, batch)
    loss.backward()
    _rel(loss.item(), g['loss'], 1e-5)
    assert set(stats) == {'centernet_loss', 'hm_loss', 'wh_loss', 'off_loss'}
    for k, v in stats.items():
        _rel(v.item(), g['stat_' + k], 1e-5)
    # Q1: the dict entry now holds clamped probabilities
    np.testing.assert_allclose(out['hm'].detach().cpu().numpy(), g['hm_after'], rtol=1e-5, atol=1e-7)
    # Q2: batch targets masked (and, rotated non-periodic, angle sigmoided) in place
    np.testing.assert_allclose(batch['wh'].cpu().numpy(), g['wh_target_after'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(batch['reg'].cpu().numpy(), g['reg_target_after'], rtol=0, atol=0)
    for k in leaves:
        ref = g['grad_' + k]
        scale = max(1e-6, np.abs(ref).max())
        assert np.abs(leaves[k].grad.cpu().numpy() - ref).max() <= 1e-4 * scale, k


@pytest.mark.parametrize('name', sorted(gin.KPS_CASES))
def test_keypoint_detection_loss_golden(golden, name):
    from losses.centernet import DetectionLoss
    out_np, batch_np, w = gin.kps_inputs(name)
    g = golden('losses_kps_' + name)
    leaves = {k: T(v).to(DEV).requires_grad_(True) for k, v in out_np.items()}
    out = dict(leaves)
    batch = {k: T(v).clone().to(DEV) for k, v in batch_np.items()}
    crit = DetectionLoss(**w)
    assert crit.with_keypoints
    loss, stats = crit(out, batch)
    loss.backward()
    _rel(loss.item(), g['loss'], 1e-5)
    assert set(stats) == {'centernet_loss', 'hm_loss', 'wh_loss', 'off_loss', 'kp_loss'}
    for k, v in stats.items():
        _rel(v.item(), g['stat_' + k], 1e-5)
    np.testing.assert_allclose(batch['kps'].cpu().numpy(), g['kps_target_after'], rtol=0, atol=0)   # masked in place
    for k in leaves:
        ref = g['grad_' + k]
        scale = max(1e-6, np.abs(ref).max())
        assert np.abs(leaves[k].grad.cpu().numpy() - ref).max() <= 1e-4 * scale, k


def test_keypoint_loss_argument_errors():
    from losses.centernet import DetectionLoss, KPSL1Loss
    with pytest.raises(TypeError):
        DetectionLoss(1.0, 0.1, 1.0, kp_weight=None, kp_indices=[[0, 1]])       # `loss *= None` in the reference
    out_np, batch_np, w = gin.kps_inputs('nopairs')
    crit = KPSL1Loss(1.0, [[0, 7]])                                             # J = 3
    with pytest.raises(IndexError):
        crit(T(out_np['kps']).to(DEV), T(batch_np['kp_reg_mask']).to(DEV), T(batch_np['ind']).to(DEV),
             T(batch_np['kps']).to(DEV))
    with pytest.raises(RuntimeError, match='kp_reg_mask'):
        KPSL1Loss(1.0)(T(out_np['kps']).to(DEV), T(batch_np['kp_reg_mask']).float().to(DEV), T(batch_np['ind']).to(DEV),
                       T(batch_np['kps']).to(DEV))


def test_uda_losses_golden(golden):
    from losses.entropy import EntropyLoss
    from losses.max_square import MaxSquareLoss
    from losses.advent import AdventLoss
    from utils.image import entropy_map
    g = golden('losses_uda')
    for tag, mod, key in (('entropy', EntropyLoss(), 'entropy_loss'), ('maxsq', MaxSquareLoss(), 'max_square_loss')):
        x = T(g['hm']).to(DEV).requires_grad_(True)
        l, st = mod({'hm': x}, None)
        assert list(st) == [key] and st[key] is l
        l *= 0.3                                   # in-place weighting as the plugins do (Q4)
        l.backward()
        _rel(l.item(), 0.3 * float(g[tag + '_loss']), 1e-5)
        ref = 0.3 * g[tag + '_grad']
        assert np.abs(x.grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max()
    x = T(g['hm']).to(DEV).requires_grad_(True)
    em = entropy_map(x)
    em.sum().backward()
    np.testing.assert_allclose(em.detach().cpu().numpy(), g['entropy_map'], rtol=1e-4, atol=1e-7)
    ref = g['entropy_map_grad_of_sum']
    assert np.abs(x.grad.cpu().numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-3)
    adv = AdventLoss()
    for label in (0, 1):
        y = T(g['advent_logits']).to(DEV).requires_grad_(True)
        l, st = adv(y, label)
        l.backward()
        _rel(l.item(), g['advent_loss_%d' % label], 1e-5)
        np.testing.assert_allclose(y.grad.cpu().numpy(), g['advent_grad_%d' % label], rtol=1e-4, atol=1e-8)


def test_full_size_losses_vs_oracle_cfg3():
    # B=16, C=6, 128x128, M=150 (BASELINE cfg3): the oracle handles this size in well under a second
    from losses.centernet import DetectionLoss
    from losses.entropy import EntropyLoss
    B, C, H, W, M = 16, 6, 128, 128, 150
    batch_np = gin.detection_batch(B, C, H, W, M, tuple(1 + (i * 7) % 20 for i in range(B)), 2, 61)
    rs = np.random.RandomState(62)
    out_np = dict(hm=(rs.standard_normal((B, C, H, W)) - 2.19).astype(np.float32),
                  wh=(rs.standard_normal((B, 2, H, W)) * 5).astype(np.float32),
                  reg=rs.standard_normal((B, 2, H, W)).astype(np.float32))
    o_out = {k: T(v).clone().requires_grad_(True) for k, v in out_np.items()}
    o_loss, o_stats, _ = ol.detection_loss(o_out, {k: T(v) for k, v in batch_np.items()}, 1.0, 0.1, 1.0)
    o_ent = ol.entropy_loss(o_out['hm'])
    (o_loss + o_ent).backward()
    leaves = {k: T(v).to(DEV).requires_grad_(True) for k, v in out_np.items()}
    batch = {k: T(v).clone().to(DEV) for k, v in batch_np.items()}
    ent, _ = EntropyLoss()({'hm': leaves['hm']}, None)
    loss, stats = DetectionLoss(1.0, 0.1, 1.0)(dict(leaves), batch)
    (loss + ent).backward()
    _rel(loss.item(), o_loss.item(), 1e-5)
    _rel(ent.item(), o_ent.item(), 1e-5)
    for k in leaves:
        ref = o_out[k].grad.numpy()
        assert np.abs(leaves[k].grad.cpu().numpy() - ref).max() <= 1e-4 * np.abs(ref).max(), k


def test_sigmoid_helper_and_gather():
    from utils.tensor import _sigmoid, _transpose_and_gather_feat
    x = torch.tensor([-20.0, -1.0, 0.0, 3.0, 20.0], device=DEV)
    raw = x.clone()
    y = _sigmoid(x)
    assert torch.allclose(x.cpu(), torch.sigmoid(raw.cpu()))                 # in place, unclamped
    assert torch.allclose(y.cpu(), torch.clamp(torch.sigmoid(raw.cpu()), 1e-4, 1 - 1e-4))
    feat = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32, device=DEV).reshape(2, 3, 4, 5)
    ind = torch.tensor([[0, 7, 19], [5, 5, 1]], device=DEV)
    got = _transpose_and_gather_feat(feat, ind).cpu()
    want = feat.cpu().permute(0, 2, 3, 1).reshape(2, 20, 3).gather(1, ind.cpu()[:, :, None].expand(2, 3, 3))
    assert torch.equal(got, want)
