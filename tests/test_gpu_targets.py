"""MI355X: batched target encoding against the golden vectors of the reference's per-image loop
(datasets/coco.py:191-221 + utils/image.py) and against the oracle on a full-size batch."""
import numpy as np
import pytest
import torch

import inputs as gin
from oracle import targets as ot

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _check(got, want, what):
    for key in ('reg_mask', 'ind'):                                   # integer outputs: bit-exact
        np.testing.assert_array_equal(got[key], want[key], err_msg='%s %s' % (what, key))
    for key in ('wh', 'reg', 'gt_dets', 'gt_areas'):                  # same double arithmetic, cast once
        np.testing.assert_array_equal(got[key], want[key], err_msg='%s %s' % (what, key))
    # heat map: exp() in double on the device may differ from libm by an ulp of the float64 value; after the
    # cast to float32 that is at most one float32 ulp (tolerance of north_star: 1e-4)
    np.testing.assert_allclose(got['hm'], want['hm'], rtol=0, atol=1.2e-7, err_msg=what + ' hm')
    assert np.array_equal(got['hm'] == 1.0, want['hm'] == 1.0)          # the peaks the focal loss keys on


@pytest.mark.parametrize('name', sorted(gin.TARGET_CASES))
def test_encode_targets_golden(golden, name):
    from datasets import encode_targets
    g = golden('targets')
    C, H, W, M, n, _ = gin.TARGET_CASES[name]
    boxes, classes = gin.target_boxes(name)
    bb = np.zeros((2, M, 4)); cc = np.zeros((2, M), np.int32)
    bb[0, :n], cc[0, :n] = boxes, classes
    bb[1, :n - 1], cc[1, :n - 1] = boxes[1:], classes[1:]               # second image: shifted object list
    out = encode_targets(T(bb).to(DEV), T(cc).to(DEV), torch.tensor([n, n - 1], dtype=torch.int32, device=DEV), C, H, W)
    assert out['reg_mask'].dtype == torch.uint8 and out['ind'].dtype == torch.int64
    got0 = {k: v[0].cpu().numpy() for k, v in out.items()}
    _check(got0, {k: g['%s__%s' % (name, k)] for k in got0}, name)
    got1 = {k: v[1].cpu().numpy() for k, v in out.items()}
    _check(got1, ot.encode_targets(boxes[1:], classes[1:], C, H, W, M), name + ' image 1')


def test_full_size_batch_matches_oracle_and_feeds_the_loss():
    from datasets import encode_targets
    from losses.centernet import DetectionLoss
    rs = np.random.RandomState(5)
    B, C, H, W, M = 16, 6, 128, 128, 150
    counts = rs.randint(0, 21, B).astype(np.int32)
    bb = np.zeros((B, M, 4)); cc = np.zeros((B, M), np.int32)
    for b in range(B):
        n = counts[b]
        cx, cy = rs.uniform(0, W, n), rs.uniform(0, H, n)
        bw, bh = rs.uniform(2, 60, n), rs.uniform(2, 60, n)
        bb[b, :n] = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
        cc[b, :n] = rs.randint(0, C, n)
    out = encode_targets(T(bb).to(DEV), T(cc).to(DEV), T(counts).to(DEV), C, H, W)
    for b in (0, 7, 15):
        want = ot.encode_targets(bb[b, :counts[b]], cc[b, :counts[b]], C, H, W, M)
        _check({k: v[b].cpu().numpy() for k, v in out.items()}, want, 'image %d' % b)
    # the encoded batch is what DetectionLoss consumes
    pred = {'hm': torch.randn(B, C, H, W, device=DEV, requires_grad=True),
            'wh': torch.randn(B, 2, H, W, device=DEV, requires_grad=True),
            'reg': torch.randn(B, 2, H, W, device=DEV, requires_grad=True)}
    loss, stats = DetectionLoss(1.0, 0.1, 1.0)(pred, out)
    loss.backward()
    assert torch.isfinite(loss) and float(stats['hm_loss']) > 0
