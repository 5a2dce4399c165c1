"""Pins oracle/decode.py against the reference outputs in tests/golden/decode_*.npz."""
import numpy as np
import pytest

import inputs as gin
from oracle import decode as od


@pytest.mark.parametrize('name', sorted(gin.DECODE_CASES))
def test_decode_matches_reference(golden, name):
    d = gin.decode_inputs(name)
    g = golden('decode_' + name)
    dets, inds, clses = od.decode_detection(d['heat'], d['wh'], d['reg'], K=d['K'],
                                            rotated=d['rotated'], return_inds=True)
    assert np.array_equal(inds, g['inds'])                      # bit-exact indices
    assert np.array_equal(clses, g['clses'])
    assert dets.shape == g['dets'].shape
    # scores / classes / integer coordinates are exact; box arithmetic within 1 ulp-ish
    np.testing.assert_allclose(dets, g["dets"], rtol=1e-5, atol=5e-5)  # angle = sigmoid*360-180 amplifies 1 ulp of exp()
    nm = od.nms(d['heat'])
    assert int((nm != 0).sum()) == int(g['nms_nonzero_count'])
    assert abs(float(nm.astype(np.float64).sum()) - float(g['nms_sum'])) < 1e-6 * max(1.0, abs(float(g['nms_sum'])))


@pytest.mark.parametrize('name', ['small', 'noreg'])
def test_decode_keypoints_match_reference(golden, name):
    d = gin.decode_inputs(name)
    g = golden('decode_' + name)
    dets, kps = od.decode_detection(d['heat'], d['wh'], d['reg'], K=d['K'], rotated=d['rotated'],
                                    kps=gin.decode_kps_inputs(name))
    assert kps.shape == g['kps'].shape == (dets.shape[0], d['K'], 4, 2)
    np.testing.assert_array_equal(kps, g['kps'])                # one gather and one f32 add per value: exact


def test_tie_order_is_score_desc_index_asc():
    heat = np.zeros((1, 2, 4, 4), np.float32)
    heat[0, 1, 1, 1] = 0.5
    heat[0, 0, 2, 2] = 0.5
    heat[0, 0, 0, 3] = 0.9
    wh = np.ones((1, 2, 4, 4), np.float32)
    dets, inds, clses = od.decode_detection(heat, wh, None, K=4, return_inds=True)
    assert inds[0, 0] == 3 and clses[0, 0] == 0
    # equal 0.5 scores: class 0 (flat index smaller) first
    assert (clses[0, 1], inds[0, 1]) == (0, 10)
    assert (clses[0, 2], inds[0, 2]) == (1, 5)
    # then suppressed zeros, lowest flat index first
    assert (clses[0, 3], inds[0, 3]) == (0, 0)


def test_k_larger_than_plane_raises():
    heat = np.zeros((1, 1, 2, 2), np.float32)
    with pytest.raises(RuntimeError):
        od.decode_detection(heat, np.zeros((1, 2, 2, 2), np.float32), None, K=5)
