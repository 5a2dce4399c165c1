"""CPU: the target-encoding oracle against golden vectors produced with the reference's own utils/image.py
(gaussian_radius, draw_umich_gaussian) inside the loop of datasets/coco.py:191-221."""
import numpy as np
import pytest

import inputs as gin
from oracle import targets as ot


@pytest.mark.parametrize('name', sorted(gin.TARGET_CASES))
def test_encode_targets_golden(golden, name):
    g = golden('targets')
    C, H, W, M, n, _ = gin.TARGET_CASES[name]
    boxes, classes = gin.target_boxes(name)
    out = ot.encode_targets(boxes, classes, C, H, W, M)
    for key, v in out.items():
        want = g['%s__%s' % (name, key)]
        assert v.dtype == want.dtype and v.shape == want.shape, key
        np.testing.assert_array_equal(v, want, err_msg=key)          # same float64 arithmetic: bit-exact
    assert out['reg_mask'][1] == 0 and out['reg_mask'][2] == 0       # degenerate boxes leave gaps, not shifts
    assert (out['hm'] == 1.0).sum() >= 1
