"""Comparison of a `Model.get_detections` result (uda/base.py:73-139) with the flat arrays stored by
tests/golden/make_golden.py::_store_detections."""
import numpy as np


def compare_detections(dets, g, prefix='', box_tol=0.0, exact_order=True, angle_tol=1e-4):
    """box_tol = 0: bit-equal boxes / scores (decode of identical head outputs is exact integer + one or two
    float32 additions).  The angle column of rotated boxes (sigmoid * 360 - 180, decode.py:64-66) goes through
    an exponential: angle_tol degrees."""
    want_keys = {'pred_boxes', 'pred_classes', 'pred_scores', 'gt_boxes', 'gt_classes', 'gt_ids', 'gt_areas'}
    if (prefix + 'pred_kps') in g.files:
        want_keys |= {'pred_kps', 'gt_kps'}
    assert set(dets) == want_keys, set(dets) ^ want_keys
    assert dets['pred_classes'].dtype == np.dtype(str(g[prefix + 'pred_classes_dtype']))
    if exact_order:
        np.testing.assert_array_equal(dets['pred_classes'], g[prefix + 'pred_classes'])
    for k in ('pred_boxes', 'pred_scores') + (('pred_kps',) if 'pred_kps' in dets else ()):
        got, want = np.asarray(dets[k]), g[prefix + k]
        assert got.shape == want.shape and got.dtype == want.dtype, (k, got.shape, want.shape, got.dtype)
        if exact_order:
            if box_tol == 0.0:
                if k == 'pred_boxes' and got.shape[-1] == 5:
                    assert np.abs(got[..., 4] - want[..., 4]).max() <= angle_tol
                    got, want = got[..., :4], want[..., :4]
                np.testing.assert_array_equal(got, want, err_msg=k)
            else:
                assert np.abs(got.astype(np.float64) - want).max() <= box_tol * max(1.0, np.abs(want).max()), k
    counts = [int(c) for c in g[prefix + 'gt_counts']]
    assert [len(b) for b in dets['gt_boxes']] == counts
    for k in ('gt_boxes', 'gt_classes', 'gt_areas') + (('gt_kps',) if 'gt_kps' in dets else ()):
        assert isinstance(dets[k], list) and len(dets[k]) == len(counts)
        got = np.concatenate([np.asarray(v) for v in dets[k]], 0)
        assert got.dtype == np.dtype(str(g[prefix + k + '_dtype'])), (k, got.dtype)
        np.testing.assert_array_equal(got, g[prefix + k], err_msg=k)
        for v, c in zip(dets[k], counts):
            assert len(v) == c
    np.testing.assert_array_equal(np.asarray(dets['gt_ids']), g[prefix + 'gt_ids'])
