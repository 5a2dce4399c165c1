"""Host logic of `uda.base.Model.get_detections` (reference uda/base.py:73-139) against the dicts the imported
reference class produced (tests/golden/getdet_*.npz): x down_ratio scaling, the reg_mask == 1 row selection, the
4/5 vs 5/6 column split for rotated boxes, the keypoint branch.  No GPU here: the decode kernel is replaced by
the numpy oracle for this test only (tests/test_gpu_dla.py runs the same fixtures through the HIP decode)."""
import types

import numpy as np
import pytest
import torch

import inputs as gin
from detections_check import compare_detections
from oracle import decode as odec

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _ns(**kw):
    return types.SimpleNamespace(**kw)


def _oracle_decode(heat, wh, reg=None, kps=None, K=100, rotated=False, nms_size=3):
    out = odec.decode_detection(heat.numpy(), wh.numpy(), None if reg is None else reg.numpy(), K=K, rotated=rotated,
                                nms_size=nms_size, kps=None if kps is None else kps.numpy())
    return tuple(T(o) for o in out) if kps is not None else T(out)


@pytest.mark.parametrize('name', sorted(gin.GETDET_CASES))
def test_get_detections_matches_the_reference_dict(golden, monkeypatch, name):
    import uda.base as ub
    monkeypatch.setattr(ub, 'decode_detection', _oracle_decode)
    g = golden('getdet_' + name)
    src, batch, K, rotated = gin.getdet_inputs(name)
    m = ub.Model()
    m.cfg = _ns(max_detections=K, model=_ns(backend=_ns(params=_ns(rotated_boxes=rotated))))
    m.backend = _ns(down_ratio=4)
    dets = m.get_detections({'source_domain': {k: T(v).clone() for k, v in src.items()}},
                            {k: T(v).clone() for k, v in batch.items()})
    compare_detections(dets, g)
