"""Tensor helpers with the reference's names (utils/tensor.py:5-25)."""
import torch

import hip_runtime as hr


def _sigmoid(x):
    """clamp(sigmoid(x), 1e-4, 1-1e-4); `x` itself is overwritten with the
    unclamped sigmoid, as `x.sigmoid_()` does in the reference (tensor.py:5-7)."""
    from hip_runtime import ops
    return ops.sigmoid_clamp_(x)


def _gather_feat(feat, ind, mask=None):
    dim = feat.size(2)
    idx = ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim)
    feat = feat.gather(1, idx)
    if mask is not None:
        feat = feat[mask.unsqueeze(2).expand_as(feat)].view(-1, dim)
    return feat


def _transpose_and_gather_feat(feat, ind):
    """feat [B,ch,H,W], ind [B,M] -> [B,M,ch]"""
    from hip_runtime import ops
    return ops.gather_feat(feat, ind)
