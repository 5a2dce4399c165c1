"""Tensor helpers with the reference's names (utils/tensor.py:5-25)."""
import torch

import hip_runtime as hr


def _sigmoid(x):
    """clamp(sigmoid(x), 1e-4, 1-1e-4); `x` itself is overwritten with the
    unclamped sigmoid, as `x.sigmoid_()` does in the reference (tensor.py:5-7)."""
    from hip_runtime import ops
    return ops.sigmoid_clamp_(x)


def _gather_feat(feat, ind, mask=None):
    """feat [B,N,ch], ind [B,M] -> rows ind of feat, [B,M,ch]; with a boolean `mask` [B,M] only the selected rows,
    flattened to [n,ch] (tensor.py:10-18).  Host-side helper on torch indexing; the kernels use
    `_transpose_and_gather_feat`."""
    rows = torch.take_along_dim(feat, ind[:, :, None], dim=1)
    return rows if mask is None else rows[mask.bool()]


def _transpose_and_gather_feat(feat, ind):
    """feat [B,ch,H,W], ind [B,M] -> [B,M,ch]"""
    from hip_runtime import ops
    return ops.gather_feat(feat, ind)
