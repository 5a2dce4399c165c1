"""Inference wrapper of the reference's export.py:19-56 on the MI355X kernels.

`CenterNet(backend, max_detections, is_rotated=False, nms=3)(x)` runs the backend without a tape and -- for the
DLA backend, in eval mode -- with every BatchNorm folded into the convolution in front of it (one launch per
conv + BN + skip connection + ReLU block: backends/dla.py `fold_batchnorm`), then the clamped sigmoid of the heat
map, the fused NMS + top-K decode and the `down_ratio` scaling, and returns `(boxes, scores, classes)` exactly like the reference's module:
boxes `[B, K, 4]` (`[x1, y1, x2, y2]` in input pixels) or `[B, K, 5]` (`[cx, cy, w, h, angle]`) for rotated
models.  The ONNX export / simplifier part of export.py is out of scope; `build_model` keeps
the reference's checkpoint lookup (`model_last.pth` / `model_best.pth` in the experiment folder).
"""
from importlib import import_module
from pathlib import Path

import torch
from torch import nn

from backends.decode import decode_detection
from hip_runtime import ops


class CenterNet(nn.Module):
    def __init__(self, backend, max_detections, is_rotated=False, nms=3):
        super().__init__()
        self.backend = backend
        self.max_detections = max_detections
        self.is_rotated = is_rotated
        self.nms = nms
        self.fold = True            # False: eval-mode BatchNorm kernels instead of folded weights
        self._fold_key = None

    def _folding(self):
        """Context for the backend call: BatchNorm-folded blocks when the backend offers them (backends/dla.py), this
        wrapper is in eval mode and `fold` is on; the folded copies are rebuilt whenever a parameter or a running
        statistic may have changed since they were made (hip_runtime.param_state_key)."""
        import contextlib
        import hip_runtime as hr
        target = getattr(self.backend, 'module', self.backend)
        mod = import_module(type(target).__module__)
        if self.training or not self.fold or not hasattr(mod, 'fold_batchnorm'):
            return contextlib.nullcontext()
        key = hr.param_state_key(target)
        if key != self._fold_key:
            mod.fold_batchnorm(target)
            self._fold_key = key
        return mod.folded_inference()

    @torch.no_grad()
    def forward(self, x):
        with self._folding():
            out = self.backend(x)
        has_kps = 'kps' in out
        dets = decode_detection(ops.sigmoid_clamp_(out['hm']), out['wh'], out['reg'],
                                kps=out['kps'] if has_kps else None, K=self.max_detections,
                                rotated=self.is_rotated, nms_size=self.nms)
        extra = ()
        if has_kps:
            dets, kps = dets
            kps[..., 0:2] *= self.backend.down_ratio
            extra = (kps,)
        dets[:, :, :4] *= self.backend.down_ratio
        if self.is_rotated:
            return (dets[:, :, :5], dets[:, :, 5], dets[:, :, 6]) + extra
        return (dets[:, :, :4], dets[:, :, 4], dets[:, :, 5]) + extra


def build_model(experiment, model_spec, without_decode_detections, max_detections, nms=3, use_last=True):
    """export.py:59-85: backend from `{'name', 'params'}`, weights from the experiment folder if present."""
    module = import_module("backends.%s" % model_spec['name'])
    backend = getattr(module, 'build')(**model_spec['params'])
    ckpt = Path(experiment) / ('model_last.pth' if use_last else 'model_best.pth')
    if ckpt.exists():
        checkpoint = torch.load(ckpt, map_location='cpu', weights_only=False)
        backend.load_state_dict(checkpoint['state_dict'])
        print("Restore weights %s successful!" % ckpt)
    else:
        print("No weights were found in folder %s" % experiment)
    if without_decode_detections:
        return backend
    return CenterNet(backend, max_detections, model_spec['params'].get('rotated_boxes', False), nms)
