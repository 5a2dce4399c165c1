"""`decode_detection` with the reference's signature (backends/decode.py:35-76),
executed by the fused NMS + top-K + box kernels of libcenternet_uda_hip.so.
`_nms` and `_topk` are exposed with the reference's names for callers that use
them separately."""
import torch

import hip_runtime as hr


def _nms(heat, kernel=3):
    hr.require_gpu(heat)
    heat = hr.f32c(heat)
    B, C, H, W = heat.shape
    out = torch.empty_like(heat)
    hr.check(hr.lib().cnuda_nms(hr.ptr(heat), hr.ptr(out), B, C, H, W, int(kernel), hr.stream()), '_nms')
    return out


def _run(heat, wh, reg, K, rotated, nms_size):
    hr.require_gpu(heat, wh, reg)
    heat, wh = hr.f32c(heat), hr.f32c(wh)
    reg = None if reg is None else hr.f32c(reg)
    B, C, H, W = heat.shape
    if K > H * W:
        raise RuntimeError("selected index k out of range")
    ncol = 7 if rotated else 6
    dets = torch.empty((B, K, ncol), dtype=torch.float32, device=heat.device)
    inds = torch.empty((B, K), dtype=torch.int64, device=heat.device)
    L = hr.lib()
    ws = hr.workspace(L.cnuda_decode_workspace_bytes(B, C, H, W, K), heat.device)
    hr.check(L.cnuda_decode_detection(hr.ptr(heat), hr.ptr(wh), hr.ptr(reg), hr.ptr(dets), hr.ptr(inds),
                                      B, C, H, W, int(K), wh.shape[1], 1 if rotated else 0, int(nms_size),
                                      hr.ptr(ws), ws.numel(), hr.stream()), 'decode_detection')
    return dets, inds


def _topk(scores, K=40):
    """(score, inds, clses, ys, xs) of the K best entries of an already-NMS'd map
    (decode.py:16-32).  nms_size=1 makes the fused kernel's NMS the identity."""
    B, C, H, W = scores.shape
    dummy = torch.zeros((B, 2, H, W), dtype=torch.float32, device=scores.device)
    dets, inds = _run(scores, dummy, dummy, K, False, 1)
    ys = torch.div(inds, W, rounding_mode='floor').float()
    xs = (inds % W).float()
    return dets[..., 4], inds, dets[..., 5].int(), ys, xs


def decode_detection(heat, wh, reg=None, kps=None, K=100, rotated=False, nms_size=3):
    """-> detections [B,K,6|7]; with `kps` [B,2J,H,W] also the decoded keypoints [B,K,J,2] (decode.py:69-74)."""
    dets, inds = _run(heat, wh, reg, K, rotated, nms_size)
    if kps is None:
        return dets
    hr.require_gpu(kps)
    kps = hr.f32c(kps)
    B, C, H, W = heat.shape
    if kps.dim() != 4 or kps.shape[0] != B or tuple(kps.shape[2:]) != (H, W) or kps.shape[1] % 2:
        raise RuntimeError("decode_detection: kps %s does not match heat %s" % (tuple(kps.shape), tuple(heat.shape)))
    J = kps.shape[1] // 2
    out = torch.empty((B, int(K), J, 2), dtype=torch.float32, device=heat.device)
    hr.check(hr.lib().cnuda_decode_keypoints(hr.ptr(kps), hr.ptr(None if reg is None else hr.f32c(reg)), hr.ptr(inds),
                                             hr.ptr(out), B, J, int(K), H, W, hr.stream()), 'decode_keypoints')
    return dets, out
