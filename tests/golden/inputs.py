"""Seeded input generators shared by make_golden.py (which runs the reference on
them, in the build container only) and by the tests (which run the oracle and
the HIP path on the same values).  numpy RandomState streams are stable across
numpy versions; parameters use an RNG-free closed form keyed by tensor name so
that no 80 MB state_dict has to be committed.
"""
import math
import zlib

import numpy as np


# ---------------------------------------------------------------------------
# decode
# ---------------------------------------------------------------------------
DECODE_CASES = {
    # name: (B, C, H, W, K, rotated, with_reg, seed)
    'small':   (2, 6, 32, 32, 20, False, True, 11),
    'cfg3':    (2, 6, 128, 128, 150, False, True, 12),
    'coco80':  (1, 80, 128, 128, 100, False, True, 13),
    'rotated': (2, 6, 40, 40, 30, True, True, 14),
    'noreg':   (1, 3, 24, 20, 16, False, False, 15),
}


def decode_inputs(name):
    B, C, H, W, K, rotated, with_reg, seed = DECODE_CASES[name]
    rs = np.random.RandomState(seed)
    logits = rs.standard_normal((B, C, H, W)).astype(np.float32) - np.float32(2.19)
    heat = (1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).astype(np.float32)
    heat = np.clip(heat, np.float32(1e-4), np.float32(1 - 1e-4))
    wh = (rs.uniform(2, 60, (B, 3 if rotated else 2, H, W))).astype(np.float32)
    if rotated:
        wh[:, 2] = rs.standard_normal((B, H, W)).astype(np.float32)
    reg = rs.uniform(0, 1, (B, 2, H, W)).astype(np.float32) if with_reg else None
    return dict(heat=heat, wh=wh, reg=reg, K=K, rotated=rotated)


# ---------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------
LOSS_CASES = {
    # name: (B, C, H, W, M, n_obj per image, wh channels, periodic, seed)
    'plain':    (2, 6, 16, 16, 8, (3, 5), 2, False, 21),
    'rotated':  (2, 6, 16, 16, 8, (2, 8), 3, False, 22),
    'periodic': (2, 6, 16, 16, 8, (4, 1), 3, True, 23),
    'nopos':    (2, 4, 12, 12, 6, (0, 0), 2, False, 24),
}


# keypoint heads (num_keypoints > 0): datasets/coco.py:183-189,226-227 schema
KPS_CASES = {
    # name: (B, C, H, W, M, n_obj per image, J, pairs, use_l1, seed)
    'pairs_l2': (2, 6, 16, 16, 8, (3, 5), 5, [[0, 1], [1, 2], [3, 4], [0, 4]], False, 61),
    'pairs_l1': (2, 6, 16, 16, 8, (2, 6), 4, [[0, 3], [2, 1]], True, 62),
    'nopairs':  (2, 6, 12, 12, 6, (4, 0), 3, None, False, 63),
}


def kps_inputs(name):
    """-> (head outputs incl. 'kps', batch incl. 'kps' / 'kp_reg_mask', DetectionLoss kwargs)."""
    B, C, H, W, M, n_obj, J, pairs, use_l1, seed = KPS_CASES[name]
    batch = detection_batch(B, C, H, W, M, n_obj, 2, seed)
    rs = np.random.RandomState(seed + 500)
    kp = rs.uniform(-12, 12, (B, M, 2 * J)).astype(np.float32)          # offsets from the centre, masked or not
    vis = (rs.uniform(0, 1, (B, M, J)) < 0.7)
    vis &= batch['reg_mask'][:, :, None].astype(bool)                   # invisible / absent keypoints
    batch['kps'] = kp
    batch['kp_reg_mask'] = np.repeat(vis, 2, axis=2).astype(np.uint8)
    out = dict(hm=(rs.standard_normal((B, C, H, W)) * 1.5 - 1.0).astype(np.float32),
               wh=(rs.standard_normal((B, 2, H, W)) * 3.0).astype(np.float32),
               reg=rs.standard_normal((B, 2, H, W)).astype(np.float32),
               kps=(rs.standard_normal((B, 2 * J, H, W)) * 8.0).astype(np.float32))
    weights = dict(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, kp_weight=0.5, kp_indices=pairs,
                   kp_distance_weight=0.3, kp_distance_weight_l1=use_l1)
    return out, batch, weights


def decode_kps_inputs(name):
    """Keypoint maps for the decode fixture `name` (same heat / wh / reg)."""
    B, C, H, W, K, rotated, with_reg, seed = DECODE_CASES[name]
    rs = np.random.RandomState(seed + 700)
    return (rs.standard_normal((B, 2 * 4, H, W)) * 6.0).astype(np.float32)


def gaussian_splat(hm, cx, cy, radius):
    """draw_umich_gaussian semantics (utils/image.py:40-57): max-merge of an
    exp(-(x^2+y^2)/(2 sigma^2)) patch, sigma = diameter/6, exact 1.0 at centre."""
    d = 2 * radius + 1
    sigma = d / 6.0
    ys, xs = np.ogrid[-radius:radius + 1, -radius:radius + 1]
    g = np.exp(-(xs * xs + ys * ys) / (2 * sigma * sigma)).astype(np.float32)
    g[g < np.finfo(np.float32).eps * g.max()] = 0
    H, W = hm.shape
    l, r = min(cx, radius), min(W - cx, radius + 1)
    t, b = min(cy, radius), min(H - cy, radius + 1)
    patch = hm[cy - t:cy + b, cx - l:cx + r]
    np.maximum(patch, g[radius - t:radius + b, radius - l:radius + r], out=patch)


def detection_batch(B, C, H, W, M, n_obj, wh_ch, seed):
    """Synthetic batch with the schema of datasets/coco.py:168-174,242-251."""
    rs = np.random.RandomState(seed)
    hm = np.zeros((B, C, H, W), np.float32)
    reg_mask = np.zeros((B, M), np.uint8)
    ind = np.zeros((B, M), np.int64)
    wh = np.zeros((B, M, wh_ch), np.float32)
    reg = np.zeros((B, M, 2), np.float32)
    for b in range(B):
        n = n_obj[b] if isinstance(n_obj, (tuple, list)) else int(n_obj)
        for k in range(n):
            cx, cy, c = rs.randint(0, W), rs.randint(0, H), rs.randint(0, C)
            gaussian_splat(hm[b, c], cx, cy, int(rs.randint(1, max(2, min(H, W) // 6))))
            reg_mask[b, k] = 1
            ind[b, k] = cy * W + cx
            wh[b, k, 0:2] = rs.uniform(2, 0.6 * W, 2)
            if wh_ch == 3:
                wh[b, k, 2] = rs.uniform(-90, 90)
            reg[b, k] = rs.uniform(0, 1, 2)
        # rows past n keep garbage-looking (but masked) targets, as a dataloader
        # could produce -- exercises the in-place masking (Q2)
        wh[b, n:] = rs.uniform(1, 5, (M - n, wh_ch))
        ind[b, n:] = rs.randint(0, H * W, M - n)
    return dict(hm=hm, reg_mask=reg_mask, ind=ind, wh=wh, reg=reg)


def loss_inputs(name):
    B, C, H, W, M, n_obj, wh_ch, periodic, seed = LOSS_CASES[name]
    batch = detection_batch(B, C, H, W, M, n_obj, wh_ch, seed)
    rs = np.random.RandomState(seed + 1000)
    out = dict(hm=(rs.standard_normal((B, C, H, W)) * 1.5 - 1.0).astype(np.float32),
               wh=(rs.standard_normal((B, wh_ch, H, W)) * 3.0).astype(np.float32),
               reg=rs.standard_normal((B, 2, H, W)).astype(np.float32))
    weights = dict(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=0.7, periodic=periodic)
    return out, batch, weights


# ---------------------------------------------------------------------------
# network parameters: closed-form, RNG-free fill keyed by the state_dict name
# ---------------------------------------------------------------------------
def _hash_uniform(name, n):
    """n pseudo-random float64 in [-1, 1): integer hash of (crc32(name), index).
    Pure uint64 arithmetic -> bit-identical on every platform."""
    seed = np.uint64((zlib.crc32(name.encode()) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF)
    x = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + seed
    x ^= x >> np.uint64(30)
    x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27)
    x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53) * 2.0 - 1.0


def fill_value(name, shape, offset_gain=1.0):
    """float32 array for state_dict entry `name` of `shape` (well-conditioned:
    kaiming-scaled pseudo-random kernels, BN statistics near identity).
    offset_gain scales the `conv_offset_mask` kernels: 1.0 gives DCN offsets of about +-1 px, at which the
    reference's own float32 result differs from its float64 result by several 1e-4 end to end (noise-like
    feature maps sampled at noisy positions); 0.1 keeps the sampling fractional (+-0.1 px) and brings that
    noise down to 1e-6 .. 3e-5, so that a plain 1e-4 comparison means something."""
    n = int(np.prod(shape)) if len(shape) else 1
    if name.endswith('num_batches_tracked'):
        return np.zeros(shape, np.int64)
    u = _hash_uniform(name, n)
    if name.endswith('running_var'):
        v = 1.0 + 0.3 * u
    elif name.endswith('running_mean'):
        v = 0.1 * u
    elif len(shape) == 1 and name.endswith('.weight'):          # BN gamma
        v = 1.0 + 0.2 * u
    elif len(shape) == 1:                                        # biases / BN beta
        v = 0.05 * u
        if name == 'hm.2.bias':
            v = v - 2.19
    else:                                                        # conv kernels: uniform with kaiming variance
        fan_in = int(np.prod(shape[1:]))
        scale = math.sqrt(6.0 / fan_in)
        if 'conv_offset_mask' in name:
            scale = 1.5 * offset_gain / math.sqrt(fan_in)        # offsets of about +-1 px (Q7)
        if '.up_' in name:
            scale = 2.0 / math.sqrt(fan_in)
        v = scale * u
    return v.reshape(shape).astype(np.float32)


def fill_state(shapes, offset_gain=1.0):
    """shapes: dict name -> shape.  Returns dict name -> np.ndarray."""
    return {k: fill_value(k, tuple(s), offset_gain) for k, s in shapes.items()}


def image_batch(B, H, W, seed):
    rs = np.random.RandomState(seed)
    return rs.standard_normal((B, 3, H, W)).astype(np.float32)


# ---------------------------------------------------------------------------
# target encoding: boxes at output resolution, some degenerate / out of range on purpose
# ---------------------------------------------------------------------------
TARGET_CASES = {
    # name: (num_classes, H, W, max_detections, n_boxes, seed)
    'cfg3': (6, 128, 128, 150, 20, 301),
    'wide': (3, 40, 160, 32, 32, 302),            # more boxes than slots are never passed: n <= M
    'tiny': (2, 8, 8, 6, 6, 303),
}


def target_boxes(name):
    C, H, W, M, n, seed = TARGET_CASES[name]
    rs = np.random.RandomState(seed)
    cx, cy = rs.uniform(-4, W + 4, n), rs.uniform(-4, H + 4, n)
    bw, bh = rs.uniform(0.5, 0.6 * W, n), rs.uniform(0.5, 0.6 * H, n)
    boxes = np.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
    boxes[1] = [W + 3.0, 2.0, W + 9.0, 5.0]        # completely outside: clipped to zero width -> skipped
    boxes[2, 2] = boxes[2, 0]                      # zero width
    classes = rs.randint(0, C, n).astype(np.int32)
    return boxes.astype(np.float64), classes


# ---------------------------------------------------------------------------
# evaluation-side batch keys (datasets/coco.py:168-174,242-251: id, gt_dets, gt_areas[, gt_kps])
# ---------------------------------------------------------------------------
def eval_extras(B, M, rotated, seed, num_keypoints=0):
    rs = np.random.RandomState(seed)
    cols = 7 if rotated else 6
    gt = rs.uniform(0, 30, (B, M, cols)).astype(np.float32)
    gt[..., cols - 2] = 1.0                                            # score column
    gt[..., cols - 1] = rs.randint(0, 6, (B, M))                       # class column
    out = dict(id=np.arange(100, 100 + B, dtype=np.int64), gt_dets=gt,
               gt_areas=rs.uniform(1, 900, (B, M)).astype(np.float32))
    if num_keypoints:
        out['gt_kps'] = rs.uniform(0, 30, (B, M, num_keypoints, 2)).astype(np.float32)
    return out


GETDET_CASES = {
    # name: (B, C, H, W, M, n_obj, K, rotated, num_keypoints, seed)
    'axis': (2, 6, 24, 20, 10, (3, 7), 12, False, 0, 81),
    'rot': (3, 4, 16, 16, 8, (8, 0, 2), 9, True, 0, 82),
    'kps': (2, 3, 16, 20, 6, (1, 4), 7, False, 4, 83),
}


def getdet_inputs(name):
    """-> (outputs['source_domain'] as the loss leaves it: clamped probabilities in 'hm' (Q1), raw wh / reg / kps;
    batch with reg_mask + the evaluation keys)."""
    B, C, H, W, M, n_obj, K, rotated, J, seed = GETDET_CASES[name]
    rs = np.random.RandomState(seed)
    logits = rs.standard_normal((B, C, H, W)).astype(np.float32) - np.float32(2.19)
    hm = np.clip((1.0 / (1.0 + np.exp(-logits.astype(np.float64)))).astype(np.float32), np.float32(1e-4), np.float32(1 - 1e-4))
    wh = rs.uniform(2, 0.6 * W, (B, 3 if rotated else 2, H, W)).astype(np.float32)
    if rotated:
        wh[:, 2] = rs.standard_normal((B, H, W)).astype(np.float32)
    src = dict(hm=hm, wh=wh, reg=rs.uniform(0, 1, (B, 2, H, W)).astype(np.float32))
    if J:
        src['kps'] = (rs.standard_normal((B, 2 * J, H, W)) * 5.0).astype(np.float32)
    batch = detection_batch(B, C, H, W, M, n_obj, 3 if rotated else 2, seed + 1)
    batch.update(eval_extras(B, M, rotated, seed + 2, J))
    return src, batch, K, rotated
