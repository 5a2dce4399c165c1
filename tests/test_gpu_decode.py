"""MI355X parity of backends.decode (HIP) against the golden vectors captured
from the reference and against the numpy oracle.  Indices are bit-exact."""
import numpy as np
import pytest
import torch

import inputs as gin
from oracle import decode as od

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _hip_decode(d, K=None):
    from backends import decode as hd
    dets, inds = hd._run(T(d['heat']), T(d['wh']), T(d['reg']), K or d['K'], d['rotated'], 3)
    torch.cuda.synchronize()
    return dets.cpu().numpy(), inds.cpu().numpy()


@pytest.mark.parametrize('name', sorted(gin.DECODE_CASES))
def test_decode_golden(golden, name):
    d = gin.decode_inputs(name)
    g = golden('decode_' + name)
    dets, inds = _hip_decode(d)
    assert np.array_equal(inds, g['inds'])
    cls_col = 6 if d['rotated'] else 5
    assert np.array_equal(dets[..., cls_col].astype(np.int32), g['clses'])
    assert np.array_equal(dets[..., cls_col - 1], g['dets'][..., cls_col - 1])       # scores exact
    np.testing.assert_allclose(dets, g['dets'], rtol=1e-5, atol=5e-5)               # fp32 tolerance 1e-4 budget


@pytest.mark.parametrize('name', ['small', 'noreg'])
def test_decode_keypoints_golden(golden, name):
    from backends.decode import decode_detection
    d = gin.decode_inputs(name)
    g = golden('decode_' + name)
    dets, kps = decode_detection(T(d['heat']), T(d['wh']), T(d['reg']), kps=T(gin.decode_kps_inputs(name)), K=d['K'],
                                 rotated=d['rotated'])
    np.testing.assert_allclose(dets.cpu().numpy(), g['dets'], rtol=1e-5, atol=5e-5)
    np.testing.assert_array_equal(kps.cpu().numpy(), g['kps'])          # a gather and one f32 add: bit-exact
    with pytest.raises(RuntimeError, match='kps'):
        decode_detection(T(d['heat']), T(d['wh']), T(d['reg']), kps=T(gin.decode_kps_inputs(name))[:, :3], K=d['K'])


def test_public_api_matches_oracle_and_has_reference_signature():
    from backends.decode import decode_detection, _nms, _topk
    d = gin.decode_inputs('small')
    out = decode_detection(T(d['heat']), T(d['wh']), reg=T(d['reg']), K=d['K'])
    want = od.decode_detection(d['heat'], d['wh'], d['reg'], K=d['K'])
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-6, atol=1e-5)
    nm = _nms(T(d['heat'])).cpu().numpy()
    assert np.array_equal(nm, od.nms(d['heat']))
    s, i, c, ys, xs = _topk(T(od.nms(d['heat'])), K=d['K'])
    os_, oi, oc, oy, ox = od.topk(od.nms(d['heat']), d['K'])
    assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(c.cpu().numpy(), oc)
    assert np.array_equal(s.cpu().numpy(), os_)
    assert np.array_equal(ys.cpu().numpy(), oy) and np.array_equal(xs.cpu().numpy(), ox)


@pytest.mark.parametrize('case', ['ties', 'constant', 'few_peaks', 'negative', 'nms5', 'big_plane'])
def test_decode_edge_cases_vs_oracle(case):
    rs = np.random.RandomState(7)
    B, C, H, W, K, nms = 2, 3, 20, 24, 50, 3
    heat = np.clip(1 / (1 + np.exp(-(rs.standard_normal((B, C, H, W)) - 2))), 1e-4, 1 - 1e-4).astype(np.float32)
    if case == 'ties':
        heat = np.round(heat * 8) / 8                      # heavy quantisation -> many equal scores
        heat = heat.astype(np.float32)
    elif case == 'constant':
        heat[:] = 0.25                                     # every cell is a "peak" with the same score
    elif case == 'few_peaks':
        heat[:] = 1e-4
        heat[0, 1, 3, 4] = 0.9
        heat[1, 2, 10, 10] = 0.7                           # fewer peaks than K -> suppressed zeros fill up
    elif case == 'negative':
        heat = rs.standard_normal((B, C, H, W)).astype(np.float32) * 3   # raw logits: Q9 literal formula
    elif case == 'nms5':
        nms = 5
    elif case == 'big_plane':
        B, C, H, W, K = 1, 2, 200, 190, 100                # 152 KB plane: the no-LDS-cache path
        heat = np.clip(1 / (1 + np.exp(-(rs.standard_normal((B, C, H, W)) - 2))), 1e-4, 1 - 1e-4).astype(np.float32)
    wh = rs.uniform(1, 9, (B, 2, H, W)).astype(np.float32)
    reg = rs.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
    from backends import decode as hd
    dets, inds = hd._run(T(heat), T(wh), T(reg), K, False, nms)
    want, winds, wcls = od.decode_detection(heat, wh, reg, K=K, nms_size=nms, return_inds=True)
    assert np.array_equal(inds.cpu().numpy(), winds)
    assert np.array_equal(dets[..., 5].cpu().numpy().astype(np.int32), wcls)
    np.testing.assert_allclose(dets.cpu().numpy(), want, rtol=1e-6, atol=1e-5)


def test_full_size_property_cfg3_scores_sorted_and_are_local_maxima():
    # BASELINE cfg sizes: B=16, C=6, 128x128, K=150 -- size-independent properties
    rs = np.random.RandomState(3)
    B, C, H, W, K = 16, 6, 128, 128, 150
    heat = np.clip(1 / (1 + np.exp(-(rs.standard_normal((B, C, H, W)) - 2.19))), 1e-4, 1 - 1e-4).astype(np.float32)
    wh = rs.uniform(1, 50, (B, 2, H, W)).astype(np.float32)
    reg = rs.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
    from backends import decode as hd
    dets, inds = hd._run(T(heat), T(wh), T(reg), K, False, 3)
    dets, inds = dets.cpu().numpy(), inds.cpu().numpy()
    assert (dets[:, :-1, 4] >= dets[:, 1:, 4]).all()                       # sorted
    cls = dets[..., 5].astype(np.int64)
    picked = np.take_along_axis(heat.reshape(B, C * H * W), cls * H * W + inds, axis=1)
    assert np.array_equal(picked, dets[..., 4])                            # scores are the map values
    nm = od.nms(heat).reshape(B, -1)
    kth = np.sort(nm, axis=1)[:, -K]
    assert np.array_equal(dets[:, -1, 4], kth)                             # the K-th score is the K-th largest
    # idempotence: decoding the already-suppressed map gives the same detections
    dets2, inds2 = hd._run(T(od.nms(heat)), T(wh), T(reg), K, False, 3)
    assert np.array_equal(inds2.cpu().numpy(), inds)


def test_errors():
    from backends.decode import decode_detection
    heat = torch.zeros(1, 1, 2, 2, device=DEV)
    with pytest.raises(RuntimeError):
        decode_detection(heat, torch.zeros(1, 2, 2, 2, device=DEV), K=5)          # K > H*W like torch.topk
    with pytest.raises(RuntimeError):
        decode_detection(torch.zeros(1, 1, 4, 4), torch.zeros(1, 2, 4, 4), K=2)   # CPU tensors: no fallback


def _trained_like_map(B, C, H, W, seed, peaks=(0, 40)):
    """What a trained model hands to the decode: background clamped to exactly 1e-4 (utils/tensor.py:5-7), so
    whole plateaus survive the NMS (hmax == heat), plus a few peaks -- planes with thousands of tied survivors and
    planes with fewer than K positive... all of which take the LDS-resident general path of the kernel."""
    rs = np.random.RandomState(seed)
    heat = np.full((B, C, H, W), 1e-4, np.float32)
    for b in range(B):
        for c in range(C):
            n = rs.randint(peaks[0], peaks[1] + 1)
            ys, xs = rs.randint(0, H, n), rs.randint(0, W, n)
            heat[b, c, ys, xs] = (rs.uniform(0.05, 0.95, n)).astype(np.float32)
    wh = rs.uniform(2, 40, (B, 2, H, W)).astype(np.float32)
    reg = rs.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
    return heat, wh, reg


@pytest.mark.parametrize('shape', [(2, 6, 128, 128, 150), (2, 3, 160, 160, 150), (1, 2, 40, 52, 100), (2, 4, 64, 64, 1),
                                   (1, 1, 33, 35, 64)])
def test_plateau_and_sparse_maps_take_the_lds_general_path(shape):
    from backends.decode import decode_detection
    B, C, H, W, K = shape
    heat, wh, reg = _trained_like_map(B, C, H, W, 31 + H)
    got = decode_detection(T(heat), T(wh), reg=T(reg), K=K).cpu().numpy()
    want = od.decode_detection(heat, wh, reg, K=K)
    assert np.array_equal(got[..., 4:], want[..., 4:])            # scores and classes: exact, ties by lowest index
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-5)


def test_noise_map_at_160_overflows_the_sorting_pool():
    # 25,600 pixels -> about 2,800 NMS survivors per plane (> the 2,048-key sorting pool): cfg5's map size
    from backends.decode import decode_detection
    rs = np.random.RandomState(77)
    heat = (1.0 / (1.0 + np.exp(-(rs.standard_normal((2, 6, 160, 160)) - 2.19)))).astype(np.float32)
    wh = rs.uniform(2, 40, (2, 3, 160, 160)).astype(np.float32)
    reg = rs.uniform(0, 1, (2, 2, 160, 160)).astype(np.float32)
    got = decode_detection(T(heat), T(wh), reg=T(reg), K=150, rotated=True).cpu().numpy()
    want = od.decode_detection(heat, wh, reg, K=150, rotated=True)
    assert np.array_equal(got[..., 5:], want[..., 5:])
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=2e-4)


@pytest.mark.parametrize('levels', [1, 2, 4, 8, 64, 0])
def test_isolated_peaks_sharing_score_bits_threshold_path(levels):
    """The pool path (K <= survivors <= 2,048) selects by a histogram of the top 12 score bits and rank-sorts the keys
    at or above the K-th key's bin.  1,849 isolated peaks (a 3-pixel lattice, zero background) whose scores take only
    `levels` distinct values (spread over [0.5, 0.95): 15 bins) put about 1,849 / levels tied keys into that bin: more than 1,024 (falls back to the pool
    sort), 513..1,024 (one lane per key), 257..512, <= 256; 0 = all scores distinct inside one bin."""
    from backends.decode import decode_detection
    rs = np.random.RandomState(91 + levels)
    B, C, H, W, K = 2, 3, 128, 128, 150
    heat = np.zeros((B, C, H, W), np.float32)
    ys, xs = np.meshgrid(np.arange(1, 128, 3), np.arange(1, 128, 3), indexing='ij')
    n = ys.size
    for b in range(B):
        for c in range(C):
            if levels:
                vals = (0.5 + 0.45 * rs.randint(0, levels, n) / levels).astype(np.float32)   # spread over 15 bins
            else:
                vals = (0.5 + rs.permutation(n) * 1e-6).astype(np.float32)      # all inside [0.5, 0.53125)
            heat[b, c, ys.ravel(), xs.ravel()] = vals
    wh = rs.uniform(2, 40, (B, 2, H, W)).astype(np.float32)
    reg = rs.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
    got = decode_detection(T(heat), T(wh), reg=T(reg), K=K).cpu().numpy()
    want = od.decode_detection(heat, wh, reg, K=K)
    assert np.array_equal(got[..., 4:], want[..., 4:])
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize('case', ['dominant', 'many_classes', 'constant80', 'seeds_overflow', 'survivors_overflow'])
def test_stage_two_seed_threshold_cases(case):
    """Stage 2 bounds the K-th largest candidate by the K-th largest of every class's ceil(K / C) best, then rank-sorts
    the classes' heads above that bound.  Shapes of the candidate set that stress it: one class holding every high
    score, more classes than K (one seed per class), tied scores across 80 classes, more seeds than the sorting buffer
    (falls back to the histogram selection), more survivors than the buffer (ditto)."""
    from backends.decode import decode_detection
    rs = np.random.RandomState({'dominant': 1, 'many_classes': 2, 'constant80': 3, 'seeds_overflow': 4,
                                'survivors_overflow': 5}[case])
    if case == 'dominant':
        B, C, H, W, K = 2, 40, 64, 64, 150
        heat = (rs.uniform(1e-4, 0.05, (B, C, H, W))).astype(np.float32)
        heat[:, 5] = rs.uniform(0.3, 0.99, (B, H, W)).astype(np.float32)
    elif case == 'many_classes':
        B, C, H, W, K = 1, 200, 32, 32, 50
        heat = rs.uniform(1e-4, 0.99, (B, C, H, W)).astype(np.float32)
    elif case == 'constant80':
        B, C, H, W, K = 1, 80, 32, 32, 100
        heat = np.full((B, C, H, W), 0.25, np.float32)
        heat[0, 7, 3, 4] = heat[0, 60, 9, 9] = 0.9
    elif case == 'seeds_overflow':
        B, C, H, W, K = 1, 600, 32, 32, 700                # ceil(700 / 600) * 600 = 1,200 seeds > 1,024
        heat = rs.uniform(1e-4, 0.99, (B, C, H, W)).astype(np.float32)
    else:
        # 50 weak classes (scores <= 0.2) pull the seed bound down; 10 strong ones then have all their 100 entries
        # above it: more than 1,024 survivors
        B, C, H, W, K = 1, 60, 32, 32, 100
        heat = rs.uniform(1e-4, 0.2, (B, C, H, W)).astype(np.float32)
        heat[:, ::6] = rs.uniform(0.5, 0.9, (B, 10, H, W)).astype(np.float32)
    wh = rs.uniform(2, 40, (B, 2, H, W)).astype(np.float32)
    reg = rs.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
    got = decode_detection(T(heat), T(wh), reg=T(reg), K=K).cpu().numpy()
    want = od.decode_detection(heat, wh, reg, K=K)
    assert np.array_equal(got[..., 4:], want[..., 4:])
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-5)


def test_negative_and_zero_scores_general_path():
    # not the reference's call path (scores are probabilities there), but the kernel defines it: Q9's literal formula
    from backends.decode import _topk
    rs = np.random.RandomState(5)
    x = rs.standard_normal((2, 3, 24, 28)).astype(np.float32)
    x[:, :, 5:9] = 0.0
    s, i, c, ys, xs = _topk(T(x), K=40)
    os_, oi, oc, oy, ox = od.topk(x, 40)
    assert np.array_equal(i.cpu().numpy(), oi) and np.array_equal(c.cpu().numpy(), oc)
    assert np.array_equal(s.cpu().numpy(), os_)


@pytest.mark.parametrize('kind', ['noise128', 'trained128', 'trained160', 'ties128', 'negative64', 'coco80_64'])
def test_row_bands_give_the_same_detections_bit_for_bit(kind):
    """Round 6: with few planes stage 1 cuts a plane into 2 or 4 bands of rows, one workgroup each (halo rows from the
    neighbouring bands, stage 2 merges classes x bands lists).  Every band count must give the detections of the
    one-workgroup-per-plane form bit for bit -- and those the oracle's -- on noise maps (fast path), trained-like maps
    (plateaus and fewer than K positives per band: the general path with halo pixels below every score), heavily tied
    scores (ties across bands resolve by pixel index), raw logits, and a class count that leaves no room for bands."""
    import hip_runtime as hr
    from backends import decode as hd
    rs = np.random.RandomState(len(kind) * 7 + 1)
    if kind == 'noise128':
        B, C, H, W, K = 3, 6, 128, 128, 150
        heat = np.clip(1 / (1 + np.exp(-(rs.standard_normal((B, C, H, W)) - 2.19))), 1e-4, 1 - 1e-4).astype(np.float32)
        wh = rs.uniform(1, 50, (B, 2, H, W)).astype(np.float32)
    elif kind == 'trained128':
        B, C, H, W, K = 2, 6, 128, 128, 150
        heat, wh, _ = _trained_like_map(B, C, H, W, 5)
    elif kind == 'trained160':
        B, C, H, W, K = 2, 3, 160, 160, 150
        heat, wh, _ = _trained_like_map(B, C, H, W, 6, peaks=(100, 400))
    elif kind == 'ties128':
        B, C, H, W, K = 2, 4, 128, 128, 150
        heat = (np.round(np.clip(1 / (1 + np.exp(-(rs.standard_normal((B, C, H, W)) - 2))), 1e-4, 1 - 1e-4) * 16) / 16).astype(np.float32)
        wh = rs.uniform(1, 50, (B, 2, H, W)).astype(np.float32)
    elif kind == 'negative64':
        B, C, H, W, K = 2, 3, 64, 64, 100
        heat = (rs.standard_normal((B, C, H, W)) * 3).astype(np.float32)
        heat[:, :, 20:30] = 0.0
        wh = rs.uniform(1, 50, (B, 2, H, W)).astype(np.float32)
    else:
        B, C, H, W, K = 8, 80, 64, 64, 100                   # 640 planes: no bands
        heat = rs.uniform(1e-4, 0.99, (B, C, H, W)).astype(np.float32)
        wh = rs.uniform(1, 50, (B, 2, H, W)).astype(np.float32)
    reg = rs.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
    L = hr.lib()
    res = []
    for bands in (1, 2, 4):
        prev = L.cnuda_decode_set_max_bands(bands)
        try:
            dets, inds = hd._run(T(heat), T(wh), T(reg), K, False, 3)
            torch.cuda.synchronize()
        finally:
            L.cnuda_decode_set_max_bands(prev)
        res.append((dets.cpu().numpy(), inds.cpu().numpy()))
    for d, i in res[1:]:
        assert np.array_equal(i, res[0][1]) and np.array_equal(d, res[0][0])
    want, winds, wcls = od.decode_detection(heat, wh, reg, K=K, return_inds=True)
    assert np.array_equal(res[2][1], winds)
    assert np.array_equal(res[2][0][..., 5].astype(np.int32), wcls)
    np.testing.assert_allclose(res[2][0], want, rtol=1e-6, atol=1e-5)
