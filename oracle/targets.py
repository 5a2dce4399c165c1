"""TEST INFRASTRUCTURE -- CPU oracle for CenterNet target encoding.

numpy restatement of the per-image loop of datasets/coco.py:191-221 (axis-aligned boxes, no keypoints) and of
utils/image.py:8-57 (`gaussian_radius`, `gaussian2D`, `draw_umich_gaussian`), in the reference's arithmetic:
float64 box math and gaussians, float32 maps, `int()` truncation of the radius, centre index from the float32
centre.  Input boxes are the augmented, output-resolution boxes the loop starts from (`bbs_aug[k]` after
`clip_out_of_image`), so the oracle begins at the `np.clip` of coco.py:199-200.

Pinned by tests/golden/targets.npz generated with the reference's own utils/image.py functions.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import numpy as np


def gaussian_radius(det_size, min_overlap=0.7):                       # utils/image.py:8-30
    height, width = det_size
    b1 = height + width
    c1 = width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * c1)) / 2
    b2 = 2 * (height + width)
    c2 = (1 - min_overlap) * width * height
    r2 = (b2 + np.sqrt(b2 ** 2 - 16 * c2)) / 2
    a3 = 4 * min_overlap
    b3 = -2 * min_overlap * (height + width)
    c3 = (min_overlap - 1) * width * height
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def gaussian2d(diameter, sigma):                                      # utils/image.py:33-39
    m = (diameter - 1.0) / 2.0
    y, x = np.ogrid[-m:m + 1, -m:m + 1]
    h = np.exp(-(x * x + y * y) / (2 * sigma * sigma))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_gaussian(heatmap, center, radius):                           # utils/image.py:42-57 (k = 1)
    diameter = 2 * radius + 1
    g = gaussian2d(diameter, diameter / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape
    left, right = min(x, radius), min(width - x, radius + 1)
    top, bottom = min(y, radius), min(height - y, radius + 1)
    mh = heatmap[y - top:y + bottom, x - left:x + right]
    mg = g[radius - top:radius + bottom, radius - left:radius + right]
    if min(mg.shape) > 0 and min(mh.shape) > 0:
        np.maximum(mh, mg, out=mh)


def encode_targets(boxes, classes, num_classes, output_h, output_w, max_detections):
    """boxes [n, 4] float64 (x1, y1, x2, y2) in output-map pixels, classes [n] int -> dict of one image's
    targets with the schema of datasets/coco.py:242-251."""
    hm = np.zeros((num_classes, output_h, output_w), np.float32)
    wh = np.zeros((max_detections, 2), np.float32)
    reg = np.zeros((max_detections, 2), np.float32)
    ind = np.zeros(max_detections, np.int64)
    reg_mask = np.zeros(max_detections, np.uint8)
    gt_det = np.zeros((max_detections, 6), np.float32)
    gt_areas = np.zeros(max_detections, np.float32)
    for k in range(min(len(boxes), max_detections)):
        bbox = np.array(boxes[k], dtype=np.float64)
        cls_id = int(classes[k])
        bbox[[0, 2]] = np.clip(bbox[[0, 2]], 0, output_w - 1)
        bbox[[1, 3]] = np.clip(bbox[[1, 3]], 0, output_h - 1)
        h, w = bbox[3] - bbox[1], bbox[2] - bbox[0]
        if h > 0 and w > 0:
            radius = max(0, int(gaussian_radius((np.ceil(h), np.ceil(w)))))
            ct = np.array([(bbox[0] + bbox[2]) / 2, (bbox[1] + bbox[3]) / 2], dtype=np.float32)
            ct_int = ct.astype(np.int32)
            draw_gaussian(hm[cls_id], ct_int, radius)
            wh[k] = 1. * w, 1. * h
            ind[k] = ct_int[1] * output_w + ct_int[0]
            reg[k] = ct - ct_int
            reg_mask[k] = 1
            gt_det[k] = (ct[0] - w / 2, ct[1] - h / 2, ct[0] + w / 2, ct[1] + h / 2, 1, cls_id)
            gt_areas[k] = w * h
    return dict(hm=hm, reg_mask=reg_mask, ind=ind, wh=wh, reg=reg, gt_dets=gt_det, gt_areas=gt_areas)
