"""CenterNet targets for a whole batch on the GPU.

`encode_targets(boxes, classes, counts, num_classes, output_h, output_w)` replaces the per-image numpy loop of
the reference's `__getitem__` (datasets/coco.py:168-174,191-221: gaussian splat into `hm`, `ind`, `wh`, `reg`,
`reg_mask`, `gt_dets`, `gt_areas`) with one kernel launch over all images, so that the batch the train step
consumes is produced where it is used.  Inputs are the augmented boxes at output resolution (what the loop
starts from): `boxes` [B, M, 4] (x1, y1, x2, y2; float64 like the reference's box arithmetic), `classes` [B, M]
int32 (already mapped through `cat_mapping`), `counts` [B] int32.  Returns the dict with the dataset's keys
and dtypes (`reg_mask` uint8, `ind` int64).  Rotated boxes and keypoints are not covered.
"""
import torch

from hip_runtime import check, lib, ptr, require_gpu, stream


def encode_targets(boxes, classes, counts, num_classes, output_h, output_w):
    require_gpu(boxes, classes, counts)
    if boxes.dim() != 3 or boxes.shape[2] != 4:
        raise RuntimeError("encode_targets: boxes must be [B, M, 4], got %s" % (tuple(boxes.shape),))
    B, M = boxes.shape[0], boxes.shape[1]
    if tuple(classes.shape) != (B, M) or tuple(counts.shape) != (B,):
        raise RuntimeError("encode_targets: classes %s / counts %s do not match boxes %s"
                           % (tuple(classes.shape), tuple(counts.shape), tuple(boxes.shape)))
    boxes = boxes.to(torch.float64).contiguous()
    classes = classes.to(torch.int32).contiguous()
    counts = counts.to(torch.int32).clamp(max=M).contiguous()
    dev = boxes.device
    out = {
        'hm': torch.empty((B, num_classes, output_h, output_w), dtype=torch.float32, device=dev),
        'reg_mask': torch.empty((B, M), dtype=torch.uint8, device=dev),
        'ind': torch.empty((B, M), dtype=torch.int64, device=dev),
        'wh': torch.empty((B, M, 2), dtype=torch.float32, device=dev),
        'reg': torch.empty((B, M, 2), dtype=torch.float32, device=dev),
        'gt_dets': torch.empty((B, M, 6), dtype=torch.float32, device=dev),
        'gt_areas': torch.empty((B, M), dtype=torch.float32, device=dev),
    }
    check(lib().cnuda_encode_targets(ptr(boxes), ptr(classes), ptr(counts), ptr(out['hm']), ptr(out['reg_mask']),
                                     ptr(out['ind']), ptr(out['wh']), ptr(out['reg']), ptr(out['gt_dets']),
                                     ptr(out['gt_areas']), B, num_classes, output_h, output_w, M, stream()),
          'encode_targets')
    return out
