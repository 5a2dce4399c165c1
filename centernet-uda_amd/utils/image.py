"""`entropy_map` (utils/image.py:121-124): per-pixel, per-class normalised
entropy contribution -p*log2(p+1e-30)/log2(C) of softmax(hm), the
discriminator's input in the ADVENT plugin.  The image-augmentation helpers of
the reference's file are dataset-side code and not part of this build."""
from hip_runtime import ops


def entropy_map(hm):
    return ops.entropy_map(hm)
