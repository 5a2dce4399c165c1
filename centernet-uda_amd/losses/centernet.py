"""Detection losses with the reference's plugin surface
(`losses.centernet.DetectionLoss(**cfg.model.backend.loss.params)`,
`forward(output, batch) -> (loss, stats)`; losses/centernet.py:7-56), computed by
fused HIP kernels: clamp(sigmoid) + focal in one pass with the `num_pos == 0`
branch taken on the device (no host sync, Q11), gather + masked L1 in one
kernel.

Side effects reproduced from the reference: `output['hm']` is rebound to the
clamped probabilities (Q1); `batch['wh']` / `batch['reg']` are masked in place
and a rotated, non-periodic angle target is replaced by its sigmoid (Q2); L1
denominators use the expanded mask (Q3).
"""
import torch

from hip_runtime import ops


class FocalLoss(torch.nn.Module):
    """CornerNet focal loss on probabilities (losses/centernet.py:59-95).  The
    product path feeds logits through `from_logits` so that sigmoid, clamp and
    loss share one kernel; calling the module on probabilities is not supported."""

    def __init__(self, weight=1.0):
        super().__init__()
        self.weight = weight

    def from_logits(self, logits, target):
        return ops.focal_loss(logits, target, self.weight)

    def forward(self, out, target):
        raise RuntimeError("FocalLoss.forward on probabilities is not part of this build: DetectionLoss calls "
                           "from_logits (sigmoid + clamp + focal fused); see losses/centernet.py docstring")


class RegL1Loss(torch.nn.Module):
    def __init__(self, weight=1.0, angle_weight=1.0):
        super().__init__()
        self.weight, self.angle_weight = weight, angle_weight

    def forward(self, output, mask, ind, target):
        return ops.reg_l1_loss(output, mask, ind, target, False, self.weight, self.angle_weight)


class PeriodicRegL1Loss(torch.nn.Module):
    def __init__(self, wh_weight=1.0, angle_weight=1.0):
        super().__init__()
        self.wh_weight, self.angle_weight = wh_weight, angle_weight

    def forward(self, output, mask, ind, target):
        return ops.reg_l1_loss(output, mask, ind, target, True, self.wh_weight, self.angle_weight)


class DetectionLoss(torch.nn.Module):
    def __init__(self, hm_weight, wh_weight, off_weight, kp_weight=None, angle_weight=1.0, periodic=False,
                 kp_indices=None, kp_distance_weight=0.1, kp_distance_weight_l1=False):
        super().__init__()
        if kp_weight is not None or kp_indices is not None:
            raise NotImplementedError("keypoint loss (KPSL1Loss, losses/centernet.py:136-189) is outside the "
                                      "hot path of this build")
        self.crit_hm = FocalLoss(weight=hm_weight)
        self.crit_reg = RegL1Loss(off_weight)
        self.crit_hw = PeriodicRegL1Loss(wh_weight, angle_weight) if periodic else RegL1Loss(wh_weight, angle_weight)
        self.with_keypoints = False

    def forward(self, output, batch):
        hm_loss, prob = self.crit_hm.from_logits(output['hm'], batch['hm'])
        output['hm'] = prob                                                    # Q1
        wh_loss = self.crit_hw(output['wh'], batch['reg_mask'], batch['ind'], batch['wh'])
        off_loss = self.crit_reg(output['reg'], batch['reg_mask'], batch['ind'], batch['reg'])
        loss = hm_loss + wh_loss + off_loss
        return loss, {'centernet_loss': loss, 'hm_loss': hm_loss, 'wh_loss': wh_loss, 'off_loss': off_loss}
