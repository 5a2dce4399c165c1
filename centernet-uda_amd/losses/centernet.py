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

    def from_logits(self, logits, target, return_den=False):
        return ops.focal_loss(logits, target, self.weight, return_den)

    def forward(self, out, target):
        raise RuntimeError("FocalLoss.forward on probabilities is not part of this build: DetectionLoss calls "
                           "from_logits (sigmoid + clamp + focal fused); see losses/centernet.py docstring")


class RegL1Loss(torch.nn.Module):
    def __init__(self, weight=1.0, angle_weight=1.0):
        super().__init__()
        self.weight, self.angle_weight = weight, angle_weight

    def forward(self, output, mask, ind, target, return_den=False):
        return ops.reg_l1_loss(output, mask, ind, target, False, self.weight, self.angle_weight, return_den)


class PeriodicRegL1Loss(torch.nn.Module):
    def __init__(self, wh_weight=1.0, angle_weight=1.0):
        super().__init__()
        self.wh_weight, self.angle_weight = wh_weight, angle_weight

    def forward(self, output, mask, ind, target, return_den=False):
        return ops.reg_l1_loss(output, mask, ind, target, True, self.wh_weight, self.angle_weight, return_den)


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
        self._global = None          # (process_group,) once use_global_normalizers() was called

    def use_global_normalizers(self, process_group=None):
        """One process per GPU: normalise like the reference's single-process DataParallel, whose loss sees the
        GATHERED batch of all devices -- focal by the global num_pos (losses/centernet.py:87-94), the L1 terms
        by the global mask sum (:120,130).  Each rank then returns `world_size` times its share of that global
        loss (its own numerator over the global divisor), so that the data-parallel wrapper's gradient AVERAGE
        is the gradient of the global loss; the stats hold the global values, as the reference logs them.
        Costs two all-reduces of three floats per step."""
        self._global = (process_group,)

    def _forward_global(self, output, batch):
        import torch.distributed as dist
        group = self._global[0]
        world = dist.get_world_size(group)
        hm_loss, prob, npos = self.crit_hm.from_logits(output['hm'], batch['hm'], return_den=True)
        output['hm'] = prob                                                    # Q1
        wh_loss, den_wh = self.crit_hw(output['wh'], batch['reg_mask'], batch['ind'], batch['wh'], return_den=True)
        off_loss, den_off = self.crit_reg(output['reg'], batch['reg_mask'], batch['ind'], batch['reg'], return_den=True)
        # divisors of this rank: num_pos, and the two mask sums (integers; the kernels hold sum + 1e-4)
        local = torch.stack([npos, torch.round(den_wh - 1e-4), torch.round(den_off - 1e-4)])
        total = local.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        one = torch.ones_like(npos)
        c_hm = torch.where(npos > 0, npos, one) / torch.where(total[0] > 0, total[0], one)   # num_pos == 0: divisor 1
        c_wh = den_wh / (total[1] + 1e-4)
        c_off = den_off / (total[2] + 1e-4)
        shares = torch.stack([hm_loss * c_hm, wh_loss * c_wh, off_loss * c_off])             # this rank's part of the global losses
        loss = shares.sum() * float(world)
        glob = shares.detach().clone()
        dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=group)
        g_loss = glob.sum()
        return loss, {'centernet_loss': g_loss, 'hm_loss': glob[0], 'wh_loss': glob[1], 'off_loss': glob[2]}

    def forward(self, output, batch):
        if self._global is not None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(self._global[0]) > 1:
                return self._forward_global(output, batch)
        hm_loss, prob = self.crit_hm.from_logits(output['hm'], batch['hm'])
        output['hm'] = prob                                                    # Q1
        wh_loss = self.crit_hw(output['wh'], batch['reg_mask'], batch['ind'], batch['wh'])
        off_loss = self.crit_reg(output['reg'], batch['reg_mask'], batch['ind'], batch['reg'])
        loss = hm_loss + wh_loss + off_loss
        return loss, {'centernet_loss': loss, 'hm_loss': hm_loss, 'wh_loss': wh_loss, 'off_loss': off_loss}
