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


class KPSL1Loss(torch.nn.Module):
    """Masked L1 on the keypoint offsets plus an L1 on the lengths of the keypoint pairs `kps_weight_indices`
    (losses/centernet.py:136-189), one fused kernel each way."""

    def __init__(self, weight=1.0, kps_weight_indices=None, distance_weight=0.1, use_l1=False):
        super().__init__()
        if weight is None:
            raise TypeError("KPSL1Loss: kp_weight is None (the reference fails on `loss *= None` as well, "
                            "losses/centernet.py:155)")
        self.weight = weight
        self.distance_weight = distance_weight
        self.kps_weight_indices = torch.tensor(kps_weight_indices) if kps_weight_indices else None
        self.use_l1 = use_l1
        self._pairs = None

    def forward(self, output, mask, ind, target, return_den=False):
        if self.kps_weight_indices is not None and (self._pairs is None or self._pairs.device != output.device):
            self._pairs = self.kps_weight_indices.to(device=output.device, dtype=torch.int32).contiguous()
            J = output.shape[1] // 2
            if self._pairs.dim() != 2 or self._pairs.shape[1] != 2 or int(self.kps_weight_indices.min()) < 0 or \
                    int(self.kps_weight_indices.max()) >= J:
                raise IndexError("KPSL1Loss: kps_weight_indices must be [P, 2] keypoint indices below %d" % J)
        return ops.kps_l1_loss(output, mask, ind, target, self._pairs, self.use_l1, self.weight,
                               self.distance_weight, return_den)


class DetectionLoss(torch.nn.Module):
    def __init__(self, hm_weight, wh_weight, off_weight, kp_weight=None, angle_weight=1.0, periodic=False,
                 kp_indices=None, kp_distance_weight=0.1, kp_distance_weight_l1=False):
        super().__init__()
        self.crit_hm = FocalLoss(weight=hm_weight)
        self.crit_reg = RegL1Loss(off_weight)
        self.crit_hw = PeriodicRegL1Loss(wh_weight, angle_weight) if periodic else RegL1Loss(wh_weight, angle_weight)
        self.with_keypoints = False
        if kp_weight is not None or kp_indices is not None:
            self.with_keypoints = True
            self.crit_kp = KPSL1Loss(kp_weight, kp_indices, kp_distance_weight, kp_distance_weight_l1)
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
        dens = [den_wh, den_off]
        parts = [wh_loss, off_loss]
        if self.with_keypoints:
            kp_loss, den_kp = self.crit_kp(output['kps'], batch['kp_reg_mask'], batch['ind'], batch['kps'],
                                           return_den=True)
            dens.append(den_kp)
            parts.append(kp_loss)
        # divisors of this rank: num_pos, and the mask sums (integers; the kernels hold sum + 1e-4)
        local = torch.stack([npos] + [torch.round(d - 1e-4) for d in dens])
        total = local.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        one = torch.ones_like(npos)
        c_hm = torch.where(npos > 0, npos, one) / torch.where(total[0] > 0, total[0], one)   # num_pos == 0: divisor 1
        # this rank's part of the global losses
        shares = torch.stack([hm_loss * c_hm] + [l * (d / (total[i + 1] + 1e-4)) for i, (l, d) in enumerate(zip(parts, dens))])
        loss = shares.sum() * float(world)
        glob = shares.detach().clone()
        dist.all_reduce(glob, op=dist.ReduceOp.SUM, group=group)
        stats = {'centernet_loss': glob.sum(), 'hm_loss': glob[0], 'wh_loss': glob[1], 'off_loss': glob[2]}
        if self.with_keypoints:
            stats['kp_loss'] = glob[3]
        return loss, stats

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
        stats = {'centernet_loss': loss, 'hm_loss': hm_loss, 'wh_loss': wh_loss, 'off_loss': off_loss}
        if self.with_keypoints:
            kp_loss = self.crit_kp(output['kps'], batch['kp_reg_mask'], batch['ind'], batch['kps'])
            loss = loss + kp_loss
            stats['centernet_loss'] = loss      # the reference's `loss += kp_loss` mutates the logged tensor too
            stats['kp_loss'] = kp_loss
        return loss, stats
