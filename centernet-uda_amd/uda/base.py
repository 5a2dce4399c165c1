"""Baseline (no-UDA) training/eval step plugin -- the reference's `uda.base.Model`
(uda/base.py:10-157) on the MI355X kernels.

The driver injects `cfg, device, backend, optimizer, centernet_loss, scheduler`
(train.py:108-116) and calls `init_done, load_model, to, epoch_start, set_phase,
step, get_detections, epoch_end, save_model`.  Multi-GPU: the reference wraps the
backend in a single-process nn.DataParallel (utils/helper.py:75-80); here
`to(device, parallel=True)` wraps it in hip_runtime.parallel.DataParallel -- one
process per GPU, bucketed RCCL all-reduce of a flat gradient arena overlapped
with the rest of backward.
"""
import contextlib
import logging

import numpy as np
import torch

from backends.decode import decode_detection
from utils.helper import load_model, save_model

log = logging.getLogger(__name__)


class Model:
    # two-domain plugins: run source and target through the backend as one batch (backends/dla.py
    # DLASeg.forward_domains) when the backend offers it; False restores the reference's literal call sequence
    batch_domains = True
    target_grad_heads = ('hm',)        # heads whose target-domain output feeds a loss (entropy / max-squares: hm)

    def __init__(self):
        self.cfg = None
        self.backend = None
        self.optimizer = None
        self.centernet_loss = None
        self.device = None
        self.scheduler = None

    # -- lifecycle hooks -------------------------------------------------------
    def init_done(self):
        pass

    def epoch_start(self):
        pass

    def epoch_end(self):
        if self.scheduler is not None:
            self.scheduler.step()

    def set_phase(self, is_training=True):
        """train.py:150,171.  With a data-parallel backend the switch from training to evaluation is a COLLECTIVE
        (every rank's BatchNorm statistics := rank 0's, hip_runtime.parallel.DataParallel.sync_buffers: what the
        reference's nn.DataParallel does on every forward): all ranks must make it, also when only rank 0 validates."""
        was_training = self.backend.training
        self.backend.train(is_training)
        if was_training and not is_training and hasattr(self.backend, 'sync_buffers'):
            self.backend.sync_buffers()

    def to(self, device, parallel=False, global_normalizers=True):
        """parallel: one process per GPU.  global_normalizers (default): the detection loss divides by the
        num_pos / mask sums of ALL ranks' batches, as the reference's gathered single-process DataParallel loss
        does (losses/centernet.py DetectionLoss.use_global_normalizers); False: every rank normalises by its
        own batch (what torch DDP recipes do) and saves the two 12-byte all-reduces per step."""
        self.backend.to(device)
        if parallel:
            from hip_runtime.parallel import DataParallel
            self.backend = DataParallel(self.backend)
            if global_normalizers and hasattr(self.centernet_loss, 'use_global_normalizers'):
                self.centernet_loss.use_global_normalizers(self.backend.process_group)

    # -- helpers shared by the subclasses --------------------------------------------
    def _to_device(self, data):
        for k in data:
            data[k] = data[k].to(device=self.device, non_blocking=True)

    def _defer_sync(self, module=None):
        """Context for every backward() of a step but the last: gradients only
        accumulate locally; the all-reduce buckets fire during the last one."""
        module = self.backend if module is None else module
        return module.no_sync() if hasattr(module, 'no_sync') else contextlib.nullcontext()

    def _finish_backward(self, *modules):
        for m in (modules or (self.backend,)):
            if hasattr(m, 'finish_gradient_sync'):
                m.finish_gradient_sync()

    @staticmethod
    def _detach_stats(stats):
        # one device->host sync per step, where the reference has it (uda/base.py:51-52)
        return {k: v.cpu().detach() for k, v in stats.items()}

    # -- the step ----------------------------------------------------------------
    def criterion(self, outputs, batch):
        return self.centernet_loss(outputs["source_domain"], batch)

    def step(self, data, is_training=True):
        self._to_device(data)
        if is_training:
            self.optimizer.zero_grad()
        outputs = {"source_domain": self.backend(data["input"])}
        loss, stats = self.criterion(outputs, data)
        if is_training:
            loss.backward()
            self._finish_backward()
            self.optimizer.step()
        stats["total_loss"] = loss
        outputs["stats"] = self._detach_stats(stats)
        return outputs

    def step_with_target_term(self, data, is_training, target_term):
        """The step shared by the single-loss UDA plugins (uda/entropy_minimization.py:11-43,
        uda/max_squares_minimization.py:11-50): source and target batch both go through the backend in train
        mode (two BatchNorm updates, Q6); `target_term(target_outputs, data) -> (weighted loss, stats)`; the
        detection loss and the target term are back-propagated separately and their gradients add up (Q5), the
        gradient exchange of a parallel run fires during the second pass only.  With `batch_domains` (default) and
        a backend that offers `forward_domains`, the two forward passes are one pass over the concatenated batch and
        the two backward passes one pass of the summed loss -- same values up to floating-point summation order."""
        self._to_device(data)
        if is_training:
            self.optimizer.zero_grad()
        batched = (self.batch_domains and hasattr(self.backend, 'forward_domains')
                   and data["input"].shape == data["target_domain_input"].shape)
        if batched:
            # one pass over source | target (per-domain BatchNorm statistics, Q6), one backward pass of the summed
            # loss: the same gradients as the reference's two forward / two backward calls up to summation order
            src, tgt = self.backend.forward_domains(data["input"], data["target_domain_input"],
                                                    target_grad_heads=self.target_grad_heads)
            outputs = {"source_domain": src, "target_domain": tgt}
        else:
            outputs = {name: self.backend(data[key])
                       for name, key in (("source_domain", "input"), ("target_domain", "target_domain_input"))}
        det_loss, stats = self.centernet_loss(outputs["source_domain"], data)
        uda_loss, uda_stats = target_term(outputs["target_domain"], data)
        if is_training:
            if batched:
                (det_loss + uda_loss).backward()
            else:
                with self._defer_sync():
                    det_loss.backward()
                uda_loss.backward()
            self._finish_backward()
            self.optimizer.step()
        stats = dict(stats, **uda_stats)
        stats["total_loss"] = det_loss + uda_loss
        outputs["stats"] = self._detach_stats(stats)
        return outputs

    # -- evaluation --------------------------------------------------------------
    def get_detections(self, outputs, batch):
        """Decoded predictions and per-image ground truth as numpy (uda/base.py:73-139).
        `outputs['source_domain']['hm']` already holds clamped probabilities because the
        loss rebinds it (Q1)."""
        src = outputs["source_domain"]
        has_kps = 'kps' in src
        rotated = bool(self.cfg.model.backend.params.rotated_boxes)
        ratio = self.backend.down_ratio
        dets = decode_detection(src["hm"], src["wh"], src["reg"], kps=src["kps"] if has_kps else None,
                                K=self.cfg.max_detections, rotated=rotated)
        if has_kps:
            dets, kps = dets
            kps[..., 0:2] *= ratio
            kps = kps.detach().cpu().numpy()
        dets = dets.detach().cpu().numpy()
        dets[:, :, :4] *= ratio
        ids = batch["id"].cpu().numpy()
        keep = (batch["reg_mask"].detach().cpu().numpy() == 1)
        gt = batch["gt_dets"].cpu().numpy()
        areas = batch["gt_areas"].cpu().numpy()
        gt[:, :, :4] *= ratio
        box_end, cls_col = (5, 6) if rotated else (4, 5)
        out = {
            'pred_boxes': dets[:, :, :box_end],
            'pred_classes': dets[:, :, cls_col].astype(np.int32),
            'pred_scores': dets[:, :, box_end],
            'gt_boxes': [], 'gt_classes': [], 'gt_ids': [], 'gt_areas': [],
        }
        for i in range(gt.shape[0]):
            rows = gt[i, keep[i]]
            out['gt_boxes'].append(rows[:, :box_end])
            out['gt_classes'].append(rows[:, cls_col].astype(np.int32))
            out['gt_ids'].append(ids[i])
            out['gt_areas'].append(areas[i, keep[i]])
        if has_kps:
            kps_gt = batch['gt_kps'].cpu().numpy() * ratio
            out['gt_kps'] = [kps_gt[i, keep[i]] for i in range(gt.shape[0])]
            out['pred_kps'] = kps
        return out

    # -- checkpoints ---------------------------------------------------------------
    def load_model(self, path, resume=False):
        return load_model(self.backend, self.optimizer, self.scheduler, path, resume)

    def save_model(self, path, epoch, with_optimizer=False):
        if with_optimizer:
            save_model(self.backend, path, epoch, self.optimizer, self.scheduler)
        else:
            save_model(self.backend, path, epoch)
