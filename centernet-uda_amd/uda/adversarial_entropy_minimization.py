"""ADVENT adversarial entropy minimisation (uda/adversarial_entropy_minimization.py:13-189).

Sequence per step (reference :77-152): freeze D; forward source and target;
D(entropy_map(target logits)); detection loss on source + backward; "fool" loss
BCE(D(.), source_label) * adversarial_weight + backward; unfreeze D;
D(entropy_map(source.detach())) vs source_label, /2, backward;
D(entropy_map(target.detach())) vs target_label, /2, backward; both optimizers
step.  Quirks kept: the `source` map fed to D is the sigmoid-clamped
probabilities the detection loss left in the output dict, not logits (Q1);
`*=` / `/=` act in place on the logged tensors (Q4); the stats key
`dis_soruce` keeps the reference's spelling.

The discriminator is the reference's 5-layer conv4x4-s2 stack (:51-68) on this
repo's convolution kernels with LeakyReLU(0.2) fused into the epilogue; module
indices 0,2,4,6,8 match the reference's nn.Sequential for checkpoints.
"""
from pathlib import Path

import torch
from torch import nn

from hip_runtime import nn as hnn
from losses.advent import AdventLoss
from uda.base import Model
from utils.helper import load_model, save_model
from utils.image import entropy_map


class AdversarialEntropyMinimization(Model):
    def __init__(self, adversarial_weight, optimizer=None):
        super().__init__()
        self.adversarial_loss = AdventLoss()
        self.adversarial_weight = adversarial_weight
        self.source_label, self.target_label = 0, 1
        self.optimizer_settings = optimizer
        self.discriminator = None
        self.discriminator_optimizer = None
        self.discriminator_scheduler = None

    def init_done(self):
        self.discriminator = self.get_fc_discriminator(num_classes=self.cfg.model.backend.params.num_classes)
        from hip_runtime import optim as hoptim
        opt = self.optimizer_settings
        if opt is None:
            self.discriminator_optimizer = hoptim.Adam(self.discriminator.parameters())
            return
        self.discriminator_optimizer = hoptim.resolve(opt.name)(self.discriminator.parameters(), **dict(opt.params))
        sched = opt.get('scheduler') if hasattr(opt, 'get') else getattr(opt, 'scheduler', None)
        if sched is not None:
            cls = getattr(torch.optim.lr_scheduler, sched.name)
            self.discriminator_scheduler = cls(optimizer=self.discriminator_optimizer, **dict(sched.params))

    def get_fc_discriminator(self, num_classes, ndf=64):
        widths = [num_classes, ndf, ndf * 2, ndf * 4, ndf * 8]
        layers = []
        for cin, cout in zip(widths[:-1], widths[1:]):
            layers += [hnn.Conv2d(cin, cout, 4, stride=2, padding=1, act_slope=0.2), hnn.Slot()]
        layers.append(hnn.Conv2d(widths[-1], 1, 4, stride=2, padding=1))
        return nn.Sequential(*layers)

    def set_phase(self, is_training=True):
        super().set_phase(is_training)
        self.discriminator.train(is_training)

    def to(self, device, parallel=False, global_normalizers=True):
        super().to(device, parallel, global_normalizers)
        self.discriminator.to(device)
        if parallel:
            from hip_runtime.parallel import DataParallel
            self.discriminator = DataParallel(self.discriminator)

    def step(self, data, is_training=True):
        self._to_device(data)
        if is_training:
            self.optimizer.zero_grad()
            self.discriminator_optimizer.zero_grad()
        D = self.discriminator
        for p in D.parameters():
            p.requires_grad = False
        batched = (self.batch_domains and hasattr(self.backend, 'forward_domains')
                   and data["input"].shape == data["target_domain_input"].shape)
        if batched:
            # one pass over source | target (uda/base.py step_with_target_term explains the equivalence); the
            # target's heat map feeds the discriminator, its other heads feed nothing
            out_s, out_t = self.backend.forward_domains(data["input"], data["target_domain_input"],
                                                        target_grad_heads=self.target_grad_heads)
        else:
            out_s = self.backend(data["input"])
            out_t = self.backend(data["target_domain_input"])
        fool_logits = D(entropy_map(out_t["hm"]))
        outputs = {"source_domain": out_s, "target_domain": out_t}

        loss, stats = self.centernet_loss(out_s, data)
        if is_training and not batched:
            with self._defer_sync():
                loss.backward()
        dtf_loss, _ = self.adversarial_loss(fool_logits, self.source_label)
        dtf_loss *= self.adversarial_weight
        if is_training:
            if batched:
                (loss + dtf_loss).backward() # the two backward calls of the reference (:101,110) as one pass
            else:
                dtf_loss.backward()          # last backward that reaches the backbone
            self._finish_backward(self.backend)

        for p in D.parameters():
            p.requires_grad = True
        source = out_s["hm"].detach()        # sigmoid-clamped probabilities (Q1)
        target = out_t["hm"].detach()
        src_logits = D(entropy_map(source))
        ds_loss, _ = self.adversarial_loss(src_logits, self.source_label)
        ds_loss /= 2.0
        if is_training:
            with self._defer_sync(D):
                ds_loss.backward()
        dt_loss, _ = self.adversarial_loss(D(entropy_map(target)), self.target_label)
        dt_loss /= 2.0
        if is_training:
            dt_loss.backward()
            self._finish_backward(D)
        outputs['source_generator'] = src_logits
        outputs['target_generator'] = out_t          # sic (reference :136)
        if is_training:
            self.optimizer.step()
            self.discriminator_optimizer.step()
        stats["total_loss"] = loss + ds_loss + dt_loss + dtf_loss
        stats["dis_soruce"] = ds_loss
        stats["dis_target"] = dt_loss
        stats["dis_fool"] = dtf_loss
        outputs["stats"] = self._detach_stats(stats)
        return outputs

    def epoch_end(self):
        super().epoch_end()
        if self.discriminator_scheduler is not None:
            self.discriminator_scheduler.step()

    def save_model(self, path, epoch, with_optimizer=False):
        super().save_model(path, epoch, with_optimizer)
        if with_optimizer:
            save_model(self.discriminator, 'discriminator.pth', epoch, self.discriminator_optimizer,
                       self.discriminator_scheduler)
        else:
            save_model(self.discriminator, path, epoch)     # sic: same path as the backend (reference :179)

    def load_model(self, path, resume=False):
        d_weights = str(Path(path).with_name('discriminator.pth'))
        load_model(self.discriminator, self.discriminator_optimizer, self.discriminator_scheduler, d_weights, resume)
        return super().load_model(path, resume=resume)
