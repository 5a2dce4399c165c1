"""Entropy-minimisation UDA step (uda/entropy_minimization.py:6-43): two forwards
(source, target; both through train-mode BN, Q6), detection loss on the source,
weighted entropy loss on the target, two separate backward passes whose gradients
accumulate (Q5), one optimizer step.  The logged `entropy_loss` is the weighted
value because the weight is applied in place (Q4)."""
from uda.base import Model
from losses.entropy import EntropyLoss


class EntropyMinimization(Model):
    def __init__(self, entropy_weight):
        super().__init__()
        self.entropy_loss = EntropyLoss()
        self.entropy_weight = entropy_weight

    def step(self, data, is_training=True):
        self._to_device(data)
        if is_training:
            self.optimizer.zero_grad()
        outputs = {
            "source_domain": self.backend(data["input"]),
            "target_domain": self.backend(data["target_domain_input"]),
        }
        c_loss, c_stats = self.centernet_loss(outputs["source_domain"], data)
        e_loss, e_stats = self.entropy_loss(outputs["target_domain"], data)
        e_loss *= self.entropy_weight
        if is_training:
            with self._defer_sync():
                c_loss.backward()
            e_loss.backward()
            self._finish_backward()
            self.optimizer.step()
        stats = {**c_stats, **e_stats}
        stats["total_loss"] = c_loss + e_loss
        outputs["stats"] = self._detach_stats(stats)
        return outputs
