"""Entropy-minimisation UDA plugin (uda/entropy_minimization.py:6-43): detection loss on the source batch plus
`entropy_weight` x EntropyLoss of the target batch's heat-map logits.  The weight is applied in place, so the logged
`entropy_loss` is the weighted value (Q4).  The step itself is uda.base.Model.step_with_target_term."""
from losses.entropy import EntropyLoss
from uda.base import Model


class EntropyMinimization(Model):
    def __init__(self, entropy_weight):
        super().__init__()
        self.entropy_weight = entropy_weight
        self.entropy_loss = EntropyLoss()

    def _target_term(self, target_outputs, data):
        loss, stats = self.entropy_loss(target_outputs, data)
        loss *= self.entropy_weight           # in place: `stats` holds the same tensor
        return loss, stats

    def step(self, data, is_training=True):
        return self.step_with_target_term(data, is_training, self._target_term)
