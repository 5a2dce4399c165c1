"""Max-squares UDA plugin (uda/max_squares_minimization.py:6-50): detection loss on the source batch plus
`max_squares_weight` x MaxSquareLoss of the target batch (weight applied in place, Q4).  The step itself is
uda.base.Model.step_with_target_term."""
from losses.max_square import MaxSquareLoss
from uda.base import Model


class MaxSquaresMinimization(Model):
    def __init__(self, max_squares_weight):
        super().__init__()
        self.max_squares_weight = max_squares_weight
        self.max_squares_loss = MaxSquareLoss()

    def _target_term(self, target_outputs, data):
        loss, stats = self.max_squares_loss(target_outputs, data)
        loss *= self.max_squares_weight       # in place: `stats` holds the same tensor
        return loss, stats

    def criterion(self, outputs, batch):
        """(source loss, weighted target loss, merged stats): the reference's method of this name (:11-20)"""
        det_loss, stats = self.centernet_loss(outputs["source_domain"], batch)
        uda_loss, uda_stats = self._target_term(outputs["target_domain"], batch)
        return det_loss, uda_loss, dict(stats, **uda_stats)

    def step(self, data, is_training=True):
        return self.step_with_target_term(data, is_training, self._target_term)
