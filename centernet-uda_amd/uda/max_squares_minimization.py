"""Max-squares UDA step (uda/max_squares_minimization.py:6-50): like entropy
minimisation with MaxSquareLoss on the target domain (weight applied in place, Q4)."""
from uda.base import Model
from losses.max_square import MaxSquareLoss


class MaxSquaresMinimization(Model):
    def __init__(self, max_squares_weight):
        super().__init__()
        self.max_squares_loss = MaxSquareLoss()
        self.max_squares_weight = max_squares_weight

    def criterion(self, outputs, batch):
        s_loss, s_stats = self.centernet_loss(outputs["source_domain"], batch)
        t_loss, t_stats = self.max_squares_loss(outputs["target_domain"], batch)
        t_loss *= self.max_squares_weight
        return s_loss, t_loss, {**s_stats, **t_stats}

    def step(self, data, is_training=True):
        self._to_device(data)
        if is_training:
            self.optimizer.zero_grad()
        outputs = {
            "source_domain": self.backend(data["input"]),
            "target_domain": self.backend(data["target_domain_input"]),
        }
        s_loss, t_loss, stats = self.criterion(outputs, data)
        if is_training:
            with self._defer_sync():
                s_loss.backward()
            t_loss.backward()
            self._finish_backward()
            self.optimizer.step()
        stats["total_loss"] = s_loss + t_loss
        outputs["stats"] = self._detach_stats(stats)
        return outputs
