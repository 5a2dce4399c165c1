"""AdventLoss plugin (losses/advent.py:5-18): BCE-with-logits of the
discriminator output against a constant domain label, mean reduction."""
import torch

from hip_runtime import ops


class AdventLoss(torch.nn.Module):
    def forward(self, y_pred, y_true):
        loss = ops.bce_with_logits_const(y_pred, float(y_true))
        return loss, {'advent_loss': loss}
