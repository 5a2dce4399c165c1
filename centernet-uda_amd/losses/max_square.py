"""MaxSquareLoss plugin (losses/max_square.py:6-14): -mean(softmax(hm)^2) / 2."""
import torch

from hip_runtime import ops


class MaxSquareLoss(torch.nn.Module):
    def forward(self, outputs, batch):
        loss = ops.max_square_loss(outputs['hm'])
        return loss, {'max_square_loss': loss}
