"""EntropyLoss plugin (losses/entropy.py:6-28): normalised softmax entropy of the
target-domain heat-map logits, one fused HIP reduction (forward) and one
elementwise kernel (backward)."""
import torch

from hip_runtime import ops


class EntropyLoss(torch.nn.Module):
    def __init__(self, eta=None):
        super().__init__()
        if eta is not None:
            raise NotImplementedError("eta-weighted entropy (losses/entropy.py:17-22) is only used by the FDA "
                                      "plugin, which is outside this build")
        self.eta = eta

    def forward(self, outputs, batch):
        loss = ops.entropy_loss(outputs['hm'])
        return loss, {'entropy_loss': loss}
