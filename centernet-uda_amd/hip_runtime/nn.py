"""nn.Module wrappers over hip_runtime.ops with the parameter / buffer names of
their torch.nn counterparts, so state_dicts stay interchangeable with the
reference's checkpoints (utils/helper.py:95-117)."""
import math

import torch
from torch import nn

from . import ops


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class Conv2d(nn.Module):
    """Dense convolution on the implicit-GEMM MFMA kernels; optional fused
    bias + ReLU / LeakyReLU epilogue (`act_slope`: <0 none, 0 ReLU, 0.2 leaky)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, act_slope=-1.0,
                 emit_stats=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.act_slope = float(act_slope)
        # the layer's only consumer is a BatchNorm2d: in training mode the GEMM's epilogue leaves per-channel sum / sum of
        # squares with the output (ops.conv2d, emit_stats) and the BatchNorm skips its own pass over it
        self.emit_stats = bool(emit_stats)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        from . import PackToken
        self._pack_token = PackToken()              # identity of these weights for the library's pack cache
        self.reset_parameters()

    def reset_parameters(self):
        # nn.Conv2d's default: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in)) for both tensors
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = 1.0 / math.sqrt(fan_in)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-bound, bound)

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.act_slope, self._pack_token,
                          emit_stats=self.emit_stats and self.training)

    def extra_repr(self):
        return '%d, %d, kernel_size=%s, stride=%s, padding=%s, bias=%s, act_slope=%g' % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding,
            self.bias is not None, self.act_slope)


class BatchNorm2d(nn.Module):
    """BatchNorm2d whose forward can absorb the residual add and ReLU that follow it."""

    def __init__(self, num_features, momentum=0.1, eps=1e-5):
        super().__init__()
        self.num_features, self.momentum, self.eps = num_features, momentum, eps
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    # defer_apply: set by an owner whose ONLY consumer of this module's output is a convolution that normalises on load
    # (ops.conv2d / cnuda_conv2d_norm_input_supported): in training mode the output tensor is then returned unwritten
    defer_apply = False

    def forward(self, x, residual=None, relu=False):
        if x.shape[1] != self.num_features:
            raise RuntimeError("BatchNorm2d: expected %d channels, got %d" % (self.num_features, x.shape[1]))
        return ops.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                  self.momentum, self.eps, residual, relu, self.num_batches_tracked,
                                  defer_apply=self.defer_apply and self.training)


class MaxPool2d(nn.Module):
    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        self.kernel_size = int(kernel_size)
        self.stride = self.kernel_size if stride is None else int(stride)
        self.padding = int(padding)

    def forward(self, x):
        return ops.max_pool2d(x, self.kernel_size, self.stride, self.padding)


class ConvTranspose2d(nn.Module):
    """nn.ConvTranspose2d(in, out, k, stride, padding, output_padding, bias=False); weight [in, out, k, k]
    (the up-sampling stages of CenterResNet, backends/resnet.py:86-94)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.output_padding = _pair(padding), _pair(output_padding)
        self.weight = nn.Parameter(torch.empty(in_channels, out_channels, *self.kernel_size))
        # nn.ConvTranspose2d's default init: kaiming_uniform(a=sqrt(5)) with fan_in = weight.size(1) * k * k
        bound = 1.0 / math.sqrt(out_channels * self.kernel_size[0] * self.kernel_size[1])
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)

    def forward(self, x):
        return ops.conv_transpose2d(x, self.weight, self.stride, self.padding, self.output_padding)


class Head(nn.Sequential):
    """A detection head: Conv2d(+ReLU), Slot (the index of the reference's nn.ReLU), Conv2d -- the reference's
    `nn.Sequential(nn.Conv2d(C, head_conv, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(head_conv, classes, 1))`
    (dla.py:474-483) with the same state_dict keys ('0.weight', '2.weight', ...).  While a tape is recorded the pair
    is one node (ops.conv_act_conv1x1: the hidden map's gradient in one pass); otherwise the layers run in turn."""

    def forward(self, x, lead=None):
        """lead: run the head on the first `lead` images of x only (backends.dla.DLASeg.forward_domains: the source half of
        a batched UDA step).  The fused node then takes all of x and leaves the leading images' share of its gradient
        where the other heads' shares are (hip_runtime.fanout) -- no slice, no zero-filled gradient, no sum."""
        # The fused node calls the kernels on the children's parameters directly, i.e. it REPLACES the children's
        # forward: it is taken only when that is unobservable and useful -- no forward (pre-)hooks on any child
        # (feature extraction, activation statistics, profilers) and something in the head or its input needs a gradient.
        if len(self) == 3 and isinstance(self[0], Conv2d) and isinstance(self[2], Conv2d) \
                and not any(m._forward_hooks or m._forward_pre_hooks for m in self) \
                and (x.requires_grad or any(p.requires_grad for p in self.parameters())) \
                and ops.conv_act_conv1x1_supported(x, self[0], self[2]):
            return ops.conv_act_conv1x1(x, self[0], self[2], lead)
        return super().forward(x if lead is None else x[:lead])


class DepthwiseConvTranspose2d(nn.Module):
    """nn.ConvTranspose2d(C, C, k, stride, padding, groups=C, bias=False); weight [C,1,k,k]."""

    def __init__(self, channels, kernel_size, stride, padding):
        super().__init__()
        self.channels, self.kernel_size, self.stride, self.padding = channels, kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(channels, 1, kernel_size, kernel_size))
        with torch.no_grad():
            self.weight.uniform_(-1.0 / kernel_size, 1.0 / kernel_size)

    def forward(self, x, skip=None):
        return ops.depthwise_conv_transpose2d(x, self.weight, self.stride, self.padding, skip)


class DepthwiseConv2d(nn.Module):
    """nn.Conv2d(C, C, k, stride, padding, groups=C, bias=False); weight [C,1,k,k] (torchvision MobileNetV2)."""

    def __init__(self, channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.channels, self.kernel_size, self.stride, self.padding = channels, kernel_size, stride, padding
        self.in_channels = self.out_channels = channels
        self.weight = nn.Parameter(torch.empty(channels, 1, kernel_size, kernel_size))
        with torch.no_grad():
            self.weight.uniform_(-1.0 / kernel_size, 1.0 / kernel_size)

    def forward(self, x):
        return ops.depthwise_conv2d(x, self.weight, self.stride, self.padding)


class Slot(nn.Module):
    """Parameter-free placeholder that keeps Sequential indices aligned with the
    reference's module lists (e.g. the ReLU at index 1 of a head, dla.py:476-483)."""

    def forward(self, x):
        return x
