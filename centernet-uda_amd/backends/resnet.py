"""ResNet CenterNet backend for MI355X (configs[0] / SURVEY §8 row M6).

Plugin contract of the reference (backends/resnet.py:103-120): `build(num_layers,
num_classes, num_keypoints=0, pretrained=True, freeze_base=False,
rotated_boxes=False)` returns an nn.Module with `.down_ratio == 4`,
`.rotated_boxes` and `forward(x[B,3,H,W]) -> {'hm','wh','reg'[, 'kps']}` raw
logits at H/4.  Head modules are *registered* in `sorted(heads)` order
(resnet.py:43) and *emitted* in `heads` insertion order (resnet.py:58).

The trunk is the reference's `torch.hub.load('pytorch/vision:v0.6.0',
f'resnet{n}')` with its last two children (avgpool, fc) removed
(resnet.py:27-30).  That third-party code is not vendored by the reference and
not installable here; its published architecture is restated below so that
the state_dict keys are the ones `nn.Sequential(*children[:-2])` produces
(`base.0.weight`, `base.1.running_mean`, `base.5.0.downsample.0.weight`, ...):

    0 conv 7x7/2 (3->64, no bias)   1 BN   2 ReLU   3 MaxPool 3x3/2 pad 1
    4..7 layer1..layer4: BasicBlock x [2,2,2,2] (18) / [3,4,6,3] (34),
         Bottleneck (stride on the 3x3, expansion 4) x [3,4,6,3] / [3,4,23,3] / [3,8,36,3]
    initialisation: kaiming_normal_(fan_out, relu) for convolutions, BN weight 1 / bias 0

then three [ConvTranspose2d 4x4 /2 pad 1 (no bias) + BN + ReLU] stages to 256
channels (resnet.py:66-98) and per-head 3x3(256->64)+ReLU+1x1 (resnet.py:43-52,
torch default initialisation -- no -2.19 bias here).

All layers run on this repo's gfx950 kernels: the transposed convolution is
the implicit-GEMM input-gradient kernel (one launch per output parity class, 4
taps each), BN+residual+ReLU is one kernel.
"""
import math
import os

import torch
from torch import nn

from hip_runtime import nn as hnn

RESNET_MODELS = {18: 512, 34: 512, 50: 2048, 101: 2048, 152: 2048}     # resnet.py:6-12
_BLOCKS = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
# file names torchvision 0.6 downloads into the hub checkpoint cache
_PRETRAINED = {18: 'resnet18-5c106cde.pth', 34: 'resnet34-333f7ec4.pth', 50: 'resnet50-19c8e357.pth',
               101: 'resnet101-5d3b4d8f.pth', 152: 'resnet152-b121ed2d.pth'}


def _conv(cin, cout, k, stride=1):
    return hnn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False, emit_stats=True)    # (every caller: conv -> BatchNorm)


class _Downsample(nn.Sequential):
    """children '0' (1x1 conv, stride) and '1' (BN)"""

    def __init__(self, cin, cout, stride):
        super().__init__(_conv(cin, cout, 1, stride), hnn.BatchNorm2d(cout))

    def forward(self, x):
        return self[1](self[0](x))


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, planes, 3, stride), hnn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3), hnn.BatchNorm2d(planes)
        if downsample is not None:
            self.downsample = downsample

    def forward(self, x):
        identity = self.downsample(x) if hasattr(self, 'downsample') else x
        y = self.bn1(self.conv1(x), relu=True)
        return self.bn2(self.conv2(y), residual=identity, relu=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, planes, 1), hnn.BatchNorm2d(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3, stride), hnn.BatchNorm2d(planes)
        self.conv3, self.bn3 = _conv(planes, planes * 4, 1), hnn.BatchNorm2d(planes * 4)
        if downsample is not None:
            self.downsample = downsample

    def forward(self, x):
        identity = self.downsample(x) if hasattr(self, 'downsample') else x
        y = self.bn1(self.conv1(x), relu=True)
        y = self.bn2(self.conv2(y), relu=True)
        return self.bn3(self.conv3(y), residual=identity, relu=True)


def _trunk(num_layers):
    block = BasicBlock if num_layers in (18, 34) else Bottleneck
    children = [_conv(3, 64, 7, 2), hnn.BatchNorm2d(64), hnn.Slot(), hnn.MaxPool2d(3, 2, 1)]
    inplanes = 64
    for i, (planes, n) in enumerate(zip((64, 128, 256, 512), _BLOCKS[num_layers])):
        stride = 1 if i == 0 else 2
        down = None
        if stride != 1 or inplanes != planes * block.expansion:
            down = _Downsample(inplanes, planes * block.expansion, stride)
        blocks = [block(inplanes, planes, stride, down)]
        inplanes = planes * block.expansion
        blocks += [block(inplanes, planes) for _ in range(1, n)]
        children.append(nn.Sequential(*blocks))
    base = nn.Sequential(*children)
    with torch.no_grad():
        for m in base.modules():
            if isinstance(m, hnn.Conv2d):       # kaiming_normal_(mode='fan_out', nonlinearity='relu')
                fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1]
                m.weight.normal_(0.0, math.sqrt(2.0 / fan_out))
    return base


class _Deconv(nn.Sequential):
    """children: 3 x (ConvTranspose2d, BatchNorm2d, ReLU slot) = indices 0..8 (resnet.py:76-98)"""

    def forward(self, x):
        for i in range(0, len(self), 3):
            x = self[i + 1](self[i](x), relu=True)
        return x


class CenterResNet(nn.Module):
    def __init__(self, num_layers, heads, pretrained, freeze_base=False, rotated_boxes=False):
        super().__init__()
        head_conv = 64
        self.inplanes = RESNET_MODELS[num_layers]
        self.deconv_with_bias = False
        self.down_ratio = 4
        self.rotated_boxes = rotated_boxes
        self.base = _trunk(num_layers)
        if pretrained:
            self._load_pretrained(num_layers)
        if freeze_base:
            for p in self.base.parameters():
                p.requires_grad = False
        self.deconv_layers = self._make_deconv_layer(3, [256, 256, 256], [4, 4, 4])
        self.heads = heads
        for head in sorted(self.heads):
            fc = hnn.Head(
                hnn.Conv2d(256, head_conv, 3, padding=1, bias=True, act_slope=0.0),
                hnn.Slot(),      # index of the reference's nn.ReLU (fused into conv '0')
                hnn.Conv2d(head_conv, self.heads[head], 1, bias=True))
            setattr(self, head, fc)

    def _load_pretrained(self, num_layers):
        """The reference downloads torchvision's ImageNet weights through torch.hub (resnet.py:27-28).  There
        is no network path in this build: the file is taken from the hub checkpoint cache if present,
        otherwise this raises like a failed download does."""
        path = os.path.join(torch.hub.get_dir(), 'checkpoints', _PRETRAINED[num_layers])
        if not os.path.isfile(path):
            raise RuntimeError("resnet%d pretrained=True: %s not found (no download in this build; place "
                               "torchvision's checkpoint there or pass pretrained=False)" % (num_layers, path))
        tv = torch.load(path, map_location='cpu')
        names = ['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4']
        state = {}
        for k, v in tv.items():
            top, _, rest = k.partition('.')
            if top in names:
                state['%d.%s' % (names.index(top), rest)] = v
        self.base.load_state_dict(state)

    def _get_deconv_cfg(self, deconv_kernel, index):
        return {4: (4, 1, 0), 3: (3, 1, 1), 2: (2, 0, 0)}[deconv_kernel]      # kernel, padding, output_padding

    def _make_deconv_layer(self, num_layers, num_filters, num_kernels):
        assert num_layers == len(num_filters) == len(num_kernels)
        layers = []
        for i in range(num_layers):
            kernel, padding, output_padding = self._get_deconv_cfg(num_kernels[i], i)
            planes = num_filters[i]
            layers += [hnn.ConvTranspose2d(self.inplanes, planes, kernel, stride=2, padding=padding,
                                           output_padding=output_padding),
                       hnn.BatchNorm2d(planes, momentum=0.1), hnn.Slot()]
            self.inplanes = planes
        return _Deconv(*layers)

    def forward(self, x):
        b = self.base
        x = b[3](b[1](b[0](x), relu=True))
        for i in range(4, len(b)):
            x = b[i](x)
        x = self.deconv_layers(x)
        return {head: getattr(self, head)(x) for head in self.heads}


def build(num_layers, num_classes, num_keypoints=0, pretrained=True, freeze_base=False, rotated_boxes=False):
    assert num_layers in RESNET_MODELS.keys()
    heads = {'hm': num_classes, 'wh': 2 if not rotated_boxes else 3, 'reg': 2}
    if num_keypoints > 0:
        heads['kps'] = num_keypoints * 2
    return CenterResNet(num_layers, heads, pretrained=pretrained, freeze_base=freeze_base,
                        rotated_boxes=rotated_boxes)
