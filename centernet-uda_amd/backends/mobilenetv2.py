"""MobileNetV2 CenterNet backend for MI355X.

Plugin contract of the reference (backends/mobilenetv2.py:168-186): `build(num_classes, num_keypoints=0,
pretrained=True, freeze_base=False, use_dcn=False, use_skip=False, rotated_boxes=False)` returns an nn.Module
with `.down_ratio == 4`, `.rotated_boxes` and `forward(x[B,3,H,W]) -> {'hm','wh','reg'[, 'kps']}` raw logits at
H/4.  Heads are registered in `sorted(heads)` order (mobilenetv2.py:75) and emitted in `heads` insertion order
(:115-117).

The trunk is `torch.hub.load('pytorch/vision:v0.6.0', 'mobilenet_v2').features` (mobilenetv2.py:31-36), a
third-party network the reference neither vendors nor pins by a test; its published architecture is restated
here so that the state_dict keys are the ones `mobilenet_v2.features` produces under the name `base`:

    base.0            ConvBNReLU(3, 32, 3x3 stride 2)           children 0 conv, 1 BN, 2 ReLU6
    base.1 .. base.17 InvertedResidual, settings (t, c, n, s) = (1,16,1,1) (6,24,2,2) (6,32,3,2) (6,64,4,2)
                      (6,96,3,1) (6,160,3,2) (6,320,1,1); `.conv` = [ConvBNReLU 1x1 expand (t != 1)],
                      ConvBNReLU 3x3 depthwise (stride), Conv2d 1x1 project, BN; residual iff stride 1 and in == out
    base.18           ConvBNReLU(320, 1280, 1x1)
    initialisation: kaiming_normal_(fan_out) for convolutions, BN weight 1 / bias 0

What the reference owns, restated from its file: three up-sampling stages `[DCN 3x3 + BN + ReLU]? +
ConvTranspose2d 4x4/2 pad 1 (no bias) + BN + ReLU` to 256 channels (:133-165; note that with `use_dcn` the first
DCN maps 1280 -> 256 and every ConvTranspose is 256 -> 256), optional 1x1 skip convolutions from trunk layers 13
and 6 added right after deconv-list entries 0 and 3 (:9-16, :66-72, :96-107 -- which only fits without `use_dcn`,
where those entries are the first two ConvTranspose layers; with both flags the reference fails on the shape
mismatch and so does this build), and per-head 3x3(256->64) + ReLU + 1x1 (:74-90, torch default init).

All layers run on this repo's gfx950 kernels: 1x1 / 3x3 convolutions on the implicit-GEMM MFMA kernels, the
depthwise 3x3 on `cnuda_dwconv2d_*`, BN (+ residual) + ReLU6 in one kernel, DCN on the deformable kernels.
"""
import math
import os

import torch
from torch import nn

from hip_runtime import nn as hnn
from hip_runtime import ops
from libs.DCNv2.dcn_v2 import DCN

# key = deconv layer index, value = feature extractor layer index (mobilenetv2.py:9-16)
SKIP_MAPPING = {3: 6, 0: 13}
SKIP_MAPPING_REVERSED = {v: k for k, v in SKIP_MAPPING.items()}
_SETTINGS = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
_PRETRAINED = 'mobilenet_v2-b0353104.pth'      # what torchvision 0.6 downloads into the hub checkpoint cache


class ConvBNReLU(nn.Sequential):
    """children '0' conv (dense or depthwise, no bias), '1' BN, '2' ReLU6 slot (fused into the BN kernel)"""

    def __init__(self, cin, cout, kernel_size=3, stride=1, groups=1):
        pad = (kernel_size - 1) // 2
        if groups == 1:
            conv = hnn.Conv2d(cin, cout, kernel_size, stride=stride, padding=pad, bias=False)
        elif groups == cin == cout:
            conv = hnn.DepthwiseConv2d(cin, kernel_size, stride=stride, padding=pad)
        else:
            raise NotImplementedError("grouped convolution other than depthwise")
        super().__init__(conv, hnn.BatchNorm2d(cout), hnn.Slot())
        self.out_channels = cout

    def forward(self, x):
        return self[1](self[0](x), relu=6)


class InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(ConvBNReLU(inp, hidden, kernel_size=1))
        layers += [ConvBNReLU(hidden, hidden, stride=stride, groups=hidden),
                   hnn.Conv2d(hidden, oup, 1, bias=False), hnn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        y = x
        for m in self.conv[:-2]:
            y = m(y)
        y = self.conv[-2](y)
        return self.conv[-1](y, residual=x if self.use_res_connect else None)      # BN + the residual add, no ReLU


def _features():
    layers = [ConvBNReLU(3, 32, stride=2)]
    cin = 32
    for t, c, n, s in _SETTINGS:
        for i in range(n):
            layers.append(InvertedResidual(cin, c, s if i == 0 else 1, t))
            cin = c
    layers.append(ConvBNReLU(cin, 1280, kernel_size=1))
    base = nn.Sequential(*layers)
    with torch.no_grad():
        for m in base.modules():
            if isinstance(m, (hnn.Conv2d, hnn.DepthwiseConv2d)):      # kaiming_normal_(mode='fan_out')
                k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
                fan_out = (m.weight.shape[0] * k * k)
                m.weight.normal_(0.0, math.sqrt(2.0 / fan_out))
    return base


class CenterMobileNetV2(nn.Module):
    def __init__(self, heads, pretrained, freeze_base=False, use_dcn=False, use_skip=False, rotated_boxes=False):
        super().__init__()
        head_conv = 64
        self.use_skip = use_skip
        self.use_dcn = use_dcn
        self.inplanes = 1280
        self.deconv_with_bias = False
        self.down_ratio = 4
        self.rotated_boxes = rotated_boxes
        self.base = _features()
        if pretrained:
            self._load_pretrained()
        if freeze_base:
            for p in self.base.parameters():
                p.requires_grad = False
        self.deconv_layer_channels = [256, 256, 256]
        self.deconv_layers = self._make_deconv_layer(3, self.deconv_layer_channels, [4, 4, 4], use_dcn)
        if self.use_skip:
            for deconv_id, fe_id in SKIP_MAPPING.items():
                in_channels = self.base[fe_id].conv[-2].out_channels
                out_channels = self.deconv_layers[deconv_id].out_channels
                setattr(self, "skip_%d" % deconv_id, hnn.Conv2d(in_channels, out_channels, 1, padding=0))
        self.heads = heads
        for head in sorted(self.heads):
            fc = hnn.Head(
                hnn.Conv2d(256, head_conv, 3, padding=1, bias=True, act_slope=0.0),
                hnn.Slot(),      # index of the reference's nn.ReLU (fused into conv '0')
                hnn.Conv2d(head_conv, self.heads[head], 1, bias=True))
            setattr(self, head, fc)

    def _load_pretrained(self):
        """The reference downloads torchvision's ImageNet weights through torch.hub (mobilenetv2.py:31-34); no
        network path here: the file is read from the hub checkpoint cache or this raises like a failed download."""
        path = os.path.join(torch.hub.get_dir(), 'checkpoints', _PRETRAINED)
        if not os.path.isfile(path):
            raise RuntimeError("mobilenet_v2 pretrained=True: %s not found (no download in this build; place "
                               "torchvision's checkpoint there or pass pretrained=False)" % path)
        tv = torch.load(path, map_location='cpu')
        self.base.load_state_dict({k[len('features.'):]: v for k, v in tv.items() if k.startswith('features.')})

    def _get_deconv_cfg(self, deconv_kernel, index):
        return {4: (4, 1, 0), 3: (3, 1, 1), 2: (2, 0, 0)}[deconv_kernel]      # kernel, padding, output_padding

    def _make_deconv_layer(self, num_layers, num_filters, num_kernels, use_dcn=False):
        assert num_layers == len(num_filters) == len(num_kernels)
        layers = []
        for i in range(num_layers):
            kernel, padding, output_padding = self._get_deconv_cfg(num_kernels[i], i)
            planes = num_filters[i]
            if use_dcn:
                layers += [DCN(self.inplanes, planes, kernel_size=(3, 3), stride=1, padding=1, dilation=1,
                               deformable_groups=1), hnn.BatchNorm2d(planes, momentum=0.1), hnn.Slot()]
            layers += [hnn.ConvTranspose2d(self.inplanes if not use_dcn else planes, planes, kernel, stride=2,
                                           padding=padding, output_padding=output_padding),
                       hnn.BatchNorm2d(planes, momentum=0.1), hnn.Slot()]
            self.inplanes = planes
        return nn.Sequential(*layers)

    def forward(self, x):
        skip = {}
        for lid, layer in enumerate(self.base):
            x = layer(x)
            if self.use_skip and lid in SKIP_MAPPING_REVERSED:
                skip[SKIP_MAPPING_REVERSED[lid]] = x
        d = self.deconv_layers
        for lid in range(0, len(d), 3):          # (conv-like, BN, ReLU) triples: BN + ReLU is one kernel
            x = d[lid](x)
            if lid in skip:                      # the skip lands between the convolution and its BN (:103-107)
                x = ops.add(getattr(self, "skip_%d" % lid)(skip[lid]), x)
            x = d[lid + 1](x, relu=True)
        return {head: getattr(self, head)(x) for head in self.heads}


def build(num_classes, num_keypoints=0, pretrained=True, freeze_base=False, use_dcn=False, use_skip=False,
          rotated_boxes=False):
    heads = {'hm': num_classes, 'wh': 2 if not rotated_boxes else 3, 'reg': 2}
    if num_keypoints > 0:
        heads['kps'] = num_keypoints * 2
    return CenterMobileNetV2(heads, pretrained=pretrained, freeze_base=freeze_base, use_dcn=use_dcn,
                             use_skip=use_skip, rotated_boxes=rotated_boxes)
