"""TEST INFRASTRUCTURE -- CPU oracle for backends/resnet.py (reference CenterResNet, configs[0]).

Two pieces:

* `forward(state, x, heads, training, num_layers)`: a functional float restatement on CPU torch tensors over a
  `state` dict keyed by the reference's state_dict names (`base.0.weight`, `base.5.0.downsample.1.running_var`,
  `deconv_layers.3.weight`, `hm.2.bias`, ...).  Restated from backends/resnet.py:
    trunk `nn.Sequential(*children[:-2])`  :27-30     deconv x3 (ConvT 4x4/2 p1, no bias + BN + ReLU) :66-98
    heads 3x3(256->64)+ReLU+1x1, emitted in `heads` order :43-58
* `torchvision_resnet(num_layers)`: the trunk as a torch.nn module whose `children()` are
  conv1, bn1, relu, maxpool, layer1..4, avgpool, fc -- what `torch.hub.load('pytorch/vision:v0.6.0', ...)`
  returns.  tests/golden/make_golden.py binds it to `torch.hub.load` so that the *reference's own*
  CenterResNet class can be imported and run to produce tests/golden/resnet18*.npz.

PARITY UNPINNED for the trunk: pytorch/vision v0.6.0 is a third-party dependency the reference fetches at
run time (resnet.py:27-28); it is not vendored, not installable here, and no reference test touches it.  The
trunk is restated from its published definition (BasicBlock [2,2,2,2] / [3,4,6,3]; Bottleneck with the stride
on the 3x3 and expansion 4; 7x7/2 stem, BN, ReLU, 3x3/2 max-pool pad 1; kaiming_normal_(fan_out) init).  The
fixtures pin what the reference itself owns: the children[:-2] slicing, the deconv stages, the heads and
their ordering, and every loss / step around them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import torch
import torch.nn.functional as F
from torch import nn

BLOCKS = {18: (2, 2, 2, 2), 34: (3, 4, 6, 3), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}
BN_MOMENTUM = 0.1
BN_EPS = 1e-5


# ---------------------------------------------------------------------------
# functional oracle
# ---------------------------------------------------------------------------
class _Net:
    def __init__(self, state, training):
        self.s, self.training = state, training

    def conv(self, name, x, stride=1, padding=0):
        return F.conv2d(x, self.s[name + '.weight'], self.s.get(name + '.bias'), stride, padding)

    def bn(self, name, x):
        s = self.s
        if self.training and (name + '.num_batches_tracked') in s:
            s[name + '.num_batches_tracked'] += 1
        return F.batch_norm(x, s[name + '.running_mean'], s[name + '.running_var'], s[name + '.weight'],
                            s[name + '.bias'], self.training, BN_MOMENTUM, BN_EPS)

    def basic(self, name, x, stride):
        identity = x
        if (name + '.downsample.0.weight') in self.s:
            identity = self.bn(name + '.downsample.1', self.conv(name + '.downsample.0', x, stride))
        y = F.relu(self.bn(name + '.bn1', self.conv(name + '.conv1', x, stride, 1)))
        y = self.bn(name + '.bn2', self.conv(name + '.conv2', y, 1, 1))
        return F.relu(y + identity)

    def bottleneck(self, name, x, stride):
        identity = x
        if (name + '.downsample.0.weight') in self.s:
            identity = self.bn(name + '.downsample.1', self.conv(name + '.downsample.0', x, stride))
        y = F.relu(self.bn(name + '.bn1', self.conv(name + '.conv1', x)))
        y = F.relu(self.bn(name + '.bn2', self.conv(name + '.conv2', y, stride, 1)))
        y = self.bn(name + '.bn3', self.conv(name + '.conv3', y))
        return F.relu(y + identity)


def forward(state, x, heads=('hm', 'wh', 'reg'), training=True, num_layers=18):
    """-> dict head -> raw logits [B, n, H/4, W/4] in `heads` order.  BN running statistics inside `state`
    are updated in place when training."""
    net = _Net(state, training)
    block = net.basic if num_layers in (18, 34) else net.bottleneck
    x = F.relu(net.bn('base.1', net.conv('base.0', x, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, n in enumerate(BLOCKS[num_layers]):
        for bi in range(n):
            x = block('base.%d.%d' % (4 + li, bi), x, 2 if (li > 0 and bi == 0) else 1)
    for i in (0, 3, 6):
        x = F.conv_transpose2d(x, state['deconv_layers.%d.weight' % i], None, 2, 1, 0)
        x = F.relu(net.bn('deconv_layers.%d' % (i + 1), x))
    out = {}
    for h in heads:
        y = F.relu(net.conv(h + '.0', x, 1, 1))
        out[h] = net.conv(h + '.2', y)
    return out


# ---------------------------------------------------------------------------
# torch.nn restatement of the hub trunk (golden generation only)
# ---------------------------------------------------------------------------
class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        out += identity
        return self.relu(out)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        out += identity
        return self.relu(out)


class _TVResNet(nn.Module):
    def __init__(self, num_layers, num_classes=1000):
        super().__init__()
        block = _BasicBlock if num_layers in (18, 34) else _Bottleneck
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        n = BLOCKS[num_layers]
        self.layer1 = self._make(block, 64, n[0], 1)
        self.layer2 = self._make(block, 128, n[1], 2)
        self.layer3 = self._make(block, 256, n[2], 2)
        self.layer4 = self._make(block, 512, n[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _make(self, block, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def torchvision_resnet(num_layers):
    return _TVResNet(num_layers)
