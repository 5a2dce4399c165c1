"""TEST INFRASTRUCTURE -- CPU oracle for backends/mobilenetv2.py (reference CenterMobileNetV2).

Two pieces:

* `forward(state, x, heads, training, use_dcn, use_skip)`: a functional float restatement on CPU torch tensors
  over a `state` dict keyed by the reference's state_dict names (`base.0.0.weight`, `base.3.conv.1.1.running_var`,
  `deconv_layers.0.conv_offset_mask.weight`, `skip_3.bias`, `hm.2.bias`, ...).  Restated from
  backends/mobilenetv2.py: trunk `mobilenet_v2.features` :31-36, skip wiring :9-16,66-72,96-107, the deconv
  stages ([DCN 3x3 + BN + ReLU] + ConvT 4x4/2 p1 no bias + BN + ReLU) :133-165, heads :74-90,115-117.  The
  deformable convolution is oracle/dcn.py (libs/DCNv2/dcn_v2.py:84-128: offset/mask conv, chunk/cat, sigmoid).
* `torchvision_mobilenet_v2()`: the trunk as a torch.nn module with a `.features` Sequential -- what
  `torch.hub.load('pytorch/vision:v0.6.0', 'mobilenet_v2')` returns.  tests/golden/make_golden.py binds it to
  `torch.hub.load` so that the *reference's own* CenterMobileNetV2 class can be imported and run to produce
  tests/golden/mbv2_*.npz.

PARITY UNPINNED for the trunk: pytorch/vision v0.6.0 is a third-party dependency the reference fetches at run
time; it is not vendored, not installable here, and no reference test touches it.  The trunk is restated from
its published definition (ConvBNReLU = conv, BN, ReLU6; InvertedResidual with settings (t, c, n, s) =
(1,16,1,1) (6,24,2,2) (6,32,3,2) (6,64,4,2) (6,96,3,1) (6,160,3,2) (6,320,1,1); 1x1 to 1280; kaiming_normal_
fan_out).  The fixtures pin what the reference itself owns: the `.features` slicing, the skip and deconv wiring,
the heads and their ordering.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import torch
import torch.nn.functional as F
from torch import nn

SETTINGS = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
SKIP_MAPPING = {3: 6, 0: 13}
BN_MOMENTUM, BN_EPS = 0.1, 1e-5


def block_plan():
    """[(name, inp, oup, stride, expand)] for base.1 .. base.17"""
    plan, cin, idx = [], 32, 1
    for t, c, n, s in SETTINGS:
        for i in range(n):
            plan.append(('base.%d' % idx, cin, c, s if i == 0 else 1, t))
            cin, idx = c, idx + 1
    return plan


class _Net:
    def __init__(self, state, training):
        self.s, self.training = state, training

    def conv(self, name, x, stride=1, padding=0, groups=1):
        return F.conv2d(x, self.s[name + '.weight'], self.s.get(name + '.bias'), stride, padding, 1, groups)

    def bn(self, name, x):
        s = self.s
        if self.training and (name + '.num_batches_tracked') in s:
            s[name + '.num_batches_tracked'] += 1
        return F.batch_norm(x, s[name + '.running_mean'], s[name + '.running_var'], s[name + '.weight'],
                            s[name + '.bias'], self.training, BN_MOMENTUM, BN_EPS)

    def cbr(self, name, x, k, stride=1, groups=1):
        return F.relu6(self.bn(name + '.1', self.conv(name + '.0', x, stride, (k - 1) // 2, groups)))

    def dcn(self, name, x):
        from oracle import dcn as oracle_dcn
        om = self.conv(name + '.conv_offset_mask', x, 1, 1)
        o1, o2, mask = torch.chunk(om, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        return oracle_dcn.dcn_v2_conv(x, offset, torch.sigmoid(mask), self.s[name + '.weight'], self.s[name + '.bias'],
                                      1, 1, 1, 1)


def forward(state, x, heads=('hm', 'wh', 'reg'), training=True, use_dcn=False, use_skip=False):
    """-> dict head -> raw logits [B, n, H/4, W/4] in `heads` order; BN running statistics inside `state` are
    updated in place when training."""
    net = _Net(state, training)
    skip = {}
    x = net.cbr('base.0', x, 3, 2)
    for name, inp, oup, stride, t in block_plan():
        y, i = x, 0
        if t != 1:
            y = net.cbr('%s.conv.%d' % (name, i), y, 1)
            i += 1
        hidden = inp * t
        y = net.cbr('%s.conv.%d' % (name, i), y, 3, stride, hidden)
        y = net.bn('%s.conv.%d' % (name, i + 2), net.conv('%s.conv.%d' % (name, i + 1), y))
        x = x + y if (stride == 1 and inp == oup) else y
        lid = int(name.split('.')[1])
        if use_skip and lid in SKIP_MAPPING.values():
            skip[{v: k for k, v in SKIP_MAPPING.items()}[lid]] = x
    x = net.cbr('base.18', x, 1)
    lid = 0
    for _ in range(3):
        if use_dcn:
            x = net.dcn('deconv_layers.%d' % lid, x)
            if lid in skip:
                x = net.conv('skip_%d' % lid, skip[lid]) + x
            x = F.relu(net.bn('deconv_layers.%d' % (lid + 1), x))
            lid += 3
        x = F.conv_transpose2d(x, state['deconv_layers.%d.weight' % lid], None, 2, 1, 0)
        if lid in skip:
            x = net.conv('skip_%d' % lid, skip[lid]) + x
        x = F.relu(net.bn('deconv_layers.%d' % (lid + 1), x))
        lid += 3
    out = {}
    for h in heads:
        y = F.relu(net.conv(h + '.0', x, 1, 1))
        out[h] = net.conv(h + '.2', y)
    return out


# ---------------------------------------------------------------------------
# torch.nn restatement of the hub trunk (golden generation only)
# ---------------------------------------------------------------------------
class _ConvBNReLU(nn.Sequential):
    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, groups=1):
        padding = (kernel_size - 1) // 2
        super().__init__(nn.Conv2d(in_planes, out_planes, kernel_size, stride, padding, groups=groups, bias=False),
                         nn.BatchNorm2d(out_planes), nn.ReLU6(inplace=True))


class _InvertedResidual(nn.Module):
    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        hidden_dim = int(round(inp * expand_ratio))
        self.use_res_connect = stride == 1 and inp == oup
        layers = []
        if expand_ratio != 1:
            layers.append(_ConvBNReLU(inp, hidden_dim, kernel_size=1))
        layers.extend([_ConvBNReLU(hidden_dim, hidden_dim, stride=stride, groups=hidden_dim),
                       nn.Conv2d(hidden_dim, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)])
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        return x + self.conv(x) if self.use_res_connect else self.conv(x)


class _TVMobileNetV2(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        features = [_ConvBNReLU(3, 32, stride=2)]
        cin = 32
        for t, c, n, s in SETTINGS:
            for i in range(n):
                features.append(_InvertedResidual(cin, c, s if i == 0 else 1, t))
                cin = c
        features.append(_ConvBNReLU(cin, 1280, kernel_size=1))
        self.features = nn.Sequential(*features)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(1280, num_classes))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)


def torchvision_mobilenet_v2():
    return _TVMobileNetV2()
