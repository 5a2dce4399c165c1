"""DLA-34 + DCNv2 CenterNet backend for MI355X.

Plugin contract of the reference (backends/dla.py:513-531, train.py:85-86,119):
`build(num_classes, num_keypoints=0, head_conv=256, down_ratio=4,
freeze_base=False, rotated_boxes=False)` returns an nn.Module with
`.down_ratio`, `.rotated_boxes` and `forward(x[B,3,H,W]) -> {'hm','wh','reg'}`
raw logits at H/4, in that key order.  The module tree reproduces the
reference's parameter names (`base.level3.tree1.root.conv.weight`,
`dla_up.ida_1.proj_2.conv.conv_offset_mask.bias`, `hm.2.bias`, ...), which are
the checkpoint wire format (utils/helper.py:95-117).

Every layer executes on this repo's gfx950 kernels (hip_runtime.ops): implicit
GEMM fp32-MFMA convolutions, fused BatchNorm + residual + ReLU, the fused
deformable convolution of libs.DCNv2, depthwise transposed-conv upsampling.
Differences from the reference that do not change results: BN, residual add
and ReLU are one kernel; ReLU lives in the conv/BN epilogues (no separate
module instances); the `.clone()` of dla.py:504 is dropped because no op here
writes in place.

What is reproduced on purpose: the level-3/level-4 `project` branches are
evaluated although their result is discarded (dla.py:207-213), so their
BatchNorm running statistics advance exactly like the reference's (and their
weights receive no gradient).  `build` starts from the ImageNet trunk like the
reference (dla.py:297-309,524-526) when the file the reference would download
(`dla34-ba72cf86.pth`) is found locally -- `$CNUDA_DLA34_WEIGHTS`, else the torch
hub checkpoint cache -- and then also carries the unused `base.fc` (Q8); there
is no network path in this build, so without the file it WARNS and keeps the
default initialisation.
"""
import logging
import math
import os
import warnings

import torch
from torch import nn

from hip_runtime import nn as hnn
from hip_runtime import ops
from hip_runtime.fanout import fork
from libs.DCNv2.dcn_v2 import DCN

log = logging.getLogger(__name__)
PRETRAINED_FILE = 'dla34-ba72cf86.pth'          # get_model_url('imagenet', 'dla34', 'ba72cf86'), dla.py:22-25

BN_MOMENTUM = 0.1
DLA34_LEVELS = (1, 1, 1, 2, 2, 1)
DLA34_CHANNELS = (16, 32, 64, 128, 256, 512)


def _bn(c):
    return hnn.BatchNorm2d(c, momentum=BN_MOMENTUM)


# ---------------------------------------------------------------------------
# BatchNorm folding for inference (export.py: SURVEY 8f row 2).  `fold_batchnorm(model)` gives every conv + BN block
# a folded weight / bias pair (w * gamma / sqrt(var + eps), beta + (conv_bias - mean) * gamma / sqrt(var + eps));
# the blocks then run as ONE kernel each -- convolution, bias, skip connection and ReLU in the GEMM epilogue -- when
# the module is in eval mode and no tape is being recorded.  The state_dict is untouched (folded tensors are plain
# attributes); `unfold_batchnorm` drops them (call it, or fold again, after the weights change).
# ---------------------------------------------------------------------------
_FOLD_GEN = [0]


def _fold_stamp(module):
    """(token, version) under which the library may cache the packed image of a module's folded weights: a token of
    the folded copy's own, and the generation of the fold (a re-fold may land on a recycled address)."""
    import hip_runtime as hr
    if not hasattr(module, '_fold_token'):
        module._fold_token = hr.PackToken()
    _FOLD_GEN[0] += 1
    module._fold_gen = _FOLD_GEN[0]


def _folded(conv_weight, conv_bias, bn):
    with torch.no_grad():
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = (conv_weight * scale.view(-1, 1, 1, 1)).contiguous()
        b = bn.bias - bn.running_mean * scale
        if conv_bias is not None:
            b = b + conv_bias * scale
        return w, b.contiguous()


_FOLDED_INFERENCE = False


class folded_inference:
    """with folded_inference(): ...  -- the only place the folded weights are used (export.CenterNet.forward enters
    it after checking that they are current); everywhere else the blocks run their BatchNorm kernels."""

    def __enter__(self):
        global _FOLDED_INFERENCE
        self.prev, _FOLDED_INFERENCE = _FOLDED_INFERENCE, True

    def __exit__(self, *exc):
        global _FOLDED_INFERENCE
        _FOLDED_INFERENCE = self.prev


def _use_folded(module):
    return (_FOLDED_INFERENCE and getattr(module, '_fold', None) is not None and not module.training
            and not torch.is_grad_enabled())


def fold_batchnorm(model):
    n = 0
    for m in model.modules():
        if hasattr(m, 'fold_batchnorm_'):
            m.fold_batchnorm_()
            n += 1
    return n


def unfold_batchnorm(model):
    for m in model.modules():
        if hasattr(m, 'fold_batchnorm_'):
            m._fold = None


def _conv(cin, cout, k, stride=1):
    return hnn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False, emit_stats=True)    # (every caller: conv -> BatchNorm)


def _norm_on_load_ok(cin, cout, k):
    """Does a k x k / stride 1 convolution cin -> cout of this library normalise its input on load?"""
    try:
        import hip_runtime as hr
        return bool(hr.lib().cnuda_conv2d_norm_input_supported(1, cin, 64, 64, cout, k, k, 1, 1, k // 2, k // 2))
    except Exception:       # (no library in this process: the structure tests on a CPU-only box build the model anyway)
        return False


class ConvBnRelu(nn.Sequential):
    """children '0' (conv) and '1' (bn); ReLU is fused into the BN kernel."""

    _fold = None

    def __init__(self, cin, cout, k, stride=1):
        super().__init__(_conv(cin, cout, k, stride), _bn(cout))

    def fold_batchnorm_(self):
        self._fold = _folded(self[0].weight, self[0].bias, self[1])
        _fold_stamp(self)

    def forward(self, x):
        if _use_folded(self):
            return ops.conv2d_infer(x, *self._fold, self[0].stride, self[0].padding, 0.0, None, self._fold_token,
                                    self._fold_gen)
        return self[1](self[0](x), relu=True)


class ConvBn(nn.Sequential):
    _fold = None

    def __init__(self, cin, cout):
        super().__init__(_conv(cin, cout, 1), _bn(cout))

    def fold_batchnorm_(self):
        self._fold = _folded(self[0].weight, self[0].bias, self[1])
        _fold_stamp(self)

    def forward(self, x):
        if _use_folded(self):
            return ops.conv2d_infer(x, *self._fold, self[0].stride, self[0].padding, -1.0, None, self._fold_token,
                                    self._fold_gen)
        return self[1](self[0](x))


class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1, self.bn1 = _conv(cin, cout, 3, stride), _bn(cout)
        self.conv2, self.bn2 = _conv(cout, cout, 3), _bn(cout)

    _fold = None

    def fold_batchnorm_(self):
        import hip_runtime as hr
        self._fold = _folded(self.conv1.weight, None, self.bn1) + _folded(self.conv2.weight, None, self.bn2)
        _fold_stamp(self)
        if not hasattr(self, '_fold_token2'):          # one identity per folded weight: two of one shape under one
            self._fold_token2 = hr.PackToken()         # token would keep taking over each other's packed image

    def forward(self, x, residual=None):
        if _use_folded(self):
            w1, b1, w2, b2 = self._fold
            y = ops.conv2d_infer(x, w1, b1, self.conv1.stride, self.conv1.padding, 0.0, None, self._fold_token,
                                 self._fold_gen)
            # conv2 + BatchNorm + skip connection + ReLU in one launch (dla.py:48-62)
            return ops.conv2d_infer(y, w2, b2, 1, 1, 0.0, x if residual is None else residual, self._fold_token2,
                                    self._fold_gen)
        if residual is None or residual is x:
            # the block's input also is its skip connection: two aliases, one gradient slot (hip_runtime.fanout) -- bn2's
            # backward leaves the skip's share there, conv1's input-gradient GEMM adds it in its epilogue
            x, residual = fork(x, 2)
        y = self.bn1(self.conv1(x), relu=True)
        return self.bn2(self.conv2(y), residual=residual, relu=True)


CAT_FREE_ROOT = os.environ.get('CNUDA_CAT_FREE_ROOT', '1') != '0'     # (A/B measurements, tests/test_gpu_kernel_switches.py)


class Root(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv, self.bn = _conv(cin, cout, 1), _bn(cout)

    _fold = None

    def fold_batchnorm_(self):
        self._fold = _folded(self.conv.weight, None, self.bn)
        _fold_stamp(self)

    def forward(self, *xs):
        if _use_folded(self):
            y = ops.conv1x1_cat_infer(xs, *self._fold, 0.0, self._fold_token, self._fold_gen) if CAT_FREE_ROOT else None
            return y if y is not None else ops.conv2d_infer(ops.cat_channels(xs), *self._fold, 1, 0, 0.0, None,
                                                            self._fold_token, self._fold_gen)
        # the convolution over the concatenation without the concatenation (round 6: ops.conv1x1_cat; 17 copies per forward
        # pass of DLA-34 and their 17 slices in the backward pass); None where no kernel takes these sources
        y = ops.conv1x1_cat(xs, self.conv.weight, self.conv._pack_token, emit_stats=self.conv.emit_stats and self.training) \
            if CAT_FREE_ROOT and self.conv.kernel_size == (1, 1) and self.conv.bias is None else None
        if y is None:
            y = self.conv(ops.cat_channels(xs))
        return self.bn(y, relu=True)


class Tree(nn.Module):
    """Hierarchical aggregation node (reference dla.py:171-224), BasicBlock leaves."""

    def __init__(self, levels, cin, cout, stride=1, level_root=False, root_dim=0):
        super().__init__()
        self.levels, self.level_root = levels, level_root
        root_dim = root_dim or 2 * cout
        if level_root:
            root_dim += cin
        if levels == 1:
            self.tree1 = BasicBlock(cin, cout, stride)
            self.tree2 = BasicBlock(cout, cout, 1)
            self.root = Root(root_dim, cout)
        else:
            self.tree1 = Tree(levels - 1, cin, cout, stride)
            self.tree2 = Tree(levels - 1, cout, cout, root_dim=root_dim + cout)
        self.downsample = hnn.MaxPool2d(stride) if stride > 1 else None
        self.project = ConvBn(cin, cout) if cin != cout else None

    def forward(self, x, children=None):
        children = [] if children is None else children
        # Every tensor with several consumers is handed out as one alias per consumer (hip_runtime.fanout.fork): the
        # consumers' gradients then meet in a slot that a convolution's input-gradient epilogue sums, not in passes of
        # the autograd engine.  A consumer whose output never reaches a loss simply leaves its alias without a gradient.
        if self.downsample is not None:
            x_pool, x = fork(x, 2)
            bottom = self.downsample(x_pool)
            takers = int(self.project is not None or self.levels == 1) + int(self.level_root)
            bs = list(fork(bottom, takers)) if takers > 1 else [bottom]
            residual = self.project(bs.pop()) if self.project is not None else (bs.pop() if self.levels == 1 else None)
            if self.level_root:
                children.append(bs.pop())
        else:                       # bottom is x itself
            takers = 1 + int(self.project is not None) + int(self.level_root)
            xs = list(fork(x, takers)) if takers > 1 else [x]
            x = xs.pop()
            residual = self.project(xs.pop()) if self.project is not None else x   # identity: forked inside the block
            if self.level_root:
                children.append(xs.pop())
        if self.levels == 1:
            x1 = self.tree1(x, residual)
            x1_next, x1_root = fork(x1, 2)
            x2 = self.tree2(x1_next)
            return self.root(x2, x1_root, *children)
        # an inner Tree recomputes its own residual; `residual` above is only
        # evaluated for its BatchNorm side effect (see module docstring)
        x1 = self.tree1(x)
        x1_next, x1_root = fork(x1, 2)
        children.append(x1_root)
        return self.tree2(x1_next, children=children)


class DLA(nn.Module):
    def __init__(self, levels=DLA34_LEVELS, channels=DLA34_CHANNELS):
        super().__init__()
        c = self.channels = list(channels)
        self.base_layer = ConvBnRelu(3, c[0], 7)
        self.level0 = ConvBnRelu(c[0], c[0], 3)
        # the stem's BatchNorm + ReLU is applied by level0's convolution while it stages its input (hip_runtime.ops.conv2d,
        # apply on load: the 16-channel full-resolution activation is neither written nor read back) -- level0 is its only
        # consumer (forward below)
        self.base_layer[1].defer_apply = _norm_on_load_ok(c[0], c[0], 3)
        self.level1 = ConvBnRelu(c[0], c[1], 3, stride=2)
        self.level2 = Tree(levels[2], c[1], c[2], 2, level_root=False)
        self.level3 = Tree(levels[3], c[2], c[3], 2, level_root=True)
        self.level4 = Tree(levels[4], c[3], c[4], 2, level_root=True)
        self.level5 = Tree(levels[5], c[4], c[5], 2, level_root=True)

    def load_pretrained_model(self, path):
        """dla.py:297-309: attaches the classifier `fc` (1x1 conv, never used by forward: Q8) sized from the
        checkpoint's last entry, then loads strictly."""
        weights = torch.load(path, map_location='cpu')
        num_classes = len(weights[list(weights.keys())[-1]])
        self.fc = hnn.Conv2d(self.channels[-1], num_classes, 1, bias=True)
        self.load_state_dict(weights)

    def forward(self, x):
        x = self.base_layer(x)
        feats = []
        for name in ('level0', 'level1', 'level2', 'level3', 'level4'):
            x, keep = fork(getattr(self, name)(x), 2)       # the next level's input | the aggregation's
            feats.append(keep)
        feats.append(self.level5(x))
        return feats


def pretrained_path():
    """Where the ImageNet dla34 state dict is looked for: `$CNUDA_DLA34_WEIGHTS`, then the file name
    model_zoo.load_url() would have cached (dla.py:303-304)."""
    env = os.environ.get('CNUDA_DLA34_WEIGHTS')
    return env if env else os.path.join(torch.hub.get_dir(), 'checkpoints', PRETRAINED_FILE)


def dla34(pretrained=False):
    """pretrained: True -> the file must exist (raises, like a failed download); 'auto' -> load it when it is
    there, warn loudly when it is not; False -> default initialisation."""
    model = DLA()
    if pretrained:
        path = pretrained_path()
        if os.path.isfile(path):
            model.load_pretrained_model(path)
        elif pretrained == 'auto':
            msg = ("dla34: ImageNet weights %s not found -- the trunk is RANDOMLY initialised, whereas the reference's "
                   "build() always starts from them (backends/dla.py:524-526; no download in this build). Put the "
                   "file there or set CNUDA_DLA34_WEIGHTS, or load a checkpoint with utils.helper.load_model." % path)
            log.warning(msg)
            warnings.warn(msg, RuntimeWarning, stacklevel=3)
        else:
            raise RuntimeError("dla34(pretrained=True): %s not found (http://dl.yf.io/dla/models, dla.py:23-25; no "
                               "network in this build) -- set CNUDA_DLA34_WEIGHTS or load a checkpoint with "
                               "utils.helper.load_model" % path)
    return model


class DeformConv(nn.Module):
    """DCN 3x3 -> BatchNorm -> ReLU (dla.py:351-372)."""

    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(_bn(cho))
        self.conv = DCN(chi, cho, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
        self.conv.emit_stats = True       # (its only consumer is the BatchNorm above: statistics from the DCN's epilogue)

    _fold = None

    def fold_batchnorm_(self):
        self._fold = _folded(self.conv.weight, self.conv.bias, self.actf[0])
        _fold_stamp(self)

    def forward(self, x):
        if _use_folded(self):
            # offsets / mask from the unchanged 27-channel convolution, then DCN + BatchNorm + ReLU as one launch
            import _ext
            c = self.conv
            from libs.DCNv2 import dcn_v2 as _dcn
            cm = c.conv_offset_mask
            taps = c.kernel_size[0] * c.kernel_size[1]
            # (round 6) offsets and mask straight out of the offset convolution's output, its mask rows sigmoid in the epilogue
            om = ops.conv2d_rowsig(x, cm.weight, cm.bias, cm.stride, cm.padding, 2 * taps, cm._pack_token) \
                if (_dcn.USE_OM and c.deformable_groups == 1 and x.shape[3] >= 2 and
                    not (cm._forward_hooks or cm._forward_pre_hooks)) else None
            if om is not None:
                return _ext.dcn_v2_forward_om(x, self._fold[0], self._fold[1], om, *c.kernel_size, *c.stride, *c.padding,
                                              *c.dilation, _act_slope=0.0, _pack_token=self._fold_token,
                                              _pack_version=self._fold_gen)[0]
            offset, mask = ops.split_offset_mask(cm(x))
            return _ext.dcn_v2_forward(x, self._fold[0], self._fold[1], offset, mask, *c.kernel_size, *c.stride,
                                       *c.padding, *c.dilation, c.deformable_groups, _act_slope=0.0,
                                       _pack_token=self._fold_token, _pack_version=self._fold_gen)
        return self.actf[0](self.conv(x), relu=True)


def _bilinear_upsample_init(up):
    """Separable triangle filter for a k = 2f transposed conv (dla.py:339-348)."""
    k = up.kernel_size
    f = math.ceil(k / 2)
    centre = (2 * f - 1 - f % 2) / (2.0 * f)
    tri = torch.tensor([1 - abs(i / f - centre) for i in range(k)], dtype=torch.float32)
    with torch.no_grad():
        up.weight.copy_((tri[:, None] * tri[None, :]).expand_as(up.weight))


class IDAUp(nn.Module):
    def __init__(self, o, channels, up_f):
        super().__init__()
        self.n = len(channels)
        for i in range(1, self.n):
            f = int(up_f[i])
            up = hnn.DepthwiseConvTranspose2d(o, f * 2, stride=f, padding=f // 2)
            _bilinear_upsample_init(up)
            setattr(self, 'proj_%d' % i, DeformConv(channels[i], o))
            setattr(self, 'up_%d' % i, up)
            setattr(self, 'node_%d' % i, DeformConv(o, o))

    def forward(self, layers, startp, endp, fan=None):
        """fan: the aggregation's alias plan (_FanPlan) -- `layers` then holds _Aliases and every read takes its own alias."""
        for i in range(startp + 1, endp):
            k = i - startp
            src, skip = (layers[i], layers[i - 1]) if fan is None else (layers[i].take(), layers[i - 1].take())
            t = getattr(self, 'up_%d' % k)(getattr(self, 'proj_%d' % k)(src), skip)   # up(..) + skip
            t = getattr(self, 'node_%d' % k)(t)
            layers[i] = t if fan is None else fan.wrap(t)

    @staticmethod
    def plan(ids, startp, endp, reads, fresh):
        """the reads of forward() on tensor ids"""
        for i in range(startp + 1, endp):
            reads[ids[i]] += 1
            reads[ids[i - 1]] += 1
            ids[i] = fresh()


class DLAUp(nn.Module):
    def __init__(self, startp, channels, scales):
        super().__init__()
        self.startp = startp
        channels, in_channels, scales = list(channels), list(channels), list(scales)
        n = len(channels)
        for i in range(n - 1):
            j = n - i - 2
            setattr(self, 'ida_%d' % i, IDAUp(channels[j], in_channels[j:], [s // scales[j] for s in scales[j:]]))
            for q in range(j + 1, n):
                scales[q] = scales[j]
                in_channels[q] = channels[j]

    def forward(self, layers, fan=None):
        layers = list(layers) if fan is None else [fan.wrap(t) for t in layers]
        out = [layers[-1]]
        for i in range(len(layers) - self.startp - 1):
            getattr(self, 'ida_%d' % i)(layers, len(layers) - i - 2, len(layers), fan)
            out.insert(0, layers[-1])
        return out

    def plan(self, ids, reads, fresh):
        out = [ids[-1]]
        for i in range(len(ids) - self.startp - 1):
            IDAUp.plan(ids, len(ids) - i - 2, len(ids), reads, fresh)
            out.insert(0, ids[-1])
        return out


class _Aliases:
    """A tensor of the aggregation with one alias per reader (hip_runtime.fanout.fork)."""

    def __init__(self, t, n):
        self.t, self.pool = t, (list(fork(t, n)) if n > 1 else None)

    def take(self):
        if self.pool is None:                    # planned for one reader: the tensor itself
            return self.t
        if not self.pool:
            raise RuntimeError("DLAUp / IDAUp read a tensor more often than _FanPlan counted: plan() and forward() drifted apart")
        return self.pool.pop()


class _FanPlan:
    """How often every tensor of DLAUp + IDAUp is read (layers are re-used as skip connections and as inputs of later
    aggregation steps): found once by walking the same loops over tensor ids, then every tensor is forked into that many
    aliases when it is produced."""

    def __init__(self, seg, n_feats):
        import collections
        reads = collections.Counter()
        counter = [n_feats]

        def fresh():
            counter[0] += 1
            return counter[0] - 1
        out = seg.dla_up.plan(list(range(n_feats)), reads, fresh)
        y = out[:seg.last_level - seg.first_level]
        IDAUp.plan(y, 0, len(y), reads, fresh)
        reads[y[-1]] += 1                      # the map the heads read
        self.reads, self.n, self.n_ids = reads, 0, counter[0]

    def start(self):
        self.n = 0
        return self

    def wrap(self, t):
        if self.n >= self.n_ids:
            raise RuntimeError("DLAUp / IDAUp produced more tensors than _FanPlan walked: plan() and forward() drifted apart")
        self.n += 1
        return _Aliases(t, self.reads[self.n - 1])


class _TapeFreeTargetHead(torch.autograd.Function):
    """Marks a target-domain head output that `forward_domains` evaluated without a tape: the value is the head's, the
    node records nothing and costs nothing -- unless a backward pass reaches it, which means a loss was attached to a
    head that `target_grad_heads` does not list."""

    @staticmethod
    def forward(ctx, y, anchor, head):
        ctx.head = head
        return y.view_as(y)

    @staticmethod
    def backward(ctx, grad):
        raise RuntimeError(
            "forward_domains evaluated the target-domain head %r without a tape (it is not in target_grad_heads), but a "
            "loss back-propagates into it.  List it -- e.g. plugin.target_grad_heads = ('hm', %r) -- or set "
            "plugin.batch_domains = False for the reference's literal two-pass sequence." % (ctx.head, ctx.head))


class DLASeg(nn.Module):
    def __init__(self, base_name, heads, pretrained, down_ratio, final_kernel, last_level, head_conv,
                 out_channel=0, freeze_base=False, rotated_boxes=False):
        super().__init__()
        assert down_ratio in (2, 4, 8, 16)
        if base_name != 'dla34':
            raise ValueError("only 'dla34' is built (dla.py:521 hard-codes it)")
        self.down_ratio, self.rotated_boxes = down_ratio, rotated_boxes
        self._fan_plan = None       # (n feature maps, _FanPlan): the aggregation's reader counts, walked once
        self.first_level, self.last_level = int(math.log2(down_ratio)), last_level
        self.base = dla34(pretrained=pretrained)
        if freeze_base:
            for p in self.base.parameters():
                p.requires_grad = False
        channels = self.base.channels
        up_channels = channels[self.first_level:]
        self.dla_up = DLAUp(self.first_level, up_channels, [2 ** i for i in range(len(up_channels))])
        out_channel = out_channel or channels[self.first_level]
        self.ida_up = IDAUp(out_channel, channels[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)])
        self.heads = dict(heads)
        for head, classes in self.heads.items():
            if head_conv > 0:
                fc = hnn.Head(
                    hnn.Conv2d(channels[self.first_level], head_conv, 3, padding=1, bias=True, act_slope=0.0),
                    hnn.Slot(),     # index of the reference's nn.ReLU (fused into conv '0')
                    hnn.Conv2d(head_conv, classes, final_kernel, padding=final_kernel // 2, bias=True))
                last = fc[2]
            else:
                fc = last = hnn.Conv2d(channels[self.first_level], classes, final_kernel,
                                       padding=final_kernel // 2, bias=True)
            with torch.no_grad():
                if 'hm' in head:
                    last.bias.fill_(-2.19)                       # dla.py:485
                else:
                    for m in fc.modules():                       # fill_fc_weights, dla.py:332-336
                        if isinstance(m, hnn.Conv2d) and m.bias is not None:
                            m.bias.zero_()
            setattr(self, head, fc)

    def features(self, x):
        """The [B, 64, H/4, W/4] map every head reads (dla.py:500-505)."""
        feats = self.base(x)
        if not (torch.is_grad_enabled() and feats[-1].requires_grad):
            feats = self.dla_up(feats)
            y = list(feats[:self.last_level - self.first_level])
            self.ida_up(y, 0, len(y))
            return y[-1]
        # recording a tape: every tensor of the aggregation is handed out as one alias per reader (hip_runtime.fanout)
        if self._fan_plan is None or self._fan_plan[0] != len(feats):     # built once: the module tree is fixed after __init__
            self._fan_plan = (len(feats), _FanPlan(self, len(feats)))
        fan = self._fan_plan[1].start()
        feats = self.dla_up(feats, fan)
        y = list(feats[:self.last_level - self.first_level])
        self.ida_up(y, 0, len(y), fan)
        return y[-1].take()

    def forward(self, x):
        feat = self.features(x)
        return {head: getattr(self, head)(f) for head, f in zip(self.heads, fork(feat, len(self.heads)))}

    def forward_domains(self, source, target, target_grad_heads=('hm',)):
        """Both domains' forward passes of a UDA step (uda/entropy_minimization.py:18-19) as ONE pass over the
        concatenated batch: every layer runs once on twice the pixels (half the launches, fuller GEMM tiles on
        the small feature maps), each BatchNorm normalises the two halves by their own batch statistics and updates
        its running statistics once per half, source first (hip_runtime.domain_groups; Q6).  Returns
        (source outputs, target outputs), both with every head like two forward() calls.

        target_grad_heads: the heads whose TARGET output feeds a loss.  The other target heads are evaluated
        without a tape -- in the reference their outputs exist but no backward pass ever reaches them -- so that
        the single backward pass of the batched graph does not pay for 16 images of zero gradient there."""
        import hip_runtime as hr
        if source.shape != target.shape:
            raise RuntimeError("forward_domains: source %s and target %s batches differ in shape"
                               % (tuple(source.shape), tuple(target.shape)))
        B = source.shape[0]
        with hr.domain_groups(2 if self.training else 1):
            feat = self.features(torch.cat([source, target], 0))
        out_s, out_t = {}, {}
        f_t = feat[B:]
        # one alias of the feature map per head (hip_runtime.fanout).  The heads that only see the source half come first:
        # the backward pass then runs the whole-batch heads first, their gradient takes the map's slot with a plain write
        # and the source-half heads add theirs on top of its leading images
        order = sorted(self.heads, key=lambda h: h in target_grad_heads)
        for head, f in zip(order, fork(feat, len(order))):
            fc = getattr(self, head)
            if head in target_grad_heads or not torch.is_grad_enabled():
                y = fc(f)
                out_s[head], out_t[head] = y[:B], y[B:]
            else:
                out_s[head] = fc(f, lead=B) if isinstance(fc, hnn.Head) else fc(f[:B])
                with torch.no_grad():
                    y_t = fc(f_t)
                # requires_grad like the reference's (uda/entropy_minimization.py:18-19) -- and a loss that does reach
                # it fails loudly in backward() instead of silently training nothing
                out_t[head] = _TapeFreeTargetHead.apply(y_t, f_t, head) if f_t.requires_grad else y_t
        # (the dictionaries keep the heads' own order, like two forward() calls)
        return {h: out_s[h] for h in self.heads}, {h: out_t[h] for h in self.heads}


def build(num_classes, num_keypoints=0, head_conv=256, down_ratio=4, freeze_base=False, rotated_boxes=False):
    heads = {'hm': num_classes, 'wh': 3 if rotated_boxes else 2, 'reg': 2}
    if num_keypoints > 0:
        heads['kps'] = num_keypoints * 2
    return DLASeg('dla34', heads, pretrained='auto', down_ratio=down_ratio, final_kernel=1, last_level=5,
                  head_conv=head_conv, freeze_base=freeze_base, rotated_boxes=rotated_boxes)
