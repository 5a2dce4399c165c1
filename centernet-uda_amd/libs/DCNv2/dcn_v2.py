"""Python face of the deformable convolution: `dcn_v2_conv`, `DCNv2`, `DCN`
with the constructor/forward signatures, parameter names (`weight`, `bias`,
`conv_offset_mask.{weight,bias}` -- checkpoint keys) and initialisation of the
reference (libs/DCNv2/dcn_v2.py:18-128), running on the MI355X kernels behind
`_ext`.  The offset/mask generating 3x3 convolution of `DCN` is this repo's own
implicit-GEMM convolution (hip_runtime.nn.Conv2d), not a vendor library.
"""
import math

import torch
from torch import nn

import _ext as _backend


# CNUDA_DCN_KEEP_COLS=0: the forward does not store the sampled columns and the weight gradient samples the input again
# (the reference's scheme, dcn_v2_cuda.cu:302-319) -- an A/B switch for the measurement in DESIGN.md section 12; the
# default keeps them (one 1 GB side output per 128 x 128 layer, read once by a plain-GEMM weight gradient)
import os as _os
_KEEP_COLUMNS = _os.environ.get('CNUDA_DCN_KEEP_COLS', '1') != '0'
# CNUDA_DCN_OM=0 (or dcn_v2.USE_OM = False): DCN.forward materialises offset and mask tensors like the reference
# (split + sigmoid kernel, its backward twin) instead of reading them out of the offset convolution's output
USE_OM = _os.environ.get('CNUDA_DCN_OM', '1') != '0'


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class _DeformConvFn(torch.autograd.Function):
    # autograd-visible argument order (dcn_v2.py:18-19): input, offset, mask, weight, bias, ...
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, pack_token=0,
                stats_box=None, regime=0):
        kh, kw = weight.shape[2], weight.shape[3]
        ctx.regime = int(regime)       # offset regime of this layer (DCN._census): which kernels the library picks
        from hip_runtime.fanout import slot_of
        ctx.slot = slot_of(input)      # where the offset convolution (the input's other consumer) meets this gradient
        ctx.geom = (kh, kw) + _pair(stride) + _pair(padding) + _pair(dilation) + (deformable_groups,)
        # keep the sampled columns (a side output of the forward kernel) for the weight gradient:
        # on a 288 GB part re-reading ~0.3 GB per layer beats re-sampling the input
        keep = input.shape[3] >= 2 and any(ctx.needs_input_grad[:5]) and _KEEP_COLUMNS
        with _offset_regime(ctx.regime):
            if keep:
                out, cols = _backend.dcn_v2_forward(input, weight, bias, offset, mask, *ctx.geom, _want_columns=True,
                                                    _pack_token=pack_token, _stats_box=stats_box)
            else:
                out, cols = _backend.dcn_v2_forward(input, weight, bias, offset, mask, *ctx.geom,
                                                    _pack_token=pack_token, _stats_box=stats_box), None
        ctx.save_for_backward(input, offset, mask, weight, bias, cols)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias, cols = ctx.saved_tensors
        from hip_runtime.arena import grad_sink
        sw, sb = grad_sink(weight), grad_sink(bias)       # arena slots: written by the kernels, not returned
        # the data-gradient walks ADD into grad_input: on top of what the slot already holds when it is the slot's own
        # buffer, else into a cleared tensor -- which an empty slot takes over
        slot = ctx.slot
        acc = slot.buf if (slot is not None and slot.buf is not None and slot.owned) else None
        with _offset_regime(ctx.regime):
            g_in, g_off, g_mask, g_w, g_b = _backend.dcn_v2_backward(
                input, weight, bias, offset, mask, grad_output, *ctx.geom, _columns=cols, _grad_weight=sw, _grad_bias=sb,
                _grad_input=acc)
        if acc is None:
            from hip_runtime.fanout import claim
            claim(slot, g_in)
        else:
            slot.included.append(acc)
        return g_in, g_off, g_mask, (None if sw is not None else g_w), (None if sb is not None else g_b), \
            None, None, None, None, None, None, None


class _DeformConvOmFn(torch.autograd.Function):
    """The same operation with offsets and mask read out of `om`, the 3T-channel output of the layer's own offset convolution
    whose mask channels already went through the sigmoid (ops.conv2d_rowsig): no split / concatenate / sigmoid tensors in the
    forward, and ONE gradient tensor for `om` in the backward -- the offsets' gradient and the gradient of the mask's LOGIT,
    which is what that convolution's backward consumes (round 6; `DCN.forward` only, deformable_groups == 1)."""

    @staticmethod
    def forward(ctx, input, om, weight, bias, stride, padding, dilation, pack_token=0, stats_box=None, regime=0):
        kh, kw = weight.shape[2], weight.shape[3]
        ctx.regime = int(regime)
        from hip_runtime.fanout import slot_of
        ctx.slot = slot_of(input)
        ctx.geom = (kh, kw) + _pair(stride) + _pair(padding) + _pair(dilation)
        keep = any(ctx.needs_input_grad[:4]) and _KEEP_COLUMNS
        with _offset_regime(ctx.regime):
            out, cols = _backend.dcn_v2_forward_om(input, weight, bias, om, *ctx.geom, _want_columns=keep,
                                                   _pack_token=pack_token, _stats_box=stats_box)
        ctx.save_for_backward(input, om, weight, bias, cols)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        input, om, weight, bias, cols = ctx.saved_tensors
        from hip_runtime.arena import grad_sink
        sw, sb = grad_sink(weight), grad_sink(bias)
        slot = ctx.slot
        acc = slot.buf if (slot is not None and slot.buf is not None and slot.owned) else None
        with _offset_regime(ctx.regime):
            g_in, g_om, g_w, g_b = _backend.dcn_v2_backward_om(input, weight, bias, om, grad_output, *ctx.geom, _columns=cols,
                                                               _grad_weight=sw, _grad_bias=sb, _grad_input=acc)
        if acc is None:
            from hip_runtime.fanout import claim
            claim(slot, g_in)
        else:
            slot.included.append(acc)
        return g_in, g_om, (None if sw is not None else g_w), (None if sb is not None else g_b), \
            None, None, None, None, None, None


class _offset_regime:
    """The library's offset regime (cnuda_dcn_set_offset_regime) for the calls inside the block; behind it the regime that
    was in force before (the setter returns it), so that nested or interleaved users do not reset each other."""

    def __init__(self, regime):
        self.regime = regime
        self.prev = 0

    def __enter__(self):
        import hip_runtime as hr
        self.prev = hr.lib().cnuda_dcn_set_offset_regime(self.regime)

    def __exit__(self, *exc):
        import hip_runtime as hr
        hr.lib().cnuda_dcn_set_offset_regime(self.prev)


def dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, pack_token=0,
                emit_stats=False, regime=0):
    """emit_stats (not part of the reference's signature): the output goes straight into a train-mode BatchNorm -- the
    kernel's epilogue then leaves the statistics with it (hip_runtime.ops.batch_norm_act finds them on the tensor)."""
    from hip_runtime import ops
    if not (emit_stats and ops.EPILOGUE_STATS):
        return _DeformConvFn.apply(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, pack_token,
                                   None, regime)
    box = []
    out = _DeformConvFn.apply(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, pack_token, box,
                              regime)
    if box:
        out._cnuda_bn_stats = box[0]
    return out


class DCNv2(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1,
                 deformable_groups=1):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        import hip_runtime as hr
        self._pack_token = hr.PackToken()           # identity of these weights for the library's pack cache
        self.reset_parameters()

    def reset_parameters(self):
        # U(-1/sqrt(fan_in), +1/sqrt(fan_in)) weights, zero bias (dcn_v2.py:75-81)
        bound = 1.0 / math.sqrt(self.in_channels * self.kernel_size[0] * self.kernel_size[1])
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            self.bias.zero_()

    def forward(self, input, offset, mask):
        taps = self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]
        assert offset.shape[1] == 2 * taps and mask.shape[1] == taps
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding,
                           self.dilation, self.deformable_groups, self._pack_token)


class DCN(DCNv2):
    """DCNv2 that predicts its own offsets and modulation mask from the input."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1,
                 deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        from hip_runtime import nn as hnn
        taps = self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = hnn.Conv2d(in_channels, 3 * taps, self.kernel_size, self.stride,
                                           self.padding, bias=True)
        self.emit_stats = False      # set by a caller whose next layer is a BatchNorm2d (backends.dla.DeformConv)
        # offset regime of this layer (cnuda_dcn_set_offset_regime): re-measured every CENSUS_EVERY training forwards by
        # a census of its own offsets -- one small launch and one host read, i.e. one synchronisation per layer and 64 steps
        self._regime, self._census_calls = 0, 0
        with torch.no_grad():        # zero init: offsets 0, mask sigmoid(0)=0.5 (dcn_v2.py:114-116, Q7)
            self.conv_offset_mask.weight.zero_()
            self.conv_offset_mask.bias.zero_()

    def forward(self, input):
        from hip_runtime import ops
        from hip_runtime.fanout import fork
        # the input feeds the offset / mask convolution AND the sampling: their gradients meet in a slot, not in the engine
        input_om, input = fork(input, 2)
        taps = self.kernel_size[0] * self.kernel_size[1]
        cm = self.conv_offset_mask
        # (a forward hook on `conv_offset_mask` -- bench.measure_dcn_offsets, a user's probe -- wants that module CALLED: then,
        # as with CNUDA_DCN_OM=0, the module runs and its output is split like the reference does)
        hooked = bool(cm._forward_hooks or cm._forward_pre_hooks or cm._backward_hooks)
        if self.deformable_groups == 1 and input.shape[3] >= 2 and USE_OM and not hooked:
            # round 6: the offset convolution's epilogue applies the mask's sigmoid and the deformable convolution reads
            # offsets and mask out of its one output tensor (no split kernels, no offset / mask tensors of their own)
            om = ops.conv2d_rowsig(input_om, cm.weight, cm.bias, cm.stride, cm.padding, 2 * taps, cm._pack_token)
            if om is not None:
                if self.training:
                    self._census_calls += 1
                    if self._census_calls % self.CENSUS_EVERY == 1:
                        self._regime = self._census(om.detach()[:, :2 * taps].contiguous())
                box = [] if (self.emit_stats and self.training and ops.EPILOGUE_STATS) else None
                out = _DeformConvOmFn.apply(input, om, self.weight, self.bias, self.stride, self.padding, self.dilation,
                                            self._pack_token, box, self._regime)
                if box:
                    out._cnuda_bn_stats = box[0]
                return out
        om = cm(input_om)
        # channels [0, 2*taps) are offsets (chunks o1|o2 re-concatenated, dcn_v2.py:120-121),
        # [2*taps, 3*taps) the mask logits
        offset, mask = ops.split_offset_mask(om)
        if self.training and self.deformable_groups == 1:
            self._census_calls += 1
            if self._census_calls % self.CENSUS_EVERY == 1:
                self._regime = self._census(offset)
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding,
                           self.dilation, self.deformable_groups, self._pack_token,
                           emit_stats=self.emit_stats and self.training, regime=self._regime)

    CENSUS_EVERY = 64
    # shares of (pixel, tap) samples beyond +-2 px / +-3 px above which the wide-window walk / the gathering forward win
    # (profiles/r5_dcn_margin_sweep.txt, r4_dcnw_large_offsets.txt: the crossovers sit near sigma = 0.75 px and 1.25 px)
    CENSUS_SHARES = (0.015, 0.03)

    def _census(self, offset):
        import hip_runtime as hr
        off = offset.detach()
        B, HW = off.shape[0], off.shape[2] * off.shape[3]
        taps = off.shape[1] // 2
        counts = torch.zeros(2, dtype=torch.int32, device=off.device)
        hr.check(hr.lib().cnuda_dcn_offset_census(hr.ptr(hr.f32c(off)), B, taps, HW, hr.ptr(counts), hr.stream()), 'census')
        n2, n3 = counts.tolist()
        total = float(B * taps * HW)
        return (1 if n2 > self.CENSUS_SHARES[0] * total else 0) | (2 if n3 > self.CENSUS_SHARES[1] * total else 0)
