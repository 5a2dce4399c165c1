"""torch.autograd plumbing around the C ABI.  Every op here launches hand-written
gfx950 kernels from libcenternet_uda_hip.so on torch's current HIP stream;
torch contributes tensors (device memory), the autograd tape and nothing else.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import check, f32c, lib, pack_stamp, prof_arm, ptr, require_gpu, stream, workspace
from .arena import grad_sink
from .fanout import accumulate_target, claim, first_writer, slot_of


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _ws(nbytes, like):
    w = workspace(nbytes, like.device)
    return ptr(w), w.numel()


def _param_grad(param, needed=True):
    """Where a parameter's gradient goes: -> (buffer the kernel writes, value handed back to autograd).
    Arena-managed parameters get their slot of the arena's staging buffer and autograd gets None (see
    arena.py, "Gradient sink"); anything else gets a fresh tensor that is returned as usual."""
    if not needed or param is None:
        return None, None
    v = grad_sink(param)
    if v is not None:
        return v, None
    t = torch.empty_like(param)
    return t, t


# ---------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------
def _conv_geom(x, weight, stride, padding):
    B, C, H, W = x.shape
    Co, Ck, kh, kw = weight.shape
    if Ck != C:
        raise RuntimeError("conv2d: input has %d channels, weight expects %d" % (C, Ck))
    (sh, sw), (ph, pw) = _pair(stride), _pair(padding)
    return (B, C, H, W, Co, kh, kw, sh, sw, ph, pw)


class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, act_slope, pack_token=0, stats_box=None, norm=None, sig_from=None):
        """sig_from (private, DCN's offset convolution): output channels >= sig_from get a sigmoid in the epilogue
        (cnuda_conv2d_forward_rowsig) and the gradient this node RECEIVES for them is taken as the gradient of their LOGITS
        -- the deformable convolution's backward multiplies by m (1 - m) where it stores (cnuda_dcn_v2_backward_om) -- so
        the backward below is the plain convolution's.
        norm: None, or (x_raw, mean, invstd, gamma, beta, imgs_per_group) when `x` is the never-written output of a
        BatchNorm + ReLU in deferred mode (batch_norm_act(defer_apply=True)): the kernel reads x_raw and normalises while it
        stages it (cnuda_conv2d_forward_norm_input); the gradient this node returns for `x` is the one with respect to the
        normalised activation, exactly what the BatchNorm's backward expects."""
        require_gpu(x, weight, bias)
        ctx.slot = slot_of(x)          # (hip_runtime.fanout: where the other consumers of x leave their share of its gradient)
        # (norm: `x` is the 4-byte placeholder of a deferred BatchNorm -- shape only; the kernel reads norm[0])
        x, weight = (x if norm is not None else f32c(x)), f32c(weight)
        bias = None if bias is None else f32c(bias)
        g = _conv_geom(x, weight, stride, padding)
        B, C, H, W, Co, kh, kw, sh, sw, ph, pw = g
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        y = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=x.device)
        L = lib()
        wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g), x)
        stats = None
        if stats_box is not None and act_slope < 0:
            # BatchNorm statistics of y from the GEMM's epilogue, where this geometry's kernel can give them
            import ctypes
            rows, bpi = ctypes.c_int(0), ctypes.c_int(0)
            blk = L.cnuda_conv2d_stats_block(*g, ctypes.byref(rows), ctypes.byref(bpi))
            if blk:
                # (stats, pixels per block on the flattened (image, pixel) axis -- or 0 --, rows, blocks per image -- or 0)
                nblk = B * bpi.value if bpi.value else (B * Ho * Wo + 127) // 128 * (128 // blk)
                stats = torch.empty((nblk, rows.value, 2), dtype=torch.float32, device=x.device)
                stats_box.append((stats, 0 if bpi.value else blk, rows.value, bpi.value))
        prof_arm('conv_fwd', B, C, H, W, Co, kh, kw, Ho, Wo)
        with pack_stamp(pack_token, weight):
            if norm is not None:
                xr, mean, invstd, gamma, beta, ipg = norm
                check(L.cnuda_conv2d_forward_norm_input(ptr(xr), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), int(ipg),
                                                        ptr(weight), ptr(bias), ptr(y), ptr(stats), *g, float(act_slope),
                                                        wp, wn, stream()), 'conv2d_forward_norm_input')
            elif sig_from is not None:
                check(L.cnuda_conv2d_forward_rowsig(ptr(x), ptr(weight), ptr(bias), ptr(y), int(sig_from), *g, wp, wn, stream()),
                      'conv2d_forward_rowsig')
            elif stats is None:
                check(L.cnuda_conv2d_forward(ptr(x), ptr(weight), ptr(bias), ptr(y), *g, float(act_slope),
                                             wp, wn, stream()), 'conv2d_forward')
            else:
                check(L.cnuda_conv2d_forward_stats(ptr(x), ptr(weight), ptr(bias), ptr(None), ptr(y), ptr(stats), *g,
                                                   float(act_slope), wp, wn, stream()), 'conv2d_forward_stats')
        ctx.geom, ctx.act_slope, ctx.has_bias, ctx.pack_token = g, act_slope, bias is not None, pack_token
        ctx.norm_ipg = None
        if norm is not None:
            # (x itself holds nothing: the weight gradient normalises x_raw again while it stages it)
            ctx.norm_ipg = int(norm[5])
            ctx.save_for_backward(norm[0], weight, y if act_slope >= 0 else None, bias, norm[1], norm[2], norm[3], norm[4])
        else:
            ctx.save_for_backward(x, weight, y if act_slope >= 0 else None, bias)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight, y, bias = ctx.saved_tensors[:4]
        g = ctx.geom
        L = lib()
        gy = f32c(gy)
        if ctx.act_slope >= 0:
            t = torch.empty_like(gy)
            check(L.cnuda_act_backward(ptr(gy), ptr(y), ptr(t), gy.numel(), float(ctx.act_slope), stream()))
            gy = t
        wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g), x)
        gx = gw = gb = None
        B, C, H, W, Co, kh, kw = g[:7]
        Ho, Wo = gy.shape[2], gy.shape[3]
        if ctx.needs_input_grad[0]:
            gx, addend, addend2 = accumulate_target(ctx.slot, x)     # the slots' content is summed in the GEMM's epilogue
            prof_arm('conv_dgrad', B, C, H, W, Co, kh, kw, Ho, Wo)
            with pack_stamp(ctx.pack_token, weight):
                check(L.cnuda_conv2d_backward_data_add(ptr(gy), ptr(weight), ptr(addend), ptr(addend2), ptr(gx), *g, wp, wn,
                                                       stream()), 'conv2d_backward_data')
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw_buf, gw = _param_grad(weight)
            gb_buf, gb = _param_grad(bias, ctx.has_bias)
            prof_arm('conv_wgrad', B, C, H, W, Co, kh, kw, Ho, Wo)
            if ctx.norm_ipg is not None:
                mean, invstd, gamma, beta = ctx.saved_tensors[4:]
                check(L.cnuda_conv2d_backward_weight_norm_input(ptr(x), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
                                                                ctx.norm_ipg, ptr(gy), ptr(gw_buf), ptr(gb_buf), *g, wp, wn,
                                                                stream()), 'conv2d_backward_weight_norm_input')
            else:
                check(L.cnuda_conv2d_backward_weight(ptr(x), ptr(gy), ptr(gw_buf), ptr(gb_buf), *g, wp, wn, stream()),
                      'conv2d_backward_weight')
        return gx, gw, gb, None, None, None, None, None, None, None


def conv2d_rowsig(x, weight, bias, stride, padding, sig_from, pack_token=0):
    """(private to libs.DCNv2.dcn_v2.DCN) y = conv2d(x, weight) + bias with a sigmoid on the channels >= sig_from; the
    gradient handed back to this node must already be that of those channels' logits.  None where no kernel has the
    epilogue (cnuda_conv2d_rowsig_supported): the caller then takes the split path."""
    g = _conv_geom(x, weight, stride, padding)
    if bias is None or not lib().cnuda_conv2d_rowsig_supported(*g):
        return None
    return _Conv2d.apply(x, weight, bias, stride, padding, -1.0, pack_token, None, None, int(sig_from))


EPILOGUE_STATS = True      # (A/B measurements flip it: profiles/microbench/ab_bn_stats.py)


def conv2d(x, weight, bias=None, stride=1, padding=0, act_slope=-1.0, pack_token=0, emit_stats=False):
    """y = act(conv2d(x, weight) + bias); act_slope < 0 none, 0 ReLU, 0.2 LeakyReLU(0.2).  pack_token: identity of
    the module that owns `weight` (hip_runtime.new_pack_token) -- lets the library keep the packed weight image
    until the weights change; 0 = re-pack on every call."""
    norm = getattr(x, '_cnuda_deferred_bn', None)        # (batch_norm_act(defer_apply=True): x was never written)
    if norm is not None:
        g = _conv_geom(x, weight, stride, padding)
        if not lib().cnuda_conv2d_norm_input_supported(*g):
            raise RuntimeError("conv2d: the input is a deferred BatchNorm output, but no apply-on-load kernel takes this "
                               "convolution (cnuda_conv2d_norm_input_supported); ask batch_norm_act to apply it")
    if not (emit_stats and EPILOGUE_STATS):
        return _Conv2d.apply(x, weight, bias, stride, padding, float(act_slope), pack_token, None, norm, None)
    # emit_stats: the caller's next layer is a train-mode BatchNorm over y.  Where the kernel can, it leaves
    # sum / sum of squares per (pixel block, channel) beside y; batch_norm_act finds them on the tensor.
    box = []
    y = _Conv2d.apply(x, weight, bias, stride, padding, float(act_slope), pack_token, box, norm, None)
    if box:
        y._cnuda_bn_stats = box[0]
    return y


class _ConvActConv1x1(Function):
    """A detection head, `nn.Sequential(Conv2d(C, Ch, k, padding) + ReLU, Conv2d(Ch, Co, 1))` (dla.py:474-483), as one
    tape node: the two forward launches are the ones two `_Conv2d` nodes make; backward computes the hidden map's
    gradient -- the 1x1 layer's input gradient times the ReLU's -- in one pass over the hidden map
    (cnuda_conv1x1_backward_data_act) instead of a K <= 8 GEMM launch followed by cnuda_act_backward's own pass."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, padding, act_slope, token1, token2, lead=None):
        require_gpu(x, w1, b1, w2, b2)
        # lead: only the first `lead` images of x go through the head (a UDA step's source half: forward_domains) -- the
        # gradient still covers all of x, zero beyond them, or simply added into what x's gradient slot already holds
        ctx.slot, ctx.full = slot_of(x), None
        if lead is not None and lead < x.shape[0]:
            ctx.full = x.shape
            x = f32c(x)[:lead]
        x, w1, w2 = f32c(x), f32c(w1), f32c(w2)
        b1 = None if b1 is None else f32c(b1)
        b2 = None if b2 is None else f32c(b2)
        g1 = _conv_geom(x, w1, 1, padding)
        B, C, H, W, Ch, kh, kw, sh, sw, ph, pw = g1
        Ho, Wo = H + 2 * ph - kh + 1, W + 2 * pw - kw + 1
        hidden = torch.empty((B, Ch, Ho, Wo), dtype=torch.float32, device=x.device)
        L = lib()
        wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g1), x)
        prof_arm('conv_fwd', B, C, H, W, Ch, kh, kw, Ho, Wo)
        with pack_stamp(token1, w1):
            check(L.cnuda_conv2d_forward(ptr(x), ptr(w1), ptr(b1), ptr(hidden), *g1, float(act_slope), wp, wn, stream()),
                  'conv2d_forward')
        g2 = _conv_geom(hidden, w2, 1, 0)
        Co = g2[4]
        y = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=x.device)
        wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g2), x)
        prof_arm('conv_fwd', B, Ch, Ho, Wo, Co, 1, 1, Ho, Wo)
        with pack_stamp(token2, w2):
            check(L.cnuda_conv2d_forward(ptr(hidden), ptr(w2), ptr(b2), ptr(y), *g2, -1.0, wp, wn, stream()),
                  'conv2d_forward')
        ctx.g1, ctx.g2, ctx.act_slope, ctx.token1 = g1, g2, float(act_slope), token1
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        ctx.save_for_backward(x, w1, b1, w2, b2, hidden)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w1, b1, w2, b2, hidden = ctx.saved_tensors
        g1, g2 = ctx.g1, ctx.g2
        L = lib()
        gy = f32c(gy)
        if gy.data_ptr() % 16:                 # (a view at an odd offset: the fused pass reads 16 bytes at a time)
            gy = gy.clone()
        B, C, H, W, Ch, kh, kw = g1[:7]
        Co, Ho, Wo = g2[4], hidden.shape[2], hidden.shape[3]
        gx = gw1 = gb1 = gw2 = gb2 = None
        if ctx.needs_input_grad[3] or (ctx.has_b2 and ctx.needs_input_grad[4]):
            gw2_buf, gw2 = _param_grad(w2)
            gb2_buf, gb2 = _param_grad(b2, ctx.has_b2)
            wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g2), x)
            prof_arm('conv_wgrad', B, Ch, Ho, Wo, Co, 1, 1, Ho, Wo)
            check(L.cnuda_conv2d_backward_weight(ptr(hidden), ptr(gy), ptr(gw2_buf), ptr(gb2_buf), *g2, wp, wn, stream()),
                  'conv2d_backward_weight')
        gh = torch.empty_like(hidden)
        check(L.cnuda_conv1x1_backward_data_act(ptr(gy), ptr(w2), ptr(hidden), ptr(gh), B, Co, Ch, Ho * Wo,
                                                ctx.act_slope, stream()), 'conv1x1_backward_data_act')
        wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g1), x)
        if ctx.needs_input_grad[0]:
            prof_arm('conv_dgrad', B, C, H, W, Ch, kh, kw, Ho, Wo)
            slot = ctx.slot
            if ctx.full is None:
                gx, addend, addend2 = accumulate_target(slot, x)
                part = gx
            elif slot is not None and slot.buf is not None and slot.owned:
                gx = slot.buf                                   # the leading images' share lands on top of the slot's total
                part = addend = gx[:B]
                addend2 = None
                slot.included.append(gx)
            else:
                gx = claim(slot, torch.empty(ctx.full, dtype=torch.float32, device=x.device))
                gx[B:].zero_()
                part, addend, addend2 = gx[:B], None, None
            with pack_stamp(ctx.token1, w1):
                check(L.cnuda_conv2d_backward_data_add(ptr(gh), ptr(w1), ptr(addend), ptr(addend2), ptr(part), *g1, wp, wn,
                                                       stream()), 'conv2d_backward_data')
        if ctx.needs_input_grad[1] or (ctx.has_b1 and ctx.needs_input_grad[2]):
            gw1_buf, gw1 = _param_grad(w1)
            gb1_buf, gb1 = _param_grad(b1, ctx.has_b1)
            prof_arm('conv_wgrad', B, C, H, W, Ch, kh, kw, Ho, Wo)
            check(L.cnuda_conv2d_backward_weight(ptr(x), ptr(gh), ptr(gw1_buf), ptr(gb1_buf), *g1, wp, wn, stream()),
                  'conv2d_backward_weight')
        return gx, gw1, gb1, gw2, gb2, None, None, None, None, None


def conv_act_conv1x1_supported(x, conv, last):
    """The fused head node applies: stride-1 `conv` with an activation, a 1x1 `last` with 1..8 outputs, planes of a
    multiple of 4 pixels, and a tape being recorded."""
    if not (torch.is_grad_enabled() and x.is_cuda and x.dim() == 4):
        return False
    (kh, kw), (ph, pw) = conv.kernel_size, conv.padding
    hw = (x.shape[2] + 2 * ph - kh + 1) * (x.shape[3] + 2 * pw - kw + 1)
    return (conv.act_slope >= 0 and conv.stride == (1, 1) and last.kernel_size == (1, 1) and last.stride == (1, 1)
            and last.padding == (0, 0) and last.act_slope < 0 and 1 <= last.out_channels <= 8 and hw > 0 and hw % 4 == 0)


def conv_act_conv1x1(x, conv, last, lead=None):
    """last(conv(x)) for the two `hip_runtime.nn.Conv2d` layers of a detection head, recorded as one tape node.
    lead: evaluate the first `lead` images of x only (the result has `lead` images; x's gradient is whole)."""
    return _ConvActConv1x1.apply(x, conv.weight, conv.bias, last.weight, last.bias, conv.padding, conv.act_slope,
                                 conv._pack_token, last._pack_token, lead)


def conv2d_infer(x, weight, bias=None, stride=1, padding=0, act_slope=-1.0, residual=None, pack_token=0,
                 pack_version=None):
    """Tape-free y = act(conv2d(x, weight) + bias + residual): the forward kernel with the skip connection in its
    epilogue -- what a BatchNorm-folded BasicBlock needs (export.py).  No autograd node is created."""
    require_gpu(x, weight, bias, residual)
    x, weight = f32c(x.detach()), f32c(weight.detach())
    bias = None if bias is None else f32c(bias.detach())
    g = _conv_geom(x, weight, stride, padding)
    B, C, H, W, Co, kh, kw, sh, sw, ph, pw = g
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    if residual is not None:
        residual = f32c(residual.detach())
        if tuple(residual.shape) != (B, Co, Ho, Wo):
            raise RuntimeError("conv2d_infer: residual %s does not match the output %s"
                               % (tuple(residual.shape), (B, Co, Ho, Wo)))
    y = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=x.device)
    L = lib()
    wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g), x)
    prof_arm('conv_fwd', B, C, H, W, Co, kh, kw, Ho, Wo)
    with pack_stamp(pack_token, weight, pack_version):
        check(L.cnuda_conv2d_forward_res(ptr(x), ptr(weight), ptr(bias), ptr(residual), ptr(y), *g, float(act_slope),
                                         wp, wn, stream()), 'conv2d_forward')
    return y


class _ConvTranspose2d(Function):
    """nn.ConvTranspose2d(Cin, Cout, k, stride, padding, bias=False) (backends/resnet.py:86-94): the input
    gradient of the convolution whose weight is this [Cin, Cout, kh, kw] tensor read as [Cout_conv, Cin_conv, ...];
    its own gradients are that convolution's forward (grad_x) and weight gradient (roles of x and grad_y
    swapped) -- the same three C entry points as _Conv2d."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, output_padding):
        require_gpu(x, weight)
        x, weight = f32c(x), f32c(weight)
        B, Ci, H, W = x.shape
        if weight.shape[0] != Ci:
            raise RuntimeError("conv_transpose2d: input has %d channels, weight expects %d" % (Ci, weight.shape[0]))
        Co, kh, kw = weight.shape[1], weight.shape[2], weight.shape[3]
        (sh, sw), (ph, pw), (oph, opw) = _pair(stride), _pair(padding), _pair(output_padding)
        Ho, Wo = (H - 1) * sh - 2 * ph + kh + oph, (W - 1) * sw - 2 * pw + kw + opw
        # geometry of the convolution  y[B,Co,Ho,Wo] -> x[B,Ci,H,W]
        g = (B, Co, Ho, Wo, Ci, kh, kw, sh, sw, ph, pw)
        if (Ho + 2 * ph - kh) // sh + 1 != H or (Wo + 2 * pw - kw) // sw + 1 != W:
            raise RuntimeError("conv_transpose2d: output_padding %s inconsistent with stride %s" %
                               ((oph, opw), (sh, sw)))
        y = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=x.device)
        L = lib()
        wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g), x)
        prof_arm('conv_dgrad', B, Co, Ho, Wo, Ci, kh, kw, H, W)
        check(L.cnuda_conv2d_backward_data(ptr(x), ptr(weight), ptr(y), *g, wp, wn, stream()),
              'conv_transpose2d(forward)')
        ctx.geom = g
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        g = ctx.geom
        B, Co, Ho, Wo, Ci, kh, kw = g[:7]
        H, W = x.shape[2], x.shape[3]
        gy = f32c(gy)
        L = lib()
        wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g), x)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            prof_arm('conv_fwd', B, Co, Ho, Wo, Ci, kh, kw, H, W)
            check(L.cnuda_conv2d_forward(ptr(gy), ptr(weight), None, ptr(gx), *g, -1.0, wp, wn, stream()),
                  'conv_transpose2d(backward data)')
        if ctx.needs_input_grad[1]:
            gw_buf, gw = _param_grad(weight)
            prof_arm('conv_wgrad', B, Co, Ho, Wo, Ci, kh, kw, H, W)
            check(L.cnuda_conv2d_backward_weight(ptr(gy), ptr(x), ptr(gw_buf), None, *g, wp, wn, stream()),
                  'conv_transpose2d(backward weight)')
        return gx, gw, None, None, None


def conv_transpose2d(x, weight, stride=1, padding=0, output_padding=0):
    return _ConvTranspose2d.apply(x, weight, stride, padding, output_padding)


# ---------------------------------------------------------------------------
# batch norm (+ residual add + ReLU)
# ---------------------------------------------------------------------------
def _act_code(relu):
    """fused activation of the BN kernels: False / True (ReLU) / 6 (ReLU6)"""
    if relu is True or relu is False or relu is None:
        return 1 if relu else 0
    if relu == 6:
        return 2
    raise ValueError("batch_norm_act: relu must be False, True or 6, got %r" % (relu,))


class _BatchNormAct(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, residual, running_mean, running_var, momentum, eps, relu, num_batches_tracked,
                groups, pre=None, defer=False):
        require_gpu(x, gamma, beta, residual)
        ctx.res_slot = None if residual is None else slot_of(residual)
        x = f32c(x)
        residual = None if residual is None else f32c(residual)
        B, C = x.shape[0], x.shape[1]
        if B % groups:
            raise RuntimeError("batch_norm_act: batch of %d images cannot be split into %d domain groups" % (B, groups))
        HW = x.numel() // (B * C)
        mean = torch.empty(groups * C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(groups * C, dtype=torch.float32, device=x.device)
        L = lib()
        wp, wn = _ws(L.cnuda_bn_workspace_bytes(B, C, HW), x)
        bpg = 0
        if pre is not None and pre[2] >= C:
            # blocks per statistics group: whole images' worth of row pieces, or whole blocks of the flattened pixel axis
            if pre[3]:
                bpg = B // groups * pre[3]
            elif (B // groups * HW) % pre[1] == 0:
                bpg = B // groups * HW // pre[1]
        # defer: the statistics only -- y stays unwritten, its consumer (ops.conv2d) normalises x while it stages it
        ctx.deferred = bool(defer and bpg and residual is None and relu in (True, 1))
        # deferred: nothing is ever written, so the output is a 4-byte placeholder of the right SHAPE (all strides 0) --
        # no 537-MB allocation at the stem, and an accidental reader (f32c raises on the mark) cannot mistake it for data
        y = torch.empty_strided(x.shape, (0,) * x.dim(), dtype=torch.float32, device=x.device) if ctx.deferred \
            else torch.empty_like(x)
        if bpg:
            # sum(x) / sum(x^2) came with x from the producing GEMM's epilogue: no statistics pass over x
            check(L.cnuda_bn_train_forward_stats(ptr(x), ptr(pre[0]), bpg, pre[2], ptr(gamma), ptr(beta), ptr(residual),
                                                 ptr(None if ctx.deferred else y), ptr(mean), ptr(invstd), ptr(running_mean), ptr(running_var),
                                                 ptr(num_batches_tracked), float(momentum), float(eps), _act_code(relu),
                                                 B, C, HW, groups, wp, wn, stream()), 'bn_train_forward_stats')
        else:
            check(L.cnuda_bn_train_forward(ptr(x), ptr(gamma), ptr(beta), ptr(residual), ptr(y), ptr(mean), ptr(invstd),
                                           ptr(running_mean), ptr(running_var), ptr(num_batches_tracked), float(momentum),
                                           float(eps),
                                           _act_code(relu), B, C, HW, groups, wp, wn, stream()), 'bn_train_forward')
        ctx.relu, ctx.dims, ctx.has_res, ctx.groups = relu, (B, C, HW), residual is not None, groups
        # y is read by the backward only where a residual entered the activation; without one the gate is recomputed
        # from x (cnuda_bn_backward, `beta` given) and y is not kept
        ctx.save_for_backward(x, y if (relu and residual is not None) else None, gamma, mean, invstd, beta)
        if ctx.deferred:
            ctx.mark_non_differentiable(mean, invstd)
            return y, mean, invstd
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy, *_unused):
        x, y, gamma, mean, invstd, beta = ctx.saved_tensors
        B, C, HW = ctx.dims
        gy = f32c(gy)
        gx = torch.empty_like(x)
        gres = first_writer(ctx.res_slot, x) if ctx.has_res else None
        gg_buf, gg = _param_grad(gamma)
        gb_buf, gb = _param_grad(beta)
        L = lib()
        wp, wn = _ws(L.cnuda_bn_workspace_bytes(B, C, HW), x)
        # without a residual the activation's gate is recomputed from x (y is not read): beta goes along for that
        regate = beta if (ctx.relu and not ctx.has_res) else None
        check(L.cnuda_bn_backward(ptr(gy), ptr(x), ptr(y), ptr(gamma), ptr(regate), ptr(mean), ptr(invstd), ptr(gx),
                                  ptr(gres), ptr(gg_buf), ptr(gb_buf), _act_code(ctx.relu), B, C, HW, ctx.groups, wp, wn,
                                  stream()), 'bn_backward')
        return gx, gg, gb, gres, None, None, None, None, None, None, None, None, None


def batch_norm_act(x, gamma, beta, running_mean, running_var, training, momentum=0.1, eps=1e-5,
                   residual=None, relu=False, num_batches_tracked=None, groups=None, defer_apply=False):
    """`num_batches_tracked` (int64 scalar buffer, optional) is incremented by the statistics kernel in training
    mode, like nn.BatchNorm2d.forward does on the host.  `groups` (default: hip_runtime.current_groups()):
    statistics groups of a batch that carries several domains, see hip_runtime.domain_groups.
    defer_apply (training, ReLU, no residual, statistics from the producing GEMM's epilogue; ignored otherwise): the
    returned tensor is NOT written -- it carries `_cnuda_deferred_bn`, and the one consumer the caller vouches for,
    ops.conv2d, normalises x while it stages it (apply on load: one write and one read of the activation less)."""
    if groups is None:
        from . import current_groups
        groups = current_groups()
    if training:
        if num_batches_tracked is not None and (num_batches_tracked.dtype != torch.int64 or
                                                not num_batches_tracked.is_cuda):
            raise RuntimeError("batch_norm_act: num_batches_tracked must be an int64 tensor on the GPU")
        from . import bump_buffer_epoch
        bump_buffer_epoch()         # the kernel rewrites the running statistics behind torch's version counters
        out = _BatchNormAct.apply(x, gamma, beta, residual, running_mean, running_var, momentum, eps, relu,
                                  num_batches_tracked, int(groups), getattr(x, '_cnuda_bn_stats', None), bool(defer_apply))
        if isinstance(out, tuple):                  # deferred: (unwritten y, mean, invstd)
            y, mean, invstd = out
            y._cnuda_deferred_bn = (f32c(x).detach(), mean, invstd, f32c(gamma).detach(), f32c(beta).detach(),
                                    x.shape[0] // int(groups))
            return y
        return out
    if torch.is_grad_enabled() and (x.requires_grad or gamma.requires_grad):
        raise RuntimeError("batch_norm_act: eval-mode BN has no backward in this build "
                           "(the reference evaluates under torch.no_grad(), train.py:172)")
    require_gpu(x, gamma, beta, residual)
    x = f32c(x)
    residual = None if residual is None else f32c(residual)
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    check(lib().cnuda_bn_eval_forward(ptr(x), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                      ptr(residual), ptr(y), float(eps), _act_code(relu), B, C, HW, stream()),
          'bn_eval_forward')
    return y


# ---------------------------------------------------------------------------
# spatial
# ---------------------------------------------------------------------------
class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x, k):
        require_gpu(x)
        ctx.slot = slot_of(x)
        x = f32c(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, H // k, W // k), dtype=torch.float32, device=x.device)
        check(lib().cnuda_maxpool2d_forward(ptr(x), ptr(y), B, C, H, W, k, stream()), 'maxpool2d_forward')
        ctx.k = k
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        slot = ctx.slot
        if slot is not None and slot.buf is not None and slot.owned:
            # another consumer's share is already in the slot's own buffer: only the arg-max cells are touched (+=)
            gx, acc = slot.buf, 1
            slot.included.append(gx)
        else:
            gx, acc = first_writer(slot, x), 0
        check(lib().cnuda_maxpool2d_backward_acc(ptr(x), ptr(f32c(gy)), ptr(gx), acc, B, C, H, W, ctx.k, stream()),
              'maxpool2d_backward')
        return gx, None


class _MaxPoolWindow(Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        require_gpu(x)
        x = f32c(x)
        B, C, H, W = x.shape
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        y = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
        check(lib().cnuda_maxpool2d_window_forward(ptr(x), ptr(y), B, C, H, W, k, s, p, stream()),
              'maxpool2d_window_forward')
        ctx.ksp = (k, s, p)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        gx = torch.empty_like(x)
        check(lib().cnuda_maxpool2d_window_backward(ptr(x), ptr(f32c(gy)), ptr(gx), B, C, H, W, *ctx.ksp, stream()),
              'maxpool2d_window_backward')
        return gx, None, None, None


def max_pool2d(x, k, stride=None, padding=0):
    """nn.MaxPool2d(k, stride, padding) (floor mode).  stride == k without padding is the DLA downsample
    (dla.py:202-203); the general window is torchvision's ResNet stem pool."""
    k = int(k)
    stride = k if stride is None else int(stride)
    if stride == k and padding == 0:
        return _MaxPool.apply(x, k)
    return _MaxPoolWindow.apply(x, k, stride, int(padding))


class _DwConvT(Function):
    @staticmethod
    def forward(ctx, x, weight, stride, padding, skip=None):
        require_gpu(x, weight)
        x, weight = f32c(x), f32c(weight)
        B, C, H, W = x.shape
        k = weight.shape[2]
        if weight.shape[0] != C or weight.shape[1] != 1 or weight.shape[3] != k:
            raise RuntimeError("depthwise conv_transpose2d: weight %s does not fit %d channels"
                               % (tuple(weight.shape), C))
        Ho, Wo = (H - 1) * stride - 2 * padding + k, (W - 1) * stride - 2 * padding + k
        y = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
        if skip is not None:
            require_gpu(skip)
            skip = f32c(skip)
            if skip.shape != y.shape:
                raise RuntimeError("depthwise conv_transpose2d: summand %s does not fit the output %s"
                                   % (tuple(skip.shape), tuple(y.shape)))
        check(lib().cnuda_dwconvt2d_add_forward(ptr(x), ptr(weight), ptr(skip), ptr(y), B, C, H, W, k, stride, padding,
                                                stream()), 'dwconvt2d_forward')
        ctx.geom = (B, C, H, W, k, stride, padding)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw_buf, gw = _param_grad(weight, ctx.needs_input_grad[1])
        L = lib()
        B, C, _, _, k = ctx.geom[:5]
        wp, wn = _ws(L.cnuda_dwconvt2d_workspace_bytes(B, C, k), x)
        check(L.cnuda_dwconvt2d_backward(ptr(x), ptr(weight), ptr(f32c(gy)), ptr(gx), ptr(gw_buf), *ctx.geom,
                                         wp, wn, stream()), 'dwconvt2d_backward')
        return gx, gw, None, None, (gy if ctx.needs_input_grad[4] else None)


def depthwise_conv_transpose2d(x, weight, stride, padding, skip=None):
    """Depthwise ConvTranspose2d; with `skip` the result is `conv_transpose(x) + skip` written in one pass."""
    return _DwConvT.apply(x, weight, int(stride), int(padding), skip)


class _DwConv(Function):
    """Depthwise convolution (groups == channels), weight [C,1,k,k], no bias."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding):
        require_gpu(x, weight)
        x, weight = f32c(x), f32c(weight)
        B, C, H, W = x.shape
        k = weight.shape[2]
        if weight.shape[0] != C or weight.shape[1] != 1 or weight.shape[3] != k:
            raise RuntimeError("depthwise conv2d: weight %s does not fit %d channels" % (tuple(weight.shape), C))
        Ho, Wo = (H + 2 * padding - k) // stride + 1, (W + 2 * padding - k) // stride + 1
        y = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
        check(lib().cnuda_dwconv2d_forward(ptr(x), ptr(weight), ptr(y), B, C, H, W, k, stride, padding, stream()),
              'dwconv2d_forward')
        ctx.geom = (B, C, H, W, k, stride, padding)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gw_buf, gw = _param_grad(weight, ctx.needs_input_grad[1])
        L = lib()
        B, C, _, _, k = ctx.geom[:5]
        wp, wn = _ws(L.cnuda_dwconv2d_workspace_bytes(B, C, k), x)
        check(L.cnuda_dwconv2d_backward(ptr(x), ptr(weight), ptr(f32c(gy)), ptr(gx), ptr(gw_buf), *ctx.geom,
                                        wp, wn, stream()), 'dwconv2d_backward')
        return gx, gw, None, None


def depthwise_conv2d(x, weight, stride=1, padding=0):
    return _DwConv.apply(x, weight, int(stride), int(padding))


class _Add(Function):
    @staticmethod
    def forward(ctx, a, b):
        require_gpu(a, b)
        if a.shape != b.shape:
            raise RuntimeError("add: shapes %s and %s differ" % (tuple(a.shape), tuple(b.shape)))
        a, b = f32c(a), f32c(b)
        out = torch.empty_like(a)
        check(lib().cnuda_add(ptr(a), ptr(b), ptr(out), a.numel(), stream()), 'add')
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


class _Cat(Function):
    @staticmethod
    def forward(ctx, *xs):
        require_gpu(*xs)
        slots = [slot_of(t) for t in xs]
        xs = [f32c(t) for t in xs]
        B, _, H, W = xs[0].shape
        chans = [t.shape[1] for t in xs]
        out = torch.empty((B, sum(chans), H, W), dtype=torch.float32, device=xs[0].device)
        off, L = 0, lib()
        for t, c in zip(xs, chans):
            if t.shape[0] != B or t.shape[2] != H or t.shape[3] != W:
                raise RuntimeError("cat: mismatching shapes")
            check(L.cnuda_copy_channels(ptr(t), ptr(out), B, c, H * W, c, 0, sum(chans), off, stream()), 'cat')
            off += c
        ctx.chans = chans
        ctx.slots = slots
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = f32c(g)
        B, Ct, H, W = g.shape
        outs, off, L = [], 0, lib()
        for i, c in enumerate(ctx.chans):
            if ctx.needs_input_grad[i]:
                t = claim(ctx.slots[i], torch.empty((B, c, H, W), dtype=torch.float32, device=g.device))
                check(L.cnuda_copy_channels(ptr(g), ptr(t), B, c, H * W, Ct, off, c, 0, stream()), 'cat_backward')
                outs.append(t)
            else:
                outs.append(None)
            off += c
        return tuple(outs)


def cat_channels(xs):
    return _Cat.apply(*xs)


def _ptr_array(ts):
    import ctypes
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


class _CatConv1x1(Function):
    """y = conv1x1(cat(xs, 1), weight) without the concatenation (cnuda_conv2d_cat_*): DLA's Root (backends/dla.py
    Root.forward; reference dla.py:150-168).  The input gradient of every source goes to that source's own tensor, with the
    other consumers' shares of its gradient added in the epilogue (hip_runtime.fanout), like _Conv2d's."""

    @staticmethod
    def forward(ctx, weight, pack_token, stats_box, *xs):
        import ctypes
        require_gpu(weight, *xs)
        ctx.slots = [slot_of(t) for t in xs]
        xs = [f32c(t) for t in xs]
        weight = f32c(weight)
        B, _, H, W = xs[0].shape
        cs = [int(t.shape[1]) for t in xs]
        Co = weight.shape[0]
        for t in xs:
            if t.shape[0] != B or t.shape[2] != H or t.shape[3] != W:
                raise RuntimeError("conv1x1_cat: mismatching shapes")
        if weight.shape[1] != sum(cs) or weight.shape[2] != 1 or weight.shape[3] != 1:
            raise RuntimeError("conv1x1_cat: weight %s for %d concatenated channels" % (tuple(weight.shape), sum(cs)))
        g = (B, sum(cs), H, W, Co, 1, 1, 1, 1, 0, 0)
        L = lib()
        y = torch.empty((B, Co, H, W), dtype=torch.float32, device=weight.device)
        wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g), y)
        stats = None
        if stats_box is not None:
            rows, bpi = ctypes.c_int(0), ctypes.c_int(0)
            blk = L.cnuda_conv2d_stats_block(*g, ctypes.byref(rows), ctypes.byref(bpi))
            if blk:
                nblk = B * bpi.value if bpi.value else (B * H * W + 127) // 128 * (128 // blk)
                stats = torch.empty((nblk, rows.value, 2), dtype=torch.float32, device=weight.device)
                stats_box.append((stats, 0 if bpi.value else blk, rows.value, bpi.value))
        cs_arr = (ctypes.c_int * len(cs))(*cs)
        prof_arm('conv_fwd', B, sum(cs), H, W, Co, 1, 1, H, W)
        with pack_stamp(pack_token, weight):
            check(L.cnuda_conv2d_cat_forward(_ptr_array(xs), cs_arr, len(xs), ptr(weight), None, ptr(y), ptr(stats), -1.0,
                                             B, H, W, Co, wp, wn, stream()), 'conv2d_cat_forward')
        ctx.cs, ctx.pack_token = cs, pack_token
        ctx.save_for_backward(weight, *xs)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        import ctypes
        weight, xs = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        cs, n = ctx.cs, len(ctx.cs)
        B, _, H, W = xs[0].shape
        Co = weight.shape[0]
        g = (B, sum(cs), H, W, Co, 1, 1, 1, 1, 0, 0)
        L = lib()
        gy = f32c(gy)
        cs_arr = (ctypes.c_int * n)(*cs)
        wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g), gy)
        gxs = [None] * n
        if any(ctx.needs_input_grad[3:]):
            outs, adds, add2s = [], [], []
            for i, x in enumerate(xs):
                if ctx.needs_input_grad[3 + i]:
                    out, a1, a2 = accumulate_target(ctx.slots[i], x)     # the slot's content is summed in the epilogue
                    gxs[i] = out
                else:
                    out, a1, a2 = torch.empty_like(x), None, None        # (a source outside the tape: its rows go to scratch)
                outs.append(out)
                adds.append(a1)
                add2s.append(a2)
            prof_arm('conv_dgrad', B, sum(cs), H, W, Co, 1, 1, H, W)
            with pack_stamp(ctx.pack_token, weight):
                check(L.cnuda_conv2d_cat_backward_data(ptr(gy), ptr(weight), _ptr_array(outs), _ptr_array(adds),
                                                       _ptr_array(add2s), cs_arr, n, B, H, W, Co, wp, wn, stream()),
                      'conv2d_cat_backward_data')
        gw = None
        if ctx.needs_input_grad[0]:
            gw_buf, gw = _param_grad(weight)
            prof_arm('conv_wgrad', B, sum(cs), H, W, Co, 1, 1, H, W)
            check(L.cnuda_conv2d_cat_backward_weight(_ptr_array(xs), cs_arr, n, ptr(gy), ptr(gw_buf), B, H, W, Co, wp, wn,
                                                     stream()), 'conv2d_cat_backward_weight')
        return (gw, None, None) + tuple(gxs)


def _cat_supported(xs, weight):
    import ctypes
    xs = list(xs)
    if not (2 <= len(xs) <= 4) or any(t.dim() != 4 or getattr(t, '_cnuda_deferred_bn', None) is not None for t in xs):
        return False
    if weight.dim() != 4 or weight.shape[2] != 1 or weight.shape[3] != 1:
        return False
    B, _, H, W = xs[0].shape
    cs = [int(t.shape[1]) for t in xs]
    return bool(lib().cnuda_conv2d_cat_supported((ctypes.c_int * len(cs))(*cs), len(cs), B, H, W, int(weight.shape[0])))


def conv1x1_cat_infer(xs, weight, bias=None, act_slope=-1.0, pack_token=0, pack_version=None):
    """Tape-free y = act(conv1x1(cat(xs, 1), weight) + bias) without the concatenation (a BatchNorm-folded Root, export.py),
    or None where no kernel takes the sources (the caller concatenates)."""
    import ctypes
    xs = list(xs)
    if not _cat_supported(xs, weight):
        return None
    require_gpu(weight, bias, *xs)
    xs = [f32c(t.detach()) for t in xs]
    weight = f32c(weight.detach())
    bias = None if bias is None else f32c(bias.detach())
    B, _, H, W = xs[0].shape
    cs = [int(t.shape[1]) for t in xs]
    Co = weight.shape[0]
    g = (B, sum(cs), H, W, Co, 1, 1, 1, 1, 0, 0)
    L = lib()
    y = torch.empty((B, Co, H, W), dtype=torch.float32, device=weight.device)
    wp, wn = _ws(L.cnuda_conv2d_workspace_bytes(*g), y)
    prof_arm('conv_fwd', B, sum(cs), H, W, Co, 1, 1, H, W)
    with pack_stamp(pack_token, weight, pack_version):
        check(L.cnuda_conv2d_cat_forward(_ptr_array(xs), (ctypes.c_int * len(cs))(*cs), len(xs), ptr(weight), ptr(bias), ptr(y),
                                         None, float(act_slope), B, H, W, Co, wp, wn, stream()), 'conv2d_cat_forward')
    return y


def conv1x1_cat(xs, weight, pack_token=0, emit_stats=False):
    """conv2d(cat(xs, 1), weight) for a 1x1 / stride 1 / bias-free convolution, or None where no kernel takes the sources as
    they are (cnuda_conv2d_cat_supported: 2 .. 4 sources of multiples of 64 channels, H * W % 4 == 0, no K split in the
    plan): the caller then concatenates.  emit_stats as ops.conv2d."""
    xs = list(xs)
    if not _cat_supported(xs, weight):
        return None
    if not (emit_stats and EPILOGUE_STATS):
        return _CatConv1x1.apply(weight, pack_token, None, *xs)
    box = []
    y = _CatConv1x1.apply(weight, pack_token, box, *xs)
    if box:
        y._cnuda_bn_stats = box[0]
    return y


class _SplitOffsetMask(Function):
    @staticmethod
    def forward(ctx, om):
        require_gpu(om)
        om = f32c(om)
        B, C3, H, W = om.shape
        if C3 % 3:
            raise RuntimeError("conv_offset_mask output must have 3*taps channels, got %d" % C3)
        T = C3 // 3
        offset = torch.empty((B, 2 * T, H, W), dtype=torch.float32, device=om.device)
        mask = torch.empty((B, T, H, W), dtype=torch.float32, device=om.device)
        check(lib().cnuda_split_offset_mask(ptr(om), ptr(offset), ptr(mask), B, T, H * W, stream()),
              'split_offset_mask')
        ctx.save_for_backward(mask)
        return offset, mask

    @staticmethod
    @once_differentiable
    def backward(ctx, goff, gmask):
        (mask,) = ctx.saved_tensors
        B, T, H, W = mask.shape
        gom = torch.empty((B, 3 * T, H, W), dtype=torch.float32, device=mask.device)
        check(lib().cnuda_split_offset_mask_backward(ptr(f32c(goff)), ptr(f32c(gmask)), ptr(mask), ptr(gom), B, T,
                                                     H * W, stream()), 'split_offset_mask_backward')
        return gom


def split_offset_mask(om):
    return _SplitOffsetMask.apply(om)


# ---------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------
def _loss_ws(like):
    return _ws(lib().cnuda_loss_workspace_bytes(), like)


class _Focal(Function):
    """(loss, prob) = focal(clamp(sigmoid(logits)), gt)."""

    @staticmethod
    def forward(ctx, logits, gt, weight):
        require_gpu(logits, gt)
        logits, gt = f32c(logits), f32c(gt)
        if logits.shape != gt.shape:
            raise RuntimeError("focal loss: prediction %s vs target %s" % (tuple(logits.shape), tuple(gt.shape)))
        prob = torch.empty_like(logits)
        out2 = torch.empty(2, dtype=torch.float32, device=logits.device)
        wp, wn = _loss_ws(logits)
        check(lib().cnuda_focal_loss_forward(ptr(logits), ptr(gt), ptr(prob), ptr(out2), logits.numel(),
                                             float(weight), wp, wn, stream()), 'focal_loss_forward')
        ctx.weight = weight
        ctx.save_for_backward(logits, gt, out2)
        den = out2[1].clone()                      # num_pos
        ctx.mark_non_differentiable(prob, den)
        return out2[0].clone(), prob, den

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss, _gprob, _gden):
        logits, gt, out2 = ctx.saved_tensors
        grad = torch.empty_like(logits)
        up = f32c(gloss.reshape(1))
        check(lib().cnuda_focal_loss_backward(ptr(logits), ptr(gt), ptr(out2), ptr(up), ptr(grad), logits.numel(),
                                              float(ctx.weight), stream()), 'focal_loss_backward')
        return grad, None, None


def focal_loss(logits, gt, weight=1.0, return_den=False):
    """-> (loss, prob[, num_pos]).  num_pos (a device scalar) is the divisor the loss used (0: none, sum of the
    negative terms only)."""
    loss, prob, den = _Focal.apply(logits, gt, weight)
    return (loss, prob, den) if return_den else (loss, prob)


class _RegL1(Function):
    @staticmethod
    def forward(ctx, feat, mask, ind, target, periodic, weight, angle_weight):
        require_gpu(feat, mask, ind, target)
        feat = f32c(feat)
        if mask.dtype != torch.uint8 or ind.dtype != torch.int64 or target.dtype != torch.float32:
            raise RuntimeError("reg_l1: expected reg_mask uint8, ind int64, target float32 "
                               "(datasets/coco.py:168-174), got %s/%s/%s" % (mask.dtype, ind.dtype, target.dtype))
        if not (mask.is_contiguous() and ind.is_contiguous() and target.is_contiguous()):
            raise RuntimeError("reg_l1: batch tensors must be contiguous (target is updated in place)")
        B, ch, H, W = feat.shape
        M = ind.shape[1]
        out2 = torch.empty(2, dtype=torch.float32, device=feat.device)
        check(lib().cnuda_reg_l1_forward(ptr(feat), ptr(mask), ptr(ind), ptr(target), ptr(out2), B, M, ch, H * W,
                                         1 if periodic else 0, float(weight), float(angle_weight), stream()),
              'reg_l1_forward')
        ctx.args = (B, M, ch, H * W, 1 if periodic else 0, float(weight), float(angle_weight))
        ctx.save_for_backward(feat, mask, ind, target, out2)
        den = out2[1].clone()                      # expanded-mask sum + 1e-4
        ctx.mark_non_differentiable(den)
        return out2[0].clone(), den

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss, _gden):
        feat, mask, ind, target, out2 = ctx.saved_tensors
        grad = torch.empty_like(feat)
        up = f32c(gloss.reshape(1))
        check(lib().cnuda_reg_l1_backward(ptr(feat), ptr(mask), ptr(ind), ptr(target), ptr(out2), ptr(up), ptr(grad),
                                          *ctx.args, stream()), 'reg_l1_backward')
        return grad, None, None, None, None, None, None


def reg_l1_loss(feat, mask, ind, target, periodic=False, weight=1.0, angle_weight=1.0, return_den=False):
    # `target` is a plain batch tensor (never requires grad); it is masked in place (Q2).
    loss, den = _RegL1.apply(feat, mask, ind, target, periodic, weight, angle_weight)
    return (loss, den) if return_den else loss


class _KpsL1(Function):
    @staticmethod
    def forward(ctx, feat, mask, ind, target, pairs, use_l1, weight, distance_weight):
        require_gpu(feat, mask, ind, target)
        feat = f32c(feat)
        if mask.dtype != torch.uint8 or ind.dtype != torch.int64 or target.dtype != torch.float32:
            raise RuntimeError("kps_l1: expected kp_reg_mask uint8, ind int64, kps float32 "
                               "(datasets/coco.py:183-189), got %s/%s/%s" % (mask.dtype, ind.dtype, target.dtype))
        if not (mask.is_contiguous() and ind.is_contiguous() and target.is_contiguous()):
            raise RuntimeError("kps_l1: batch tensors must be contiguous (target is updated in place)")
        B, ch, H, W = feat.shape
        M = ind.shape[1]
        if ch % 2 or tuple(target.shape) != (B, M, ch) or tuple(mask.shape) != (B, M, ch):
            raise RuntimeError("kps_l1: output %s vs kps %s / kp_reg_mask %s"
                               % (tuple(feat.shape), tuple(target.shape), tuple(mask.shape)))
        P = 0 if pairs is None else pairs.shape[0]
        out2 = torch.empty(2, dtype=torch.float32, device=feat.device)
        ctx.args = (B, M, ch // 2, H * W, P, 1 if use_l1 else 0, float(weight), float(distance_weight))
        check(lib().cnuda_kps_l1_forward(ptr(feat), ptr(mask), ptr(ind), ptr(target), ptr(pairs), ptr(out2), *ctx.args,
                                         stream()), 'kps_l1_forward')
        ctx.pairs = pairs
        ctx.save_for_backward(feat, mask, ind, target, out2)
        den = out2[1].clone()
        ctx.mark_non_differentiable(den)
        return out2[0].clone(), den

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss, _gden):
        feat, mask, ind, target, out2 = ctx.saved_tensors
        grad = torch.empty_like(feat)
        up = f32c(gloss.reshape(1))
        check(lib().cnuda_kps_l1_backward(ptr(feat), ptr(mask), ptr(ind), ptr(target), ptr(ctx.pairs), ptr(out2), ptr(up),
                                          ptr(grad), *ctx.args, stream()), 'kps_l1_backward')
        return grad, None, None, None, None, None, None, None


def kps_l1_loss(feat, mask, ind, target, pairs=None, use_l1=False, weight=1.0, distance_weight=0.1, return_den=False):
    """KPSL1Loss (losses/centernet.py:136-189); `pairs` an int32 device tensor [P, 2] or None; `target` is masked
    in place."""
    loss, den = _KpsL1.apply(feat, mask, ind, target, pairs, use_l1, weight, distance_weight)
    return (loss, den) if return_den else loss


class _SoftmaxLoss(Function):
    @staticmethod
    def forward(ctx, logits, kind):
        require_gpu(logits)
        logits = f32c(logits)
        B, C = logits.shape[0], logits.shape[1]
        HW = logits.numel() // (B * C)
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        wp, wn = _loss_ws(logits)
        check(lib().cnuda_softmax_loss_forward(ptr(logits), ptr(out), B, C, HW, kind, wp, wn, stream()),
              'softmax_loss_forward')
        ctx.args = (B, C, HW, kind)
        ctx.save_for_backward(logits)
        return out[0].clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss):
        (logits,) = ctx.saved_tensors
        grad = torch.empty_like(logits)
        up = f32c(gloss.reshape(1))
        check(lib().cnuda_softmax_loss_backward(ptr(logits), ptr(up), ptr(grad), *ctx.args, stream()),
              'softmax_loss_backward')
        return grad, None


def entropy_loss(hm_logits):
    return _SoftmaxLoss.apply(hm_logits, 0)


def max_square_loss(hm_logits):
    return _SoftmaxLoss.apply(hm_logits, 1)


class _EntropyMap(Function):
    @staticmethod
    def forward(ctx, logits):
        require_gpu(logits)
        logits = f32c(logits)
        B, C = logits.shape[0], logits.shape[1]
        HW = logits.numel() // (B * C)
        out = torch.empty_like(logits)
        check(lib().cnuda_entropy_map_forward(ptr(logits), ptr(out), B, C, HW, stream()), 'entropy_map_forward')
        ctx.args = (B, C, HW)
        ctx.save_for_backward(logits)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (logits,) = ctx.saved_tensors
        grad = torch.empty_like(logits)
        check(lib().cnuda_entropy_map_backward(ptr(logits), ptr(f32c(g)), ptr(grad), *ctx.args, stream()),
              'entropy_map_backward')
        return grad


def entropy_map(hm):
    return _EntropyMap.apply(hm)


class _BceConst(Function):
    @staticmethod
    def forward(ctx, logits, label):
        require_gpu(logits)
        logits = f32c(logits)
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        check(lib().cnuda_bce_const_forward(ptr(logits), float(label), ptr(out), logits.numel(), stream()),
              'bce_const_forward')
        ctx.label = float(label)
        ctx.save_for_backward(logits)
        return out[0].clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss):
        (logits,) = ctx.saved_tensors
        grad = torch.empty_like(logits)
        up = f32c(gloss.reshape(1))
        check(lib().cnuda_bce_const_backward(ptr(logits), ctx.label, ptr(up), ptr(grad), logits.numel(), stream()),
              'bce_const_backward')
        return grad, None


def bce_with_logits_const(logits, label):
    return _BceConst.apply(logits, label)


# ---------------------------------------------------------------------------
# small helpers without gradients
# ---------------------------------------------------------------------------
def sigmoid_clamp_(x):
    """x <- sigmoid(x) in place; returns clamp(x, 1e-4, 1-1e-4) (utils/tensor.py:5-7)."""
    require_gpu(x)
    if x.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("sigmoid_clamp_ is an in-place, gradient-free helper; "
                           "training code goes through focal_loss / reg_l1_loss")
    if not x.is_contiguous() or x.dtype != torch.float32:
        raise RuntimeError("sigmoid_clamp_: expected a contiguous float32 tensor")
    y = torch.empty_like(x)
    check(lib().cnuda_sigmoid_clamp_(ptr(x), ptr(y), x.numel(), stream()), 'sigmoid_clamp_')
    return y


def gather_feat(feat, ind):
    """feat [B,ch,H,W], ind [B,M] int64 -> [B,M,ch] (no gradient; the losses gather internally)."""
    require_gpu(feat, ind)
    feat = f32c(feat.detach())
    B, ch, H, W = feat.shape
    M = ind.shape[1]
    out = torch.empty((B, M, ch), dtype=torch.float32, device=feat.device)
    check(lib().cnuda_gather_feat(ptr(feat), ptr(ind.contiguous()), ptr(out), B, M, ch, H * W, stream()),
          'gather_feat')
    return out


def adam_step_(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step):
    require_gpu(param, grad, exp_avg, exp_avg_sq)
    for t in (param, grad, exp_avg, exp_avg_sq):
        if not t.is_contiguous() or t.dtype != torch.float32 or t.numel() != param.numel():
            raise RuntimeError("adam_step_: flat contiguous float32 tensors of equal length required")
    check(lib().cnuda_adam_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), float(lr),
                                float(beta1), float(beta2), float(eps), float(weight_decay), int(step), stream()),
          'adam_step')
