"""Drop-in for the reference's native pybind module `_ext`
(libs/DCNv2/src/vision.cpp:4-8; imported as `import _ext as _backend` by
libs/DCNv2/dcn_v2.py:13).  Same two function names, same positional argument
order (input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg),
same return values; the work is done by libcenternet_uda_hip.so on the current
HIP stream.  Deformable PSROI pooling is not provided (no backend calls it).
"""
import torch

import hip_runtime as hr


def _shapes(input, weight, offset, mask, kh, kw, dg):
    if input.dim() != 4 or weight.dim() != 4:
        raise RuntimeError("dcn_v2: input and weight must be 4-D")
    B, C, H, W = input.shape
    Co, Ck, wkh, wkw = weight.shape
    if (wkh, wkw) != (kh, kw):
        raise RuntimeError("Input shape and kernel shape wont match: (%d x %d vs %d x %d)." % (kh, kw, wkh, wkw))
    if Ck != C:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (C, Ck))
    return B, C, H, W, Co


def _out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    return ((H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)


def dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                   pad_h, pad_w, dilation_h, dilation_w, deformable_group, _want_columns=False, _act_slope=-1.0,
                   _pack_token=0, _pack_version=None, _stats_box=None):
    hr.require_gpu(input, weight, bias, offset, mask)
    input, weight, bias, offset, mask = [hr.f32c(t) for t in (input, weight, bias, offset, mask)]
    B, C, H, W, Co = _shapes(input, weight, offset, mask, kernel_h, kernel_w, deformable_group)
    Ho, Wo = _out_hw(H, W, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w)
    T = kernel_h * kernel_w
    if tuple(offset.shape) != (B, 2 * T * deformable_group, Ho, Wo) or \
            tuple(mask.shape) != (B, T * deformable_group, Ho, Wo):
        raise RuntimeError("dcn_v2_forward: offset %s / mask %s do not match output %dx%d"
                           % (tuple(offset.shape), tuple(mask.shape), Ho, Wo))
    geom = (B, C, H, W, Co, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
            dilation_h, dilation_w, deformable_group)
    out = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=input.device)
    cols = None
    if _want_columns and W >= 2:          # (deformable_group > 1: the groups' buffers one behind the other, same size)
        cols = torch.empty((B, T * C, Ho * Wo), dtype=torch.float32, device=input.device)
    L = hr.lib()
    nbytes = L.cnuda_dcn_v2_workspace_bytes(*geom)
    ws = hr.workspace(nbytes, input.device)
    if deformable_group == 1 and W >= 2:
        hr.prof_arm('dcn_fwd', B, C, H, W, Co, kernel_h, kernel_w, Ho, Wo)
    # _stats_box (private): the caller's next layer is a train-mode BatchNorm (DeformConv, backends/dla.py:351-372) -- where
    # the kernel can, its epilogue leaves the per-channel sum / sum of squares of `out` per pixel block (appended to the box)
    stats = None
    if _stats_box is not None and _act_slope < 0:
        import ctypes
        rows = ctypes.c_int(0)
        blk = L.cnuda_dcn_v2_stats_block(*geom, ctypes.byref(rows))
        if blk:
            stats = torch.empty(((B * Ho * Wo + 127) // 128 * (128 // blk), rows.value, 2), dtype=torch.float32,
                                device=input.device)
            _stats_box.append((stats, blk, rows.value, 0))
    # _act_slope (not part of the reference's signature): fused epilogue activation of the BatchNorm-folded
    # inference path; -1 = none = the reference's operation
    with hr.pack_stamp(_pack_token, weight, _pack_version):     # (private) identity of the weights: pack cache
        if stats is not None:
            hr.check(L.cnuda_dcn_v2_forward_stats(hr.ptr(input), hr.ptr(weight), hr.ptr(bias), hr.ptr(offset), hr.ptr(mask),
                                                  hr.ptr(out), hr.ptr(cols), hr.ptr(stats), blk, rows.value, *geom, hr.ptr(ws),
                                                  ws.numel(),
                                                  hr.stream()), 'dcn_v2_forward_stats')
        else:
            hr.check(L.cnuda_dcn_v2_forward_act(hr.ptr(input), hr.ptr(weight), hr.ptr(bias), hr.ptr(offset), hr.ptr(mask),
                                                hr.ptr(out), hr.ptr(cols), float(_act_slope), *geom, hr.ptr(ws),
                                                ws.numel(), hr.stream()),
                     'dcn_v2_forward')
    return (out, cols) if _want_columns else out


def dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w,
                    pad_h, pad_w, dilation_h, dilation_w, deformable_group, _columns=None, _grad_weight=None,
                    _grad_bias=None, _grad_input=None):
    """_grad_input: a buffer that already holds another consumer's share of the input's gradient (hip_runtime.fanout):
    the data-gradient walks add into it instead of into a cleared tensor."""
    hr.require_gpu(input, weight, bias, offset, mask, grad_output)
    input, weight, bias, offset, mask, grad_output = [
        hr.f32c(t) for t in (input, weight, bias, offset, mask, grad_output)]
    B, C, H, W, Co = _shapes(input, weight, offset, mask, kernel_h, kernel_w, deformable_group)
    geom = (B, C, H, W, Co, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
            dilation_h, dilation_w, deformable_group)
    grads = [_grad_input if _grad_input is not None else torch.empty_like(input), torch.empty_like(offset),
             torch.empty_like(mask)]
    # the parameter gradients may be written straight into caller-owned buffers (the arena's gradient sink)
    grads += [_grad_weight if _grad_weight is not None else torch.empty_like(weight),
              _grad_bias if _grad_bias is not None else torch.empty_like(bias)]
    L = hr.lib()
    nbytes = L.cnuda_dcn_v2_workspace_bytes(*geom)
    ws = hr.workspace(nbytes, input.device)
    if deformable_group == 1:
        hr.prof_arm('dcn_bwd', B, C, H, W, Co, kernel_h, kernel_w, grad_output.shape[2], grad_output.shape[3])
    hr.check(L.cnuda_dcn_v2_backward_acc(hr.ptr(input), hr.ptr(weight), hr.ptr(bias), hr.ptr(offset), hr.ptr(mask),
                                         hr.ptr(grad_output), hr.ptr(_columns), hr.ptr(grads[0]),
                                         1 if _grad_input is not None else 0, *[hr.ptr(g) for g in grads[1:]], *geom,
                                         hr.ptr(ws), ws.numel(), hr.stream()),
             'dcn_v2_backward')
    return grads        # [grad_input, grad_offset, grad_mask, grad_weight, grad_bias]


# ---------------------------------------------------------------------------------------------------------------------
# Round 6 (not part of the reference's `_ext`): offsets and mask read straight out of `om`, the 3T-channel output of DCN's own
# offset convolution whose mask channels already went through the sigmoid (hip_runtime.ops.conv2d_rowsig) -- see
# include/centernet_uda_hip.h, cnuda_dcn_v2_forward_om.  Used by libs.DCNv2.dcn_v2.DCN only; deformable_group == 1.
# ---------------------------------------------------------------------------------------------------------------------
def dcn_v2_forward_om(input, weight, bias, om, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                      _want_columns=False, _pack_token=0, _stats_box=None, _act_slope=-1.0, _pack_version=None):
    hr.require_gpu(input, weight, bias, om)
    input, weight, bias, om = [hr.f32c(t) for t in (input, weight, bias, om)]
    B, C, H, W = input.shape
    Co = weight.shape[0]
    if weight.shape[1] != C or tuple(weight.shape[2:]) != (kernel_h, kernel_w):
        raise RuntimeError("dcn_v2_forward_om: weight %s does not match %d input channels / a %dx%d kernel"
                           % (tuple(weight.shape), C, kernel_h, kernel_w))
    Ho, Wo = _out_hw(H, W, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w)
    T = kernel_h * kernel_w
    if tuple(om.shape) != (B, 3 * T, Ho, Wo):
        raise RuntimeError("dcn_v2_forward_om: om %s does not match [%d, %d, %d, %d]" % (tuple(om.shape), B, 3 * T, Ho, Wo))
    geom = (B, C, H, W, Co, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, 1)
    out = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=input.device)
    cols = torch.empty((B, T * C, Ho * Wo), dtype=torch.float32, device=input.device) if _want_columns else None
    L = hr.lib()
    ws = hr.workspace(L.cnuda_dcn_v2_workspace_bytes(*geom), input.device)
    hr.prof_arm('dcn_fwd', B, C, H, W, Co, kernel_h, kernel_w, Ho, Wo)
    stats, blk, nrows = None, 0, 0
    if _stats_box is not None and _act_slope < 0:
        import ctypes
        rows = ctypes.c_int(0)
        blk = L.cnuda_dcn_v2_stats_block(*geom, ctypes.byref(rows))
        if blk:
            nrows = rows.value
            stats = torch.empty(((B * Ho * Wo + 127) // 128 * (128 // blk), nrows, 2), dtype=torch.float32, device=input.device)
            _stats_box.append((stats, blk, nrows, 0))
    with hr.pack_stamp(_pack_token, weight, _pack_version):
        hr.check(L.cnuda_dcn_v2_forward_om(hr.ptr(input), hr.ptr(weight), hr.ptr(bias), hr.ptr(om), hr.ptr(out), hr.ptr(cols),
                                           hr.ptr(stats), blk, nrows, float(_act_slope), *geom, hr.ptr(ws), ws.numel(),
                                           hr.stream()),
                 'dcn_v2_forward_om')
    return out, cols


def dcn_v2_backward_om(input, weight, bias, om, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                       dilation_h, dilation_w, _columns=None, _grad_weight=None, _grad_bias=None, _grad_input=None):
    """-> [grad_input, grad_om, grad_weight, grad_bias]; grad_om: the offsets' gradient in channels 0 .. 2T-1, the gradient of
    the mask's LOGIT in channels 2T .. 3T-1."""
    hr.require_gpu(input, weight, bias, om, grad_output)
    input, weight, bias, om, grad_output = [hr.f32c(t) for t in (input, weight, bias, om, grad_output)]
    B, C, H, W = input.shape
    Co = weight.shape[0]
    geom = (B, C, H, W, Co, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, 1)
    grads = [_grad_input if _grad_input is not None else torch.empty_like(input), torch.empty_like(om),
             _grad_weight if _grad_weight is not None else torch.empty_like(weight),
             _grad_bias if _grad_bias is not None else torch.empty_like(bias)]
    L = hr.lib()
    ws = hr.workspace(L.cnuda_dcn_v2_workspace_bytes(*geom), input.device)
    hr.prof_arm('dcn_bwd', B, C, H, W, Co, kernel_h, kernel_w, grad_output.shape[2], grad_output.shape[3])
    hr.check(L.cnuda_dcn_v2_backward_om(hr.ptr(input), hr.ptr(weight), hr.ptr(bias), hr.ptr(om), hr.ptr(grad_output),
                                        hr.ptr(_columns), hr.ptr(grads[0]), 1 if _grad_input is not None else 0,
                                        hr.ptr(grads[1]), hr.ptr(grads[2]), hr.ptr(grads[3]), *geom, hr.ptr(ws), ws.numel(),
                                        hr.stream()), 'dcn_v2_backward_om')
    return grads
