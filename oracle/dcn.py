"""TEST INFRASTRUCTURE -- CPU oracle for `_ext.dcn_v2_forward/backward`.

Restates the per-sample sequence of the reference's CPU extension
(libs/DCNv2/src/cpu/dcn_v2_cpu.cpp:22-111 forward, :113-229 backward): bias
broadcast, im2col, `W_flat @ columns`; and for the gradients `W_flat^T @ gout`,
coordinate/mask gradients, col2im scatter, im2col again, `gout @ columns^T`,
`gout @ ones`.  The scalar loops live in oracle/dcn_ref.inc (plain C, built by
`make -C oracle`); the GEMMs are CPU torch.matmul exactly where the reference
calls at::matmul.

Parity pin: the reference's native sources need <TH/TH.h>, which this image
does not ship, so they are unbuildable here (no oracle/_ref).  This oracle is
pinned by the reference's own known-answer tests instead (libs/DCNv2/testcpu.py
:32-67 zero-offset identity, :69-97 gradcheck tolerances), by an fp64
finite-difference check of this same code (dtype=torch.float64) and by
cross-checks against plain convolution for zero / integer offsets
(tests/test_oracle_dcn.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile the C oracle (gcc) if the shared object is missing or stale."""
    so = os.path.join(_HERE, "libdcn_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("dcn_ref.c", "dcn_ref.inc")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _geom(input, weight, stride, pad, dil):
    B, C, H, W = input.shape
    Co, Ck, kh, kw = weight.shape
    if Ck != C:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (C, Ck))
    Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    return B, C, H, W, Co, kh, kw, Ho, Wo


def _fn(name, dtype):
    prefix = {torch.float32: "dcnf_", torch.float64: "dcnd_"}[dtype]
    return getattr(_lib(), prefix + name)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _ints(*v):
    return [ctypes.c_int(int(x)) for x in v]


def dcn_v2_forward(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg):
    """Same argument order as the native `_ext.dcn_v2_forward` (src/dcn_v2.h:10-23)."""
    input, weight, bias, offset, mask = [t.contiguous() for t in (input, weight, bias, offset, mask)]
    if tuple(weight.shape[2:]) != (kh, kw):
        raise RuntimeError("Input shape and kernel shape wont match")
    B, C, H, W, Co, kh, kw, Ho, Wo = _geom(input, weight, (sh, sw), (ph, pw), (dh, dw))
    dt = input.dtype
    out = torch.empty(B, Co, Ho, Wo, dtype=dt)
    col = torch.empty(C * kh * kw, Ho * Wo, dtype=dt)
    wflat = weight.view(Co, C * kh * kw)
    im2col = _fn("im2col", dt)
    g = _ints(C, H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg)
    for b in range(B):
        im2col(_p(input[b]), _p(offset[b]), _p(mask[b]), _p(col), *g)
        out[b] = (bias.view(Co, 1) + torch.matmul(wflat, col)).view(Co, Ho, Wo)
    return out


def dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kh, kw, sh, sw, ph, pw, dh, dw, dg):
    """Returns [grad_input, grad_offset, grad_mask, grad_weight, grad_bias]
    (the reference's order, dcn_v2_cpu.cpp:226-228)."""
    input, weight, offset, mask, grad_output = [
        t.contiguous() for t in (input, weight, offset, mask, grad_output)]
    B, C, H, W, Co, kh, kw, Ho, Wo = _geom(input, weight, (sh, sw), (ph, pw), (dh, dw))
    dt = input.dtype
    gin = torch.zeros_like(input)
    goff = torch.zeros_like(offset)
    gmask = torch.zeros_like(mask)
    gw = torch.zeros_like(weight)
    gb = torch.zeros_like(bias)
    col = torch.empty(C * kh * kw, Ho * Wo, dtype=dt)
    wflat = weight.view(Co, C * kh * kw)
    f_im2col, f_col2im, f_coord = _fn("im2col", dt), _fn("col2im", dt), _fn("col2im_coord", dt)
    g = _ints(C, H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, dg)
    for b in range(B):
        go = grad_output[b].reshape(Co, Ho * Wo)
        dcol = torch.matmul(wflat.t(), go).contiguous()
        f_coord(_p(dcol), _p(input[b]), _p(offset[b]), _p(mask[b]), _p(goff[b]), _p(gmask[b]), *g)
        f_col2im(_p(dcol), _p(offset[b]), _p(mask[b]), _p(gin[b]), *g)
        f_im2col(_p(input[b]), _p(offset[b]), _p(mask[b]), _p(col), *g)
        gw += torch.matmul(go, col.t()).view_as(gw)
        gb += go.sum(dim=1)
    return [gin, goff, gmask, gw, gb]


class _DCNv2Oracle(torch.autograd.Function):
    """autograd wrapper with the Python-side argument order of
    libs/DCNv2/dcn_v2.py:18-19 (input, offset, mask, weight, bias, ...)."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, dg):
        ctx.geom = (weight.shape[2], weight.shape[3], stride, stride, padding, padding, dilation, dilation, dg)
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return dcn_v2_forward(input.detach(), weight.detach(), bias.detach(), offset.detach(),
                              mask.detach(), *ctx.geom)

    @staticmethod
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        gi, go, gm, gw, gb = dcn_v2_backward(input, weight, bias, offset, mask, grad_output, *ctx.geom)
        return gi, go, gm, gw, gb, None, None, None, None


def dcn_v2_conv(input, offset, mask, weight, bias, stride=1, padding=1, dilation=1, deformable_groups=1):
    return _DCNv2Oracle.apply(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups)
