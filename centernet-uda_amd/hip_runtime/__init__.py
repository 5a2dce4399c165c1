"""ctypes binding of libcenternet_uda_hip.so (the C ABI in include/centernet_uda_hip.h).

This is the only place the product path touches native code.  There is no
fallback: if the shared library is missing or a call fails, a RuntimeError is
raised (the reference raises RuntimeError out of its pybind module too,
libs/DCNv2/src/dcn_v2.h:35).  torch is used for device memory and the current
HIP stream only.
"""
import ctypes
import os
import threading

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_ROOT, 'libcenternet_uda_hip.so')
ABI_VERSION = 2

_lib = None
_lock = threading.Lock()
_workspaces = {}

c_int, c_size_t, c_void_p, c_float = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float


def lib():
    """Load the HIP library (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "%s not found: build it with `make -C centernet-uda_amd/csrc` "
                        "(or __graft_entry__.build()); there is no CPU/eager fallback" % LIB_PATH)
                l = ctypes.CDLL(LIB_PATH)
                l.cnuda_last_error.restype = ctypes.c_char_p
                l.cnuda_abi_version.restype = c_int
                if l.cnuda_abi_version() != ABI_VERSION:
                    raise RuntimeError("libcenternet_uda_hip.so ABI %d != expected %d"
                                       % (l.cnuda_abi_version(), ABI_VERSION))
                for name in dir(_Sig):
                    if name.startswith('cnuda_'):
                        restype, argtypes = getattr(_Sig, name)
                        fn = getattr(l, name)
                        fn.restype, fn.argtypes = restype, argtypes
                _lib = l
    return _lib


_I = c_int
_P = c_void_p


_LL = ctypes.c_longlong
_F = c_float
_WS = [_P, c_size_t, _P]          # workspace, workspace_bytes, stream


class _Sig:
    """restype, argtypes for every exported symbol (mirrors include/centernet_uda_hip.h)."""
    cnuda_decode_workspace_bytes = (c_size_t, [_I] * 5)
    cnuda_decode_detection = (_I, [_P] * 5 + [_I] * 8 + _WS)
    cnuda_nms = (_I, [_P, _P] + [_I] * 5 + [_P])
    cnuda_dcn_v2_workspace_bytes = (c_size_t, [_I] * 14)
    cnuda_dcn_v2_forward = (_I, [_P] * 6 + [_I] * 14 + _WS)
    cnuda_dcn_v2_backward = (_I, [_P] * 11 + [_I] * 14 + _WS)
    cnuda_dcn_v2_forward_cols = (_I, [_P] * 7 + [_I] * 14 + _WS)
    cnuda_dcn_v2_backward_cols = (_I, [_P] * 12 + [_I] * 14 + _WS)
    cnuda_conv2d_workspace_bytes = (c_size_t, [_I] * 11)
    cnuda_conv2d_forward = (_I, [_P] * 4 + [_I] * 11 + [_F] + _WS)
    cnuda_conv2d_backward_data = (_I, [_P] * 3 + [_I] * 11 + _WS)
    cnuda_conv2d_backward_weight = (_I, [_P] * 4 + [_I] * 11 + _WS)
    cnuda_bn_workspace_bytes = (c_size_t, [_I, _I, _LL])
    cnuda_bn_train_forward = (_I, [_P] * 10 + [_F, _F, _I, _I, _I, _LL, _I] + _WS)
    cnuda_bn_eval_forward = (_I, [_P] * 7 + [_F, _I, _I, _I, _LL, _P])
    cnuda_bn_backward = (_I, [_P] * 10 + [_I, _I, _I, _LL, _I] + _WS)
    cnuda_maxpool2d_forward = (_I, [_P] * 2 + [_I] * 5 + [_P])
    cnuda_maxpool2d_backward = (_I, [_P] * 3 + [_I] * 5 + [_P])
    cnuda_maxpool2d_window_forward = (_I, [_P] * 2 + [_I] * 7 + [_P])
    cnuda_maxpool2d_window_backward = (_I, [_P] * 3 + [_I] * 7 + [_P])
    cnuda_dwconvt2d_forward = (_I, [_P] * 3 + [_I] * 7 + [_P])
    cnuda_dwconvt2d_workspace_bytes = (c_size_t, [_I] * 3)
    cnuda_dwconvt2d_backward = (_I, [_P] * 5 + [_I] * 7 + _WS)
    cnuda_dwconv2d_workspace_bytes = (c_size_t, [_I] * 3)
    cnuda_dwconv2d_forward = (_I, [_P] * 3 + [_I] * 7 + [_P])
    cnuda_dwconv2d_backward = (_I, [_P] * 5 + [_I] * 7 + _WS)
    cnuda_add = (_I, [_P] * 3 + [_LL, _P])
    cnuda_act_backward = (_I, [_P] * 3 + [_LL, _F, _P])
    cnuda_copy_channels = (_I, [_P, _P, _I, _I, _LL, _I, _I, _I, _I, _P])
    cnuda_split_offset_mask = (_I, [_P] * 3 + [_I, _I, _LL, _P])
    cnuda_split_offset_mask_backward = (_I, [_P] * 4 + [_I, _I, _LL, _P])
    cnuda_loss_workspace_bytes = (c_size_t, [])
    cnuda_focal_loss_forward = (_I, [_P] * 4 + [_LL, _F] + _WS)
    cnuda_focal_loss_backward = (_I, [_P] * 5 + [_LL, _F, _P])
    cnuda_reg_l1_forward = (_I, [_P] * 5 + [_I, _I, _I, _LL, _I, _F, _F, _P])
    cnuda_reg_l1_backward = (_I, [_P] * 7 + [_I, _I, _I, _LL, _I, _F, _F, _P])
    cnuda_kps_l1_forward = (_I, [_P] * 6 + [_I, _I, _I, _LL, _I, _I, _F, _F, _P])
    cnuda_kps_l1_backward = (_I, [_P] * 8 + [_I, _I, _I, _LL, _I, _I, _F, _F, _P])
    cnuda_decode_keypoints = (_I, [_P] * 4 + [_I] * 5 + [_P])
    cnuda_softmax_loss_forward = (_I, [_P, _P, _I, _I, _LL, _I] + _WS)
    cnuda_softmax_loss_backward = (_I, [_P] * 3 + [_I, _I, _LL, _I, _P])
    cnuda_entropy_map_forward = (_I, [_P, _P, _I, _I, _LL, _P])
    cnuda_entropy_map_backward = (_I, [_P] * 3 + [_I, _I, _LL, _P])
    cnuda_bce_const_forward = (_I, [_P, _F, _P, _LL, _P])
    cnuda_bce_const_backward = (_I, [_P, _F, _P, _P, _LL, _P])
    cnuda_sigmoid_clamp_ = (_I, [_P, _P, _LL, _P])
    cnuda_gather_feat = (_I, [_P] * 3 + [_I, _I, _I, _LL, _P])
    cnuda_encode_targets = (_I, [_P] * 10 + [_I] * 5 + [_P])
    cnuda_adam_step = (_I, [_P] * 4 + [_LL] + [_F] * 5 + [_I, _P])
    cnuda_set_matrix_mode = (_I, [_I])
    cnuda_get_matrix_mode = (_I, [])
    cnuda_prof_enable = (_I, [_I])
    cnuda_prof_arm = (_I, [_I])
    cnuda_prof_collect = (_I, [_P, _P, _I])


# ---------------------------------------------------------------------------
# statistics groups of a batch that carries several domains
# ---------------------------------------------------------------------------
_GROUPS = 1


class domain_groups:
    """with domain_groups(2): backend(torch.cat([source, target])) -- every train-mode BatchNorm inside the block
    normalises the two halves of the batch by their own statistics and updates its running statistics once per
    half, in order: exactly what two consecutive forward calls do (uda/entropy_minimization.py:18-19, Q6), in half
    the launches and with twice the pixels per GEMM."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        global _GROUPS
        self.prev, _GROUPS = _GROUPS, self.n

    def __exit__(self, *exc):
        global _GROUPS
        _GROUPS = self.prev


def current_groups():
    return _GROUPS


def check(rc, what=''):
    if rc != 0:
        msg = lib().cnuda_last_error().decode('utf-8', 'replace')
        raise RuntimeError(msg or ('%s failed with code %d' % (what, rc)))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("centernet-uda_amd ops run on MI355X only: got a %s tensor "
                               "(there is no CPU fallback; the CPU oracle lives in oracle/ for tests)"
                               % t.device)


def f32c(t):
    """contiguous fp32 view/copy -- the native side reads raw NCHW fp32 (dcn_v2_cuda.cu:58,219-220)."""
    if t.dtype != torch.float32:
        raise RuntimeError("expected float32, got %s" % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def workspace(nbytes, device):
    """Grow-only scratch buffer per device.  All kernels of one step run on one
    stream, so stream order makes reuse between consecutive calls safe."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream().cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


# ---------------------------------------------------------------------------
# kernel timer used by bench.py (hipEvents recorded inside the library around the
# main implicit-GEMM launch of a call, on the launch stream)
# ---------------------------------------------------------------------------
class _Prof:
    enabled = False
    table = []          # tag -> [(kernel name, algorithmic FLOPs, algorithmic HBM bytes) per sub-kernel]
    cap = 0


def set_matrix_mode(mode):
    """0: f32 MFMA (default).  1: exact three-way bf16 split of every f32 operand, six partial products on the
    bf16 MFMA, f32 accumulation (include/centernet_uda_hip.h, cnuda_set_matrix_mode).  Process-wide."""
    check(lib().cnuda_set_matrix_mode(int(mode)), 'set_matrix_mode')


def get_matrix_mode():
    return int(lib().cnuda_get_matrix_mode())


def _bm(m, n):
    """mirror of pick_bm() in csrc/conv.hip / dcn.hip (tile rows for M output rows and N pixels)."""
    bm = 128 if m > 64 else (64 if m > 32 else 32)
    n_tiles = (n + 127) // 128
    while bm > 32 and n_tiles * ((m + bm - 1) // bm) < 512:
        bm >>= 1
    return bm


def _ws():
    """mirror of wave_specialised() in csrc/runtime.hip"""
    return os.environ.get('CNUDA_WS', '1')[:1] != '0'


def _fwd_name(bm, loader):
    """kernel template instance launch_fwd() picks (csrc/conv.hip): the 8-wave producer / consumer variant for the
    64- and 128-row tiles in matrix mode 0, the 4-wave kernel otherwise"""
    if get_matrix_mode() == 1:
        return 'igemm_fwd_kernel<%d, %s> [split bf16 x3]' % (bm, loader)
    return ('igemm_fwd_ws_kernel<%d, %s>' if (_ws() and bm >= 64) else 'igemm_fwd_kernel<%d, %s>') % (bm, loader)


def _smallc(C, Co, kh, kw, stride):
    return stride == 1 and C <= 16 and Co <= 32 and C * kh * kw <= 148 and C * (3 + kh) <= 96


def prof_arm(kind, B, C, H, W, Co, kh, kw, Ho, Wo):
    """Called by the conv / DCN ops right before the C-ABI call when profiling is on.  The kernel name is
    the template instance the library will pick (same selection rules), so that bench.py's per-kernel
    aggregation lines up with rocprofv3's kernel names."""
    if not _Prof.enabled or len(_Prof.table) >= _Prof.cap:
        return
    flops = 2.0 * B * Ho * Wo * Co * C * kh * kw          # 2*Cout*Cin*kh*kw*Ho*Wo per image
    stride = max(1, round(H / max(Ho, 1)))
    tf = lambda v: 'true' if v else 'false'
    if kind == 'conv_fwd':
        if _smallc(C, Co, kh, kw, stride):
            name = 'smallc_fwd_kernel<%d>' % ((Co + 15) // 16)
        else:
            name = _fwd_name(_bm(Co, B * Ho * Wo), 'ConvFwdLoader<%s>' % tf(C % 16 == 0))
    elif kind == 'conv_dgrad':
        if stride == 1 and _smallc(Co, C, kh, kw, 1):
            name = 'smallc_fwd_kernel<%d>' % ((C + 15) // 16)
        elif stride > 1 and H % stride == 0 and W % stride == 0 and Co % 16 == 0 and \
                (-(-kh // stride)) * (-(-kw // stride)) <= 9:
            name = 'igemm_fwd*_kernel<*, ConvDgradClassLoader> (group of stride^2 class launches)'
        else:
            name = _fwd_name(_bm(C, B * H * W), 'ConvDgradLoader')
    elif kind == 'conv_wgrad':
        if _smallc(C, Co, kh, kw, stride):
            name = 'smallc_wgrad_kernel<%d>' % ((Co + 15) // 16)
        else:
            wide = Co <= 32 or (C % 64 == 0 and (C * kh * kw) % 128 == 0)
            ws = _ws() and C % 64 == 0 and Co > 32
            name = 'igemm_wgrad%s_kernel<ConvWLoader<%d>, %d, %d>' % ('_ws' if ws else '', 2 if C % 64 == 0 else 0,
                                                                      32 if Co <= 32 else 64, 128 if wide else 64)
    elif kind == 'dcn_fwd':
        bm = _bm(Co, B * Ho * Wo)
        name = ('dcn_sample_kernel + igemm_fwd_kernel<%d, DcnColsLoader>' if Co > bm else
                'igemm_fwd_kernel<%d, DcnFwdLoader>') % bm
    elif kind == 'dcn_bwd':
        # three kernels, timed separately (ProfGroup in csrc/dcn.hip): the column-gradient GEMM (a 1x1
        # convolution over grad_output with 9*C output channels: all of the entry point's MFMA work), then
        # the two HBM-streaming consumers with their algorithmic bytes
        T, px = kh * kw, B * Ho * Wo
        name = _fwd_name(_bm(T * C, px), 'ConvFwdLoader<%s>' % tf(Co % 16 == 0))
        coord_bytes = 4.0 * px * (T * C + 3 * T + 3 * T + 4 * T) + 4.0 * B * C * H * W
        col2im_bytes = 4.0 * px * T * C + 16.0 * px * T * ((C + 15) // 16) + 4.0 * B * C * H * W
        _Prof.table.append([(name, flops, 0.0), ('dcn_coord_grad_kernel', 0.0, coord_bytes),
                            ('dcn_col2im_kernel', 0.0, col2im_bytes)])
        lib().cnuda_prof_arm(len(_Prof.table) - 1)
        return
    else:
        raise ValueError(kind)
    _Prof.table.append([(name, flops, 0.0)])
    lib().cnuda_prof_arm(len(_Prof.table) - 1)


def prof_begin(max_records=16384):
    check(lib().cnuda_prof_enable(int(max_records)), 'prof_enable')
    _Prof.enabled, _Prof.table, _Prof.cap = True, [], int(max_records)


def prof_end():
    """-> {kernel name: {'launches', 'ms', 'flops', 'bytes'}} and disables the timer."""
    n = sum(len(e) for e in _Prof.table)          # one record per (armed call, sub-kernel)
    tags = (ctypes.c_int * max(n, 1))()
    ms = (ctypes.c_float * max(n, 1))()
    got = lib().cnuda_prof_collect(tags, ms, n)
    out = {}
    for i in range(got):
        name, flops, nbytes = _Prof.table[tags[i] & 0xffffff][tags[i] >> 24]
        d = out.setdefault(name, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
        d['launches'] += 1
        d['ms'] += float(ms[i])
        d['flops'] += flops
        d['bytes'] += nbytes
    _Prof.enabled, _Prof.table = False, []
    lib().cnuda_prof_enable(0)
    return out
