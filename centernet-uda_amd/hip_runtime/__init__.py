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
    cnuda_decode_set_max_bands = (_I, [_I])
    cnuda_decode_set_stage2_threads = (_I, [_I])
    cnuda_decode_detection = (_I, [_P] * 5 + [_I] * 8 + _WS)
    cnuda_nms = (_I, [_P, _P] + [_I] * 5 + [_P])
    cnuda_dcn_v2_workspace_bytes = (c_size_t, [_I] * 14)
    cnuda_dcn_v2_forward = (_I, [_P] * 6 + [_I] * 14 + _WS)
    cnuda_dcn_v2_backward = (_I, [_P] * 11 + [_I] * 14 + _WS)
    cnuda_dcn_v2_forward_cols = (_I, [_P] * 7 + [_I] * 14 + _WS)
    cnuda_dcn_v2_forward_act = (_I, [_P] * 7 + [_F] + [_I] * 14 + _WS)
    cnuda_dcn_v2_forward_stats = (_I, [_P] * 8 + [_I] * 16 + _WS)
    cnuda_dcn_v2_stats_block = (_I, [_I] * 14 + [_P])
    cnuda_dcn_v2_forward_om = (_I, [_P] * 7 + [_I] * 2 + [_F] + [_I] * 14 + _WS)
    cnuda_dcn_v2_backward_om = (_I, [_P] * 7 + [_I] + [_P] * 3 + [_I] * 14 + _WS)
    cnuda_dcn_v2_backward_cols = (_I, [_P] * 12 + [_I] * 14 + _WS)
    cnuda_dcn_v2_backward_acc = (_I, [_P] * 8 + [_I] + [_P] * 4 + [_I] * 14 + _WS)
    cnuda_conv2d_workspace_bytes = (c_size_t, [_I] * 11)
    cnuda_conv2d_forward = (_I, [_P] * 4 + [_I] * 11 + [_F] + _WS)
    cnuda_conv2d_forward_res = (_I, [_P] * 5 + [_I] * 11 + [_F] + _WS)
    cnuda_conv2d_rowsig_supported = (_I, [_I] * 11)
    cnuda_conv2d_forward_rowsig = (_I, [_P] * 4 + [_I] * 12 + _WS)
    cnuda_conv2d_rowquads_supported = (_I, [_I] * 11)
    cnuda_conv2d_cat_supported = (_I, [_P, _I, _I, _I, _I, _I])
    cnuda_conv2d_cat_forward = (_I, [_P, _P, _I, _P, _P, _P, _P, _F, _I, _I, _I, _I] + _WS)
    cnuda_conv2d_cat_backward_data = (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I] + _WS)
    cnuda_conv2d_cat_backward_weight = (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I] + _WS)
    cnuda_conv2d_forward_rowquads = (_I, [_P] * 3 + [_I] * 11 + _WS)
    cnuda_conv2d_backward_data = (_I, [_P] * 3 + [_I] * 11 + _WS)
    cnuda_conv2d_backward_data_add = (_I, [_P] * 5 + [_I] * 11 + _WS)
    cnuda_conv2d_stats_block = (_I, [_I] * 11 + [_P, _P])
    cnuda_conv2d_forward_stats = (_I, [_P] * 6 + [_I] * 11 + [_F] + _WS)
    cnuda_bn_train_forward_stats = (_I, [_P, _P, _LL, _I] + [_P] * 9 + [_F, _F, _I, _I, _I, _LL, _I] + _WS)
    cnuda_conv2d_backward_weight = (_I, [_P] * 4 + [_I] * 11 + _WS)
    cnuda_conv2d_norm_input_supported = (_I, [_I] * 11)
    cnuda_conv2d_forward_norm_input = (_I, [_P] * 5 + [_I] + [_P] * 4 + [_I] * 11 + [_F] + _WS)
    cnuda_conv2d_backward_weight_norm_input = (_I, [_P] * 5 + [_I] + [_P] * 3 + [_I] * 11 + _WS)
    cnuda_bn_workspace_bytes = (c_size_t, [_I, _I, _LL])
    cnuda_bn_train_forward = (_I, [_P] * 10 + [_F, _F, _I, _I, _I, _LL, _I] + _WS)
    cnuda_bn_eval_forward = (_I, [_P] * 7 + [_F, _I, _I, _I, _LL, _P])
    cnuda_bn_backward = (_I, [_P] * 11 + [_I, _I, _I, _LL, _I] + _WS)
    cnuda_maxpool2d_forward = (_I, [_P] * 2 + [_I] * 5 + [_P])
    cnuda_maxpool2d_backward = (_I, [_P] * 3 + [_I] * 5 + [_P])
    cnuda_maxpool2d_backward_acc = (_I, [_P] * 3 + [_I] * 6 + [_P])
    cnuda_maxpool2d_window_forward = (_I, [_P] * 2 + [_I] * 7 + [_P])
    cnuda_maxpool2d_window_backward = (_I, [_P] * 3 + [_I] * 7 + [_P])
    cnuda_dwconvt2d_forward = (_I, [_P] * 3 + [_I] * 7 + [_P])
    cnuda_dwconvt2d_add_forward = (_I, [_P] * 4 + [_I] * 7 + [_P])
    cnuda_dwconvt2d_workspace_bytes = (c_size_t, [_I] * 3)
    cnuda_dwconvt2d_backward = (_I, [_P] * 5 + [_I] * 7 + _WS)
    cnuda_dwconv2d_workspace_bytes = (c_size_t, [_I] * 3)
    cnuda_dwconv2d_forward = (_I, [_P] * 3 + [_I] * 7 + [_P])
    cnuda_dwconv2d_backward = (_I, [_P] * 5 + [_I] * 7 + _WS)
    cnuda_add = (_I, [_P] * 3 + [_LL, _P])
    cnuda_act_backward = (_I, [_P] * 3 + [_LL, _F, _P])
    cnuda_conv1x1_backward_data_act = (_I, [_P] * 4 + [_I] * 3 + [_LL, _F, _P])
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
    cnuda_pack_cache_attach = (_I, [_P, c_size_t])
    cnuda_pack_stamp = (_I, [ctypes.c_ulonglong, ctypes.c_ulonglong])
    cnuda_pack_refresh = (_I, [_P, ctypes.c_size_t, ctypes.c_ulonglong, ctypes.c_ulonglong, _P, ctypes.c_size_t, _P])
    cnuda_pack_cache_used = (c_size_t, [])
    cnuda_pack_cache_fills = (ctypes.c_ulonglong, [])
    cnuda_pack_cache_resets = (ctypes.c_ulonglong, [])
    cnuda_prof_enable = (_I, [_I])
    cnuda_prof_arm = (_I, [_I])
    cnuda_prof_collect = (_I, [_P, _P, _P, _I])
    cnuda_prof_name_len = (_I, [])
    cnuda_launch_log_enable = (_I, [_I])
    cnuda_launch_log_collect = (_I, [ctypes.c_char_p, c_size_t])
    cnuda_dcn_set_fused_min_tiles = (_I, [_I])
    cnuda_dcn_set_scatter_margin = (_I, [_I])
    cnuda_dcn_set_offset_regime = (_I, [_I])
    cnuda_dcn_set_walk_tile = (_I, [_I])
    cnuda_dcn_offset_census = (_I, [_P, _I, _I, _LL, _P, _P])
    cnuda_conv_set_halo_policy = (_I, [_I, _I])
    cnuda_conv_set_splitk_policy = (_I, [_I])


# ---------------------------------------------------------------------------
# Parameter epoch: bumped by everything in this package that rewrites parameter memory behind torch's back (the
# fused Adam kernel, the data-parallel broadcast).  Together with the tensors' own `_version` counters (which torch
# bumps for its in-place ops: stock optimizers, load_state_dict) it tells derived copies of the weights -- the
# BatchNorm-folded inference weights -- when they are stale.
# ---------------------------------------------------------------------------
_PARAM_EPOCH = 0


def bump_param_epoch(rewritten=None):
    """`rewritten`: the flat tensor whose contents just changed (the fused Adam's parameter arena).  Every packed weight
    image of the pack cache that was built from it in the epoch that ends here is rebuilt by one launch and carried
    into the new epoch (cnuda_pack_refresh) -- instead of ~150 small pack launches spread over the next step."""
    global _PARAM_EPOCH
    old = _PARAM_EPOCH
    _PARAM_EPOCH += 1
    if rewritten is not None and rewritten.is_cuda and _PACK['arena'] is not None and not _PACK['off']:
        if _PACK.get('table') is None or _PACK['table'].device != rewritten.device:
            _PACK['table'] = torch.empty(1 << 18, dtype=torch.uint8, device=rewritten.device)
        check(lib().cnuda_pack_refresh(ptr(rewritten), rewritten.numel() * rewritten.element_size(), old, _PARAM_EPOCH,
                                       ptr(_PACK['table']), _PACK['table'].numel(), stream()), 'pack_refresh')


# Buffer epoch: bumped by every train-mode BatchNorm call (the statistics kernel rewrites running_mean / running_var /
# num_batches_tracked through raw pointers: torch's version counters never see it).  Kept apart from _PARAM_EPOCH on
# purpose: the packed-weight cache is stamped with the parameter epoch and must not be invalidated by a BN call.
_BUFFER_EPOCH = 0


def bump_buffer_epoch():
    global _BUFFER_EPOCH
    _BUFFER_EPOCH += 1


def param_state_key(module):
    """Changes whenever a parameter or buffer of `module` may have changed: the parameter epoch (fused Adam, parallel
    broadcast), the buffer epoch (train-mode BatchNorm forward: AdaBN-style re-estimation of the running statistics
    with no optimizer step in between), and per tensor its identity, storage address and torch version counter
    (stock optimizers, load_state_dict, `.to()`)."""
    per_tensor = tuple((id(t), t.data_ptr(), t._version)
                       for t in list(module.parameters()) + list(module.buffers()))
    return (_PARAM_EPOCH, _BUFFER_EPOCH, hash(per_tensor))


# ---------------------------------------------------------------------------
# Pack cache (csrc/pack.hip): the packed weight image of a convolution only changes when the weights do.  Modules
# that own weights take a token (new_pack_token) and stamp their calls with (token, version of the weight tensor);
# the library then re-packs only after an optimizer step / load_state_dict.  The arena is one buffer allocated
# here on first use (CNUDA_PACK_CACHE_MB, default 768; 0 disables).  Functional calls without a token never cache.
# Loophole, as for torch's own saved-tensor checks: in-place writes through `weight.data` bump no version.
# ---------------------------------------------------------------------------
_PACK = {'arena': None, 'next_token': 1, 'off': os.environ.get('CNUDA_PACK_CACHE_MB', '768') == '0'}


def new_pack_token():
    t = _PACK['next_token']
    _PACK['next_token'] += 1
    return t


class PackToken:
    """A module's identity in the pack cache, as a module attribute.  A DEEP copy of the module (an EMA / mean-teacher
    copy, swa_utils.AveragedModel, copy.deepcopy(model)) or an unpickled module owns different weight buffers, so it
    takes a fresh identity: with a shared token two live models would take over each other's cached image on every
    call (each call re-packs; results stay right, the cache is defeated).  A shallow copy shares the weights and keeps
    the token."""
    __slots__ = ('id',)

    def __init__(self):
        self.id = new_pack_token()

    def __int__(self):
        return self.id

    __index__ = __int__

    def __deepcopy__(self, memo):
        return PackToken()

    def __reduce__(self):
        return (PackToken, ())

    def __repr__(self):
        return 'PackToken(%d)' % self.id


class pack_stamp:
    """with pack_stamp(token, weight[, version]): <one C-ABI call that packs `weight`>"""
    __slots__ = ('token', 'version', 'device')

    def __init__(self, token, weight, version=None):
        self.token = 0 if _PACK['off'] else int(token)
        v = weight._version if version is None else int(version)
        self.version = ((_PARAM_EPOCH & 0xffffffff) << 32) | (v & 0xffffffff)
        self.device = weight.device

    def __enter__(self):
        if self.token:
            if _PACK['arena'] is None:
                mb = int(os.environ.get('CNUDA_PACK_CACHE_MB', '768'))
                _PACK['arena'] = torch.empty(mb << 20, dtype=torch.uint8, device=self.device)
                check(lib().cnuda_pack_cache_attach(ptr(_PACK['arena']), _PACK['arena'].numel()), 'pack_cache_attach')
            lib().cnuda_pack_stamp(self.token, self.version)

    def __exit__(self, *exc):
        if self.token:
            lib().cnuda_pack_stamp(0, 0)


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


# torch.cuda.current_stream() builds a Stream object through three Python layers (~4 us; two of them per operator call:
# a third of the host time of a decode call, 2.6 of the 6.6 ms of the launch-bound ResNet-18 step): the raw handle of
# the current device's current stream comes straight from the C layer
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_cur_device = getattr(torch._C, '_cuda_getDevice', None)


def _stream_handle():
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def stream():
    return c_void_p(_stream_handle())


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
    if getattr(t, '_cnuda_deferred_bn', None) is not None:
        # the never-written output of a BatchNorm in deferred mode (ops.batch_norm_act(defer_apply=True)): a 4-byte
        # placeholder that only ops.conv2d's apply-on-load kernels may consume -- anything else would read garbage
        raise RuntimeError("this tensor is the unwritten output of a deferred BatchNorm (apply on load): only ops.conv2d "
                           "may consume it; build the BatchNorm without defer_apply for other consumers")
    return t if t.is_contiguous() else t.contiguous()


def workspace(nbytes, device):
    """Grow-only scratch buffer per device.  All kernels of one step run on one
    stream, so stream order makes reuse between consecutive calls safe."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream_handle())
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
    table = []          # tag -> [(algorithmic FLOPs, algorithmic HBM bytes) per timed sub-kernel of the call]
    shapes = []         # tag -> (kind, B, C, H, W, Co, kh, kw, Ho, Wo)
    cap = 0


def set_matrix_mode(mode):
    """0: f32 MFMA (default).  1: exact three-way bf16 split of every f32 operand, six partial products on the
    bf16 MFMA, f32 accumulation (include/centernet_uda_hip.h, cnuda_set_matrix_mode).  Process-wide."""
    check(lib().cnuda_set_matrix_mode(int(mode)), 'set_matrix_mode')


def get_matrix_mode():
    return int(lib().cnuda_get_matrix_mode())


def prof_arm(kind, B, C, H, W, Co, kh, kw, Ho, Wo):
    """Called by the conv / DCN ops right before the C-ABI call when profiling is on: records the call's algorithmic
    work (FLOPs per SURVEY 8d, bytes for the HBM-streaming kernels) under a tag; the kernel NAME of every timed
    launch comes back from the library itself (cnuda_prof_collect), which is the only place that knows which
    template instance it selected."""
    if not _Prof.enabled or len(_Prof.table) >= _Prof.cap:
        return
    flops = 2.0 * B * Ho * Wo * Co * C * kh * kw          # 2*Cout*Cin*kh*kw*Ho*Wo per image
    if kind == 'dcn_bwd':
        # three timed kernels (ProfGroup in csrc/dcn.hip): sub 0 the column-gradient GEMM (a 1x1 convolution over
        # grad_output with 9*C output channels: all of the entry point's MFMA work), then the two HBM-streaming
        # consumers with their algorithmic bytes
        T, px = kh * kw, B * Ho * Wo
        coord_bytes = 4.0 * px * (T * C + 3 * T + 3 * T + 4 * T) + 4.0 * B * C * H * W
        col2im_bytes = 4.0 * px * T * C + 16.0 * px * T * ((C + 15) // 16) + 4.0 * B * C * H * W
        # (sub 3: both consumers as one launch, dcn_bwd_data_kernel: dcol and the geometry records once each way,
        # the input, grad_offset / grad_mask out, grad_input out)
        fused_bytes = 4.0 * px * T * C + 16.0 * px * T + 12.0 * px * T + 8.0 * B * C * H * W
        _Prof.table.append([(flops, 0.0), (0.0, coord_bytes), (0.0, col2im_bytes), (0.0, fused_bytes), (flops, 0.0)])   # 4: weight gradient
    elif kind in ('conv_fwd', 'conv_dgrad', 'conv_wgrad', 'dcn_fwd'):
        # algorithmic bytes beside the FLOPs: the two activation tensors once each (the weights are noise) -- the layers with
        # 2..32 output channels at full resolution are bound by these, not by the matrix pipe; the DCN forward also reads
        # offsets / mask and, in training, writes its sampled columns
        nbytes = 4.0 * B * (C * H * W + Co * Ho * Wo)
        if kind == 'dcn_fwd':
            nbytes += 4.0 * B * Ho * Wo * (3 * kh * kw + kh * kw * C)
        _Prof.table.append([(flops, nbytes)])
    else:
        raise ValueError(kind)
    _Prof.shapes.append((kind, B, C, H, W, Co, kh, kw, Ho, Wo))
    lib().cnuda_prof_arm(len(_Prof.table) - 1)


def prof_begin(max_records=16384):
    check(lib().cnuda_prof_enable(int(max_records)), 'prof_enable')
    _Prof.enabled, _Prof.table, _Prof.shapes, _Prof.cap = True, [], [], int(max_records)


def prof_end(by_shape=False):
    """-> {kernel name: {'launches', 'ms', 'flops', 'bytes'}} and disables the timer.  by_shape: the keys are
    (kernel name, (kind, B, C, H, W, Co, kh, kw, Ho, Wo)) -- one row per layer shape."""
    n = sum(len(e) for e in _Prof.table)          # one record per (armed call, sub-kernel)
    L = lib()
    nl = int(L.cnuda_prof_name_len())
    tags = (ctypes.c_int * max(n, 1))()
    ms = (ctypes.c_float * max(n, 1))()
    names = ctypes.create_string_buffer(max(n, 1) * nl)
    got = L.cnuda_prof_collect(tags, ms, names, n)
    out = {}
    for i in range(got):
        flops, nbytes = _Prof.table[tags[i] & 0xffffff][tags[i] >> 24]
        name = names.raw[i * nl:(i + 1) * nl].split(b'\0', 1)[0].decode() or 'unnamed launch'
        key = (name, _Prof.shapes[tags[i] & 0xffffff]) if by_shape else name
        d = out.setdefault(key, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0})
        d['launches'] += 1
        d['ms'] += float(ms[i])
        d['flops'] += flops
        d['bytes'] += nbytes
    _Prof.enabled, _Prof.table, _Prof.shapes = False, [], []
    lib().cnuda_prof_enable(0)
    return out


# ---------------------------------------------------------------------------
# test aids (include/centernet_uda_hip.h, "Test aids")
# ---------------------------------------------------------------------------
def launch_counts():
    """-> {kernel symbol name: launches since the log was enabled}; enables the log on first use (it stays on: one
    predictable branch per launch)."""
    L = lib()
    L.cnuda_launch_log_enable(1)
    buf = ctypes.create_string_buffer(1 << 18)
    n = L.cnuda_launch_log_collect(buf, len(buf))
    if n < 0:
        check(n, 'launch_log_collect')
    out = {}
    for line in buf.value.decode().split('\n'):
        if line:
            name, _, count = line.rpartition('\t')
            out[name] = int(count)
    return out


class launch_log:
    """with launch_log() as log: ...   ->  log.names: the library's kernels launched inside the block, by their own
    symbol names (what a rocprofv3 kernel trace shows); log.counts: launches of each.  Blocks may nest."""

    def __enter__(self):
        self.before = launch_counts()
        self.names, self.counts = [], {}
        return self

    def __exit__(self, *exc):
        after = launch_counts()
        self.counts = {k: v - self.before.get(k, 0) for k, v in after.items() if v > self.before.get(k, 0)}
        self.names = list(self.counts)


class dcn_fused_min_tiles:
    """with dcn_fused_min_tiles(1): every DCN backward inside the block that CAN take the one-launch data-gradient
    walk (dcn_bwd_data_kernel) takes it, whatever its size; with dcn_fused_min_tiles(2 ** 31 - 1): none does."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        self.prev = lib().cnuda_dcn_set_fused_min_tiles(self.n)

    def __exit__(self, *exc):
        lib().cnuda_dcn_set_fused_min_tiles(self.prev)


class splitk:
    """with splitk(max_tiles): forward-type convolution GEMMs with fewer than `max_tiles` pixel x row tiles (and a long K)
    inside the block cut K over the grid (cnuda_conv_set_splitk_policy); 0 = never."""

    def __init__(self, max_tiles):
        self.max_tiles = int(max_tiles)

    def __enter__(self):
        self.prev = lib().cnuda_conv_set_splitk_policy(self.max_tiles)

    def __exit__(self, *exc):
        lib().cnuda_conv_set_splitk_policy(self.prev)


class halo_conv:
    """with halo_conv(level, min_tiles): which 3x3 / stride-1 convolutions inside the block take the halo-tile kernels
    (cnuda_conv_set_halo_policy: level 0 none, 1 every eligible layer, 2 the 32-row GEMMs; calls with at least
    `min_tiles` pixel tiles).  Tests use halo_conv(1, 1)."""

    def __init__(self, level, min_tiles):
        self.level, self.min_tiles = int(level), int(min_tiles)

    def __enter__(self):
        prev = lib().cnuda_conv_set_halo_policy(-1, 0)
        self.prev = (prev & 0xff, prev >> 8)
        lib().cnuda_conv_set_halo_policy(self.level, self.min_tiles)

    def __exit__(self, *exc):
        lib().cnuda_conv_set_halo_policy(*self.prev)
