"""No-GPU checks of the C-ABI library: it loads, exports every symbol the header
declares, reports its ABI version, answers workspace queries, and the Python
shim refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, 'include', 'centernet_uda_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cnuda_\w+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import hip_runtime as hr
    assert os.path.exists(hr.LIB_PATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(hr.LIB_PATH)
    names = _declared()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), n
    # and the ctypes signature table covers all of them
    for n in names:
        assert hasattr(hr._Sig, n) or n in ('cnuda_abi_version', 'cnuda_last_error'), n


def test_abi_version_and_workspace_queries():
    import hip_runtime as hr
    L = hr.lib()
    assert L.cnuda_abi_version() == hr.ABI_VERSION == 2
    assert L.cnuda_decode_workspace_bytes(16, 6, 128, 128, 150) >= 16 * 6 * 150 * 8
    assert L.cnuda_dcn_v2_workspace_bytes(16, 64, 128, 128, 64, 3, 3, 1, 1, 1, 1, 1, 1, 1) > 0
    assert L.cnuda_dcn_v2_workspace_bytes(16, 63, 128, 128, 64, 3, 3, 1, 1, 1, 1, 1, 1, 2) == 0   # C % dg != 0
    assert b'deformable_group' in L.cnuda_last_error()
    assert L.cnuda_conv2d_workspace_bytes(16, 64, 128, 128, 64, 3, 3, 1, 1, 1, 1) > 0
    assert L.cnuda_bn_workspace_bytes(16, 64, 16384) > 0
    assert L.cnuda_loss_workspace_bytes() > 0


def test_invalid_arguments_return_error_codes_without_a_gpu():
    import hip_runtime as hr
    L = hr.lib()
    # null pointers / bad geometry are rejected before anything touches the device
    assert L.cnuda_nms(None, None, 1, 1, 4, 4, 3, None) == -1
    assert L.cnuda_decode_detection(None, None, None, None, None, 1, 1, 4, 4, 2, 2, 0, 3, None, 0, None) == -1
    assert L.cnuda_adam_step(None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, None) == -1


def test_no_cpu_fallback_anywhere():
    from backends.decode import decode_detection
    from hip_runtime import ops
    import _ext
    z = torch.zeros
    with pytest.raises(RuntimeError, match='MI355X only'):
        decode_detection(z(1, 1, 4, 4), z(1, 2, 4, 4), K=2)
    with pytest.raises(RuntimeError, match='MI355X only'):
        ops.conv2d(z(1, 3, 8, 8), z(4, 3, 3, 3), None, 1, 1)
    with pytest.raises(RuntimeError, match='MI355X only'):
        _ext.dcn_v2_forward(z(1, 4, 4, 4), z(2, 4, 3, 3), z(2), z(1, 18, 4, 4), z(1, 9, 4, 4), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match='MI355X only'):
        ops.entropy_loss(z(1, 3, 4, 4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'centernet-uda_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cuh')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, os.path.join(dirpath, f)


def test_plugin_signatures_match_reference():
    import inspect
    import uda
    from losses.centernet import DetectionLoss
    from backends.decode import decode_detection
    from libs.DCNv2.dcn_v2 import DCN, DCNv2, dcn_v2_conv  # noqa: F401
    assert list(inspect.signature(DetectionLoss.__init__).parameters)[1:] == [
        'hm_weight', 'wh_weight', 'off_weight', 'kp_weight', 'angle_weight', 'periodic', 'kp_indices',
        'kp_distance_weight', 'kp_distance_weight_l1']
    assert list(inspect.signature(decode_detection).parameters) == ['heat', 'wh', 'reg', 'kps', 'K', 'rotated',
                                                                     'nms_size']
    assert list(inspect.signature(uda.EntropyMinimization.__init__).parameters)[1:] == ['entropy_weight']
    assert list(inspect.signature(uda.MaxSquaresMinimization.__init__).parameters)[1:] == ['max_squares_weight']
    assert list(inspect.signature(uda.AdversarialEntropyMinimization.__init__).parameters)[1:] == [
        'adversarial_weight', 'optimizer']
    for name in ('init_done', 'epoch_start', 'epoch_end', 'step', 'set_phase', 'to', 'criterion', 'get_detections',
                 'load_model', 'save_model'):
        assert callable(getattr(uda.Model, name))
    assert list(inspect.signature(DCN.__init__).parameters)[1:] == [
        'in_channels', 'out_channels', 'kernel_size', 'stride', 'padding', 'dilation', 'deformable_groups']


def test_uda_package_resolves_plugins_lazily_and_names_what_is_missing():
    import importlib
    import uda
    for name in ('Model', 'EntropyMinimization', 'MaxSquaresMinimization', 'AdversarialEntropyMinimization'):
        cls = getattr(uda, name)                      # what hydra's `uda.<ClassName>` lookup does (train.py:104-106)
        assert cls.__name__ == name and name in dir(uda)
        assert issubclass(cls, uda.Model)
    assert uda.MaxSquaresMinimization is importlib.import_module('uda.max_squares_minimization').MaxSquaresMinimization
    with pytest.raises(AttributeError, match='rfft'):
        uda.FDA
    with pytest.raises(AttributeError):
        uda.NoSuchPlugin


def test_gather_feat_helper_matches_the_reference_formula():
    """utils/tensor.py:10-18 restated with take_along_dim: same rows, same masked flattening."""
    import torch
    from utils.tensor import _gather_feat
    g = torch.Generator().manual_seed(3)
    feat = torch.randn(3, 20, 5, generator=g)
    ind = torch.randint(0, 20, (3, 7), generator=g)
    mask = torch.rand(3, 7, generator=g) > 0.4
    dim = feat.size(2)
    want = feat.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), dim))
    assert torch.equal(_gather_feat(feat, ind), want)
    assert torch.equal(_gather_feat(feat, ind, mask), want[mask.unsqueeze(2).expand_as(want)].view(-1, dim))
    assert torch.equal(_gather_feat(feat, ind, mask.to(torch.uint8)), _gather_feat(feat, ind, mask))
