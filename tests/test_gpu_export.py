"""MI355X: the inference wrapper (export.py:19-56) -- eval forward, clamped sigmoid, decode, down_ratio scaling,
output split -- against the oracle's decode applied to the same head tensors."""
import ast

import numpy as np
import pytest
import torch

import inputs as gin
from oracle import decode as oracle_decode

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize('rotated', [False, True])
def test_centernet_wrapper_matches_oracle_decode(golden, rotated):
    from backends import dla
    from export import CenterNet
    g = golden('dla_rot' if rotated else 'dla_axis')
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    backend = dla.build(num_classes=6, rotated_boxes=rotated)
    backend.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes).items()})
    backend = backend.to(DEV).eval()
    x = T(gin.image_batch(2, 96, 96, 91)).to(DEV)
    K = 20
    model = CenterNet(backend, K, is_rotated=rotated).eval()          # BatchNorm-folded inference path
    boxes, scores, classes = model(x)
    assert boxes.shape == (2, K, 5 if rotated else 4) and scores.shape == (2, K) and classes.shape == (2, K)
    assert not boxes.requires_grad
    with model._folding(), torch.no_grad():                # the same (BatchNorm-folded) head tensors the wrapper decoded
        out = backend(x)
    hm = np.clip(1.0 / (1.0 + np.exp(-out['hm'].double().cpu().numpy())), 1e-4, 1 - 1e-4).astype(np.float32)
    want = oracle_decode.decode_detection(hm, out['wh'].cpu().numpy(), out['reg'].cpu().numpy(), K=K, rotated=rotated)
    want[:, :, :4] *= backend.down_ratio
    nb = 5 if rotated else 4
    # scores come from float32 sigmoids on both sides: equal to 1 ulp; a swap of two detections would show up as
    # a class / box mismatch far above the tolerance
    np.testing.assert_allclose(scores.cpu().numpy(), want[:, :, nb], rtol=0, atol=2e-7)
    np.testing.assert_array_equal(classes.cpu().numpy(), want[:, :, nb + 1])
    np.testing.assert_allclose(boxes.cpu().numpy(), want[:, :, :nb], rtol=1e-6, atol=1e-4)


def test_centernet_wrapper_with_keypoint_head():
    """A backend with num_keypoints > 0 adds a 'kps' head (backends/resnet.py build(); export.py:29-54): the wrapper
    returns a fourth output, the decoded keypoints scaled by down_ratio."""
    from backends import resnet
    from export import CenterNet
    torch.manual_seed(7)
    backend = resnet.build(18, num_classes=3, num_keypoints=5, pretrained=False).to(DEV).eval()
    x = torch.randn(2, 3, 64, 64, device=DEV)
    K = 12
    boxes, scores, classes, kps = CenterNet(backend, K)(x)
    assert kps.shape == (2, K, 5, 2) and boxes.shape == (2, K, 4)
    with torch.no_grad():
        out = backend(x)
    assert out['kps'].shape == (2, 10, 16, 16)
    # a random-init backend scores every cell ~0.505: neighbouring candidates are 1 ulp apart, so the oracle gets the
    # very float32 heat map the wrapper decoded (the sigmoid kernel has its own test) -- the ranking is then bit-exact
    from hip_runtime import ops
    hm = ops.sigmoid_clamp_(out['hm'].clone()).cpu().numpy()
    want, want_kps = oracle_decode.decode_detection(hm, out['wh'].cpu().numpy(), out['reg'].cpu().numpy(), K=K,
                                                    kps=out['kps'].cpu().numpy())
    np.testing.assert_array_equal(classes.cpu().numpy(), want[:, :, 5])
    np.testing.assert_allclose(kps.cpu().numpy(), want_kps * backend.down_ratio, rtol=1e-6, atol=1e-5)


def test_build_model_reads_the_experiment_folder(tmp_path):
    from export import CenterNet, build_model
    from utils.helper import save_model
    from backends import resnet
    src = resnet.build(18, num_classes=3, pretrained=False)
    with torch.no_grad():
        src.hm[2].bias.fill_(-1.5)
    save_model(src, tmp_path / 'model_last.pth', epoch=4)
    spec = {'name': 'resnet', 'params': {'num_layers': 18, 'num_classes': 3, 'pretrained': False, 'rotated_boxes': False}}
    model = build_model(tmp_path, spec, without_decode_detections=False, max_detections=10)
    assert isinstance(model, CenterNet) and model.max_detections == 10 and model.is_rotated is False
    assert torch.all(model.backend.hm[2].bias == -1.5)
    boxes, scores, classes = model.to(DEV).eval()(torch.randn(1, 3, 64, 64, device=DEV))
    assert boxes.shape == (1, 10, 4) and torch.all(scores[:, :-1] >= scores[:, 1:])
    backend = build_model(tmp_path / 'missing', spec, without_decode_detections=True, max_detections=10)
    assert isinstance(backend, resnet.CenterResNet)


def test_batchnorm_folding_matches_the_batchnorm_kernels_and_tracks_the_weights(golden):
    """eval-mode inference with every BatchNorm folded into the convolution in front of it
    (conv + bias + skip connection + ReLU in one GEMM epilogue; DCN + BN + ReLU likewise) equals the unfolded
    eval forward to rounding, never records a tape, and follows the weights when they change."""
    from backends import dla
    from export import CenterNet
    from hip_runtime import optim
    g = golden('dla_axis')
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    backend = dla.build(num_classes=6)
    backend.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, 0.3).items()})
    backend = backend.to(DEV).eval()
    x = T(gin.image_batch(2, 128, 128, 92)).to(DEV)
    with torch.no_grad():
        want = backend(x)                                  # eval-mode BatchNorm kernels
    assert all(m._fold is None for m in backend.modules() if hasattr(m, 'fold_batchnorm_'))
    model = CenterNet(backend, 30).eval()
    with model._folding():
        with torch.no_grad():
            got = backend(x)
    n_folded = sum(1 for m in backend.modules() if getattr(m, '_fold', None) is not None)
    assert n_folded >= 40                                 # 16 DeformConv + the trunk's conv/BN blocks
    for k in want:
        scale = want[k].abs().max().item()
        assert (got[k] - want[k]).abs().max().item() <= 1e-4 * scale, k          # measured 2e-5
        assert not got[k].requires_grad
    # outside the wrapper's context the blocks run their BatchNorm kernels again (e.g. plugin evaluation)
    with torch.no_grad():
        again = backend(x)
    assert all(torch.equal(again[k], want[k]) for k in want)
    boxes0, scores0, _ = model(x)
    # the weights change: a training step through the fused Adam (torch's version counters do not see it)
    backend.train()
    opt = optim.Adam([p for p in backend.parameters() if p.requires_grad], lr=1e-2)
    out = backend(x)
    sum(v.mean() for v in out.values()).backward()
    opt.step()
    backend.eval()
    with torch.no_grad():
        want2 = backend(x)
    assert (want2['hm'] - want['hm']).abs().max().item() > 1e-3      # the step really moved the model
    with model._folding():
        with torch.no_grad():
            got2 = backend(x)
    for k in want2:
        assert (got2[k] - want2[k]).abs().max().item() <= 1e-4 * want2[k].abs().max().item(), k
    # ... and through load_state_dict (torch's in-place copy)
    backend.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, 0.3).items()})
    boxes1, scores1, _ = model(x)
    assert torch.equal(scores1, scores0) and torch.equal(boxes1, boxes0)
    # fold = False: the wrapper runs the BatchNorm kernels
    model.fold = False
    boxes2, scores2, _ = model(x)
    assert (scores2 - scores0).abs().max().item() <= 1e-4


def test_folded_weights_follow_running_statistics_re_estimated_without_an_optimizer_step(golden):
    """AdaBN-style use: train-mode forward passes over target images (running statistics move, no parameter does),
    then export.  The statistics kernel writes running_mean / running_var through raw pointers, so torch's version
    counters and the parameter epoch never change -- the folded copies must still be rebuilt
    (hip_runtime.bump_buffer_epoch)."""
    from backends import dla
    from export import CenterNet
    g = golden('dla_axis')
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    backend = dla.build(num_classes=6)
    backend.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, 0.3).items()})
    backend = backend.to(DEV).eval()
    x = T(gin.image_batch(2, 128, 128, 92)).to(DEV)
    model = CenterNet(backend, 30).eval()
    with model._folding():
        with torch.no_grad():
            before = {k: v.clone() for k, v in backend(x).items()}
    backend.train()
    with torch.no_grad():
        for seed in (93, 94, 95):
            backend(T(gin.image_batch(2, 128, 128, seed)).to(DEV) * 1.5 + 0.2)
    backend.eval()
    with torch.no_grad():
        want = backend(x)                                  # eval-mode BatchNorm kernels on the new statistics
    assert (want['hm'] - before['hm']).abs().max().item() > 1e-3
    with model._folding():
        with torch.no_grad():
            got = backend(x)
    for k in want:
        assert (got[k] - want[k]).abs().max().item() <= 1e-4 * want[k].abs().max().item(), k


def test_steady_state_inference_packs_no_weight(golden):
    """Every folded weight has its own identity in the library's pack cache: after the first call of the wrapper no
    cached weight image is written again (the two 3x3 convolutions of a BasicBlock have one shape; under one token they
    took over each other's slot on every call: 20 pack launches per forward)."""
    import hip_runtime as hr
    from backends import dla
    from export import CenterNet
    g = golden('dla_axis')
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    backend = dla.build(num_classes=6)
    backend.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, 0.3).items()})
    backend = backend.to(DEV).eval()
    x = T(gin.image_batch(2, 128, 128, 96)).to(DEV)
    model = CenterNet(backend, 30).eval()
    first = model(x)
    model(x)
    fills, used = hr.lib().cnuda_pack_cache_fills(), hr.lib().cnuda_pack_cache_used()
    again = model(x)
    torch.cuda.synchronize()
    assert hr.lib().cnuda_pack_cache_fills() == fills and hr.lib().cnuda_pack_cache_used() == used
    assert all(torch.equal(a, b) for a, b in zip(first, again))
