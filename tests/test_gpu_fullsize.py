"""Steps at BASELINE.json's FULL size (DLA-34 + DCNv2, 512 x 512, 16 source [+ 16 target] images) on the MI355X,
checked through size-independent properties -- the CPU oracle needs about a minute per image at this size:

* configs[1] vs configs[2]: `EntropyMinimization(entropy_weight = 0).step` adds a zero gradient to the detection
  gradient, so parameters after Adam and the detection statistics must equal `uda.base.Model.step`'s
  (uda/base.py:31-56 vs uda/entropy_minimization.py:11-43); BatchNorm's running statistics differ by the second
  momentum update (Q6), which is checked as such;
* a step is a function of (parameters, batch): two runs from the same state agree (the only order-dependent
  arithmetic left is the float atomics of col2im's stragglers);
* loss values of the full-size batch are finite and equal the loss kernels evaluated on the step's own head
  outputs by the CPU oracle (the loss is cheap on the CPU even at this size);
* configs[3] (max-squares, 512 x 512) and configs[4] (rotated + periodic + ADVENT, 640 x 640) the same way, the
  latter with the discriminator, its three adversarial statistics and its whole gradient re-evaluated on the CPU.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _run(uda_name, weight=None, steps=1, seed=42, size=512, before_step=None):
    import bench
    plugin = bench.build_plugin(DEV, parallel=False, uda_name=uda_name)
    if weight is not None:
        plugin.entropy_weight = weight
    batch = bench.synthetic_batch(16, size, seed, DEV, rotated=bench.UDA_WORKLOADS[uda_name][2])
    if before_step is not None:
        before_step(plugin)
    outs = [plugin.step(dict(batch)) for _ in range(steps)]
    torch.cuda.synchronize()
    model = plugin.backend
    params = {n: p.detach().clone() for n, p in model.named_parameters()}
    bufs = {n: b.detach().clone() for n, b in model.named_buffers()}
    return plugin, batch, outs, params, bufs


def test_full_size_step_properties():
    from oracle import losses as ol
    _, batch, out_b, p_base, b_base = _run('none')
    _, _, out_e, p_ent0, b_ent0 = _run('entropy', weight=0.0)
    sb, se = out_b[0]['stats'], out_e[0]['stats']
    for k in ('centernet_loss', 'hm_loss', 'wh_loss', 'off_loss'):
        assert np.isfinite(float(sb[k]))
        assert abs(float(sb[k]) - float(se[k])) <= 1e-6 * abs(float(sb[k])), k
    assert float(se['entropy_loss']) == 0.0 and abs(float(se['total_loss']) - float(sb['total_loss'])) <= 1e-6 * abs(float(sb['total_loss']))
    # same Adam step (+-lr * g / (|g| + eps)) from identical detection gradients.  Elements whose gradient is
    # rounding noise -- the DCN biases in front of a BatchNorm have an analytically zero gradient -- may take the
    # other sign when col2im's straggler atomics land in another order: at most 2 * lr apart, and rare elsewhere.
    # (Round 5: the BatchNorm statistics come from the producing GEMM's epilogue, whose pixel blocks follow the launch's
    # tile shape -- 16 images and 16 + 16 images pick different tiles on the small maps, so the two runs' statistics
    # differ in their last bit and a few more noise-level gradients change sign: measured 0.12 % of the elements.)
    lr, moved, total = 5e-5, 0, 0
    for n in p_base:
        d = (p_base[n] - p_ent0[n]).abs()
        assert d.max().item() <= 2.1 * lr, (n, d.max().item())
        if n.endswith('.conv.bias'):
            continue
        moved += int((d > 1e-7).sum())
        total += d.numel()
    assert moved <= 3e-3 * total, (moved, total)
    for n in b_base:
        if n.endswith('num_batches_tracked'):
            assert int(b_ent0[n]) == 2 * int(b_base[n]) == 2, n                       # Q6
    # the loss of the full-size batch, re-evaluated by the CPU oracle on the step's own head outputs
    src = {k: v.detach().float().cpu() for k, v in out_b[0]['source_domain'].items()}
    logits = torch.log(src['hm'] / (1 - src['hm']))                                   # the dict holds probabilities (Q1)
    cb = {k: v.cpu() for k, v in batch.items()}
    loss, stats, _ = ol.detection_loss(dict(src, hm=logits), cb, 1.0, 0.1, 1.0, 1.0, False)
    for k, v in stats.items():
        assert abs(float(v) - float(sb[k])) <= 2e-4 * max(abs(float(v)), 1e-3), (k, float(v), float(sb[k]))


def test_full_size_uda_step_is_reproducible():
    """Forward passes are deterministic (fixed-order reductions everywhere); the gradients are too up to the float
    atomics of col2im's stragglers.  Compared after ONE step: Adam turns a sign change of a rounding-noise
    gradient into a +-lr move, so later steps of two runs legitimately drift apart by ~1e-4."""
    pl1, _, o1, _, b1 = _run('entropy')
    g1 = {n: p.grad.detach().clone() for n, p in pl1.backend.named_parameters() if p.grad is not None}
    pl2, _, o2, _, b2 = _run('entropy')
    g2 = {n: p.grad.detach().clone() for n, p in pl2.backend.named_parameters() if p.grad is not None}
    for k in o1[0]['stats']:
        assert float(o1[0]['stats'][k]) == float(o2[0]['stats'][k]), k
    for dom in ('source_domain', 'target_domain'):
        for k in ('hm', 'wh', 'reg'):
            assert torch.equal(o1[0][dom][k], o2[0][dom][k]), (dom, k)
    assert sorted(g1) == sorted(g2)
    for n in g1:
        scale = g2[n].abs().max().item()
        if n.endswith('.conv.bias') and 'ida' in n:
            continue                      # DCN bias in front of a BatchNorm: the gradient is rounding noise only
        assert (g1[n] - g2[n]).abs().max().item() <= 1e-4 * max(scale, 1e-12), n
    for n in b1:
        if b1[n].is_floating_point():
            assert torch.equal(b1[n], b2[n]), n


def _grads(plugin, module=None):
    m = plugin.backend if module is None else module
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}


def _same_step_twice(run):
    """`run() -> (plugin, outs)`, twice from the same state: statistics and head outputs bit-equal, gradients equal up to
    the float atomics of col2im's stragglers (1e-4 of the tensor's maximum; DCN biases in front of a BatchNorm are
    rounding noise only)."""
    pl1, o1 = run()
    g1 = _grads(pl1)
    pl2, o2 = run()
    g2 = _grads(pl2)
    for k in o1['stats']:
        assert float(o1['stats'][k]) == float(o2['stats'][k]), k
    for dom in ('source_domain', 'target_domain'):
        for k in o1[dom]:
            assert torch.equal(o1[dom][k], o2[dom][k]), (dom, k)
    assert sorted(g1) == sorted(g2)
    for n in g1:
        if n.endswith('.conv.bias') and 'ida' in n:
            continue
        assert (g1[n] - g2[n]).abs().max().item() <= 1e-4 * max(g2[n].abs().max().item(), 1e-12), n
    return pl1, o1, pl2, o2


def test_full_size_max_squares_step_properties():
    """configs[3] (uda/max_squares_minimization.py:11-50) at 512 x 512, 16 + 16 images: the detection statistics and the
    weighted max-squares term re-evaluated by the CPU oracle on the step's OWN head outputs (2e-4), weight 0 = the
    configs[1] step (same detection gradient -> same parameters after Adam), two runs agree."""
    from oracle import losses as ol
    _, batch, out_b, p_base, _ = _run('none')

    def zero_weight(plugin):
        plugin.max_squares_weight = 0.0
    _, _, out_0, p_0, b_0 = _run('maxsq', before_step=zero_weight)
    sb, s0 = out_b[0]['stats'], out_0[0]['stats']
    for k in ('centernet_loss', 'hm_loss', 'wh_loss', 'off_loss'):
        assert abs(float(sb[k]) - float(s0[k])) <= 1e-6 * abs(float(sb[k])), k
    assert float(s0['max_square_loss']) == 0.0
    lr, moved, total = 5e-5, 0, 0
    for n in p_base:
        d = (p_base[n] - p_0[n]).abs()
        assert d.max().item() <= 2.1 * lr, (n, d.max().item())
        if not n.endswith('.conv.bias'):
            moved += int((d > 1e-7).sum())
            total += d.numel()
    assert moved <= 3e-3 * total, (moved, total)
    assert all(int(v) == 2 for n, v in b_0.items() if n.endswith('num_batches_tracked'))           # Q6

    def run():
        pl, _, outs, _, _ = _run('maxsq')
        return pl, outs[0]
    pl, o, _, _ = _same_step_twice(run)
    st = o['stats']
    assert list(st) == ['centernet_loss', 'hm_loss', 'wh_loss', 'off_loss', 'max_square_loss', 'total_loss']
    src = {k: v.detach().float().cpu() for k, v in o['source_domain'].items()}
    logits = torch.log(src['hm'] / (1 - src['hm']))                                    # the dict holds probabilities (Q1)
    cb = {k: v.cpu() for k, v in batch.items()}
    _, want, _ = ol.detection_loss(dict(src, hm=logits), cb, 1.0, 0.1, 1.0, 1.0, False)
    for k, v in want.items():
        assert abs(float(v) - float(st[k])) <= 2e-4 * max(abs(float(v)), 1e-3), (k, float(v), float(st[k]))
    msq = 0.3 * float(ol.max_square_loss(o['target_domain']['hm'].detach().float().cpu()))   # logged WEIGHTED (Q4)
    assert abs(msq - float(st['max_square_loss'])) <= 2e-4 * abs(msq), (msq, float(st['max_square_loss']))
    assert abs(float(st['total_loss']) - (float(st['centernet_loss']) + float(st['max_square_loss']))) <= 1e-5 * abs(float(st['total_loss']))


def test_full_size_advent_step_properties():
    """configs[4] (uda/adversarial_entropy_minimization.py:77-152) at 640 x 640, 16 + 16 images, rotated boxes, periodic
    angle loss, the 5-layer conv4x4-s2 discriminator on 160 x 160 maps.  On the step's OWN head outputs the CPU oracle
    re-evaluates: the detection statistics (rotated + periodic, 2e-4); entropy_map + discriminator + BCE for the three
    adversarial statistics and `source_generator` with the discriminator's weights from BEFORE the step; the
    discriminator's whole gradient (its two backward calls) element by element.  Two runs agree."""
    import math
    import torch.nn.functional as F
    from oracle import losses as ol
    d_before = {}

    def grab(plugin):
        d_before.update({k: v.detach().clone().cpu() for k, v in plugin.discriminator.state_dict().items()})

    def run():
        pl, batch, outs, _, _ = _run('advent', size=640, before_step=grab)
        run.batch = batch
        return pl, outs[0]
    pl, o, _, _ = _same_step_twice(run)
    st = o['stats']
    assert list(st) == ['centernet_loss', 'hm_loss', 'wh_loss', 'off_loss', 'total_loss', 'dis_soruce', 'dis_target',
                        'dis_fool']
    assert all(math.isfinite(float(v)) for v in st.values()), st
    assert o['source_domain']['hm'].shape == (16, 6, 160, 160) and o['source_domain']['wh'].shape == (16, 3, 160, 160)
    src = {k: v.detach().float().cpu() for k, v in o['source_domain'].items()}
    tgt_logits = o['target_domain']['hm'].detach().float().cpu()
    logits = torch.log(src['hm'] / (1 - src['hm']))
    cb = {k: v.cpu() for k, v in run.batch.items()}
    _, want, _ = ol.detection_loss(dict(src, hm=logits), cb, 1.0, 0.1, 1.0, 1.0, True)
    for k, v in want.items():
        assert abs(float(v) - float(st[k])) <= 2e-4 * max(abs(float(v)), 1e-3), (k, float(v), float(st[k]))
    # the discriminator on the CPU (uda/adversarial_entropy_minimization.py:51-68), weights as they were before the step;
    # in float64 (the value) and in float32 (how far a float32 evaluation of the SAME graph lands from it: near its
    # initialisation the discriminator's weight gradients are sums of +- terms that cancel to ~1e-3 of their parts)
    def cpu_discriminator(dtype):
        w = {k: v.clone().to(dtype).requires_grad_(True) for k, v in d_before.items()}

        def D(x):
            for i in (0, 2, 4, 6):
                x = F.leaky_relu(F.conv2d(x, w['%d.weight' % i], w['%d.bias' % i], 2, 1), 0.2)
            return F.conv2d(x, w['8.weight'], w['8.bias'], 2, 1)
        s_logits = D(ol.entropy_map(src['hm'].to(dtype)))        # softmax over PROBABILITIES (Q1)
        t_logits = D(ol.entropy_map(tgt_logits.to(dtype)))
        ds, dt = ol.advent_loss(s_logits, 0) / 2, ol.advent_loss(t_logits, 1) / 2
        fool = 1e-3 * ol.advent_loss(t_logits, 0)
        (ds + dt).backward()
        return s_logits.detach(), {'dis_soruce': ds.item(), 'dis_target': dt.item(), 'dis_fool': fool.item()}, \
            {k: v.grad.double() for k, v in w.items()}
    s_logits, want_d, g64 = cpu_discriminator(torch.float64)
    _, _, g32 = cpu_discriminator(torch.float32)
    assert s_logits.shape == (16, 1, 5, 5)
    for k, v in want_d.items():
        assert abs(v - float(st[k])) <= 2e-4 * abs(v), (k, v, float(st[k]))
    tot = float(st['centernet_loss']) + float(st['dis_soruce']) + float(st['dis_target']) + float(st['dis_fool'])
    assert abs(tot - float(st['total_loss'])) <= 1e-5 * abs(tot)
    got_sg = o['source_generator'].detach().double().cpu()
    assert (got_sg - s_logits).abs().max().item() <= 1e-4 * max(1.0, s_logits.abs().max().item())
    dg = _grads(pl, pl.discriminator)
    assert sorted(dg) == sorted(g64)
    for n, g in dg.items():
        ref = g64[n]
        scale, err = ref.abs().max().item(), (g.double().cpu() - ref).abs().max().item()
        noise = (g32[n] - ref).abs().max().item()
        print('discriminator %-10s %8d elements  err/max %.2e  float32-on-CPU noise/max %.2e'
              % (n, ref.numel(), err / max(scale, 1e-30), noise / max(scale, 1e-30)))
        # element by element: 1e-4 of the tensor's maximum, or 8 x the distance of the CPU's own float32 evaluation
        assert err <= max(1e-4 * scale, 8 * noise), (n, err, scale, noise)
    assert all(p.requires_grad for p in pl.discriminator.parameters())


@pytest.mark.parametrize('B,C,S,Co', [(32, 256, 128, 6), (16, 256, 128, 2), (32, 64, 128, 576), (32, 16, 128, 256),
                                      (32, 256, 128, 64)])
def test_full_size_1x1_convolutions_match_fp64_and_repeat(B, C, S, Co):
    """The 128-row tiles, the wave-specialised kernels, the buffer-addressed loaders and the 16-byte buffer-store
    epilogue only run at sizes the per-operator tests (small tensors, seconds on the CPU oracle) never reach: 1x1
    convolutions of the heads / the DCN column-gradient GEMM at 128 x 128, B = 32, forward, input gradient and weight
    gradient against an fp64 contraction on the GPU, twice, bit-identical (a dropped store shows as an O(1) error
    in a handful of elements: max-norm, not mean)."""
    from hip_runtime import ops
    g = torch.Generator(device=DEV).manual_seed(B * 1000 + C + Co)
    x = torch.randn(B, C, S, S, device=DEV, generator=g)
    w = torch.randn(Co, C, 1, 1, device=DEV, generator=g) * 0.1
    gy = torch.randn(B, Co, S, S, device=DEV, generator=g)
    runs = []
    for _ in range(2):
        xx, ww = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = ops.conv2d(xx, ww, None, 1, 0)
        y.backward(gy)
        runs.append((y.detach(), xx.grad, ww.grad))
    assert all(torch.equal(a, b) for a, b in zip(*runs))
    y, gx, gw = runs[0]
    w2 = w.double()[:, :, 0, 0]
    for name, got, want in (('y', y, torch.einsum('bchw,oc->bohw', x.double(), w2)),
                            ('gx', gx, torch.einsum('bohw,oc->bchw', gy.double(), w2)),
                            ('gw', gw[:, :, 0, 0], torch.einsum('bohw,bchw->oc', gy.double(), x.double()))):
        scale = want.abs().max().item()
        assert (got.double() - want).abs().max().item() <= 1e-4 * scale, name


@pytest.mark.parametrize('C,Co,S', [(64, 128, 128), (128, 256, 64)], ids=['64to128_128sq', '128to256_64sq'])
def test_full_size_strided_convolution_matches_fp64(C, Co, S):
    """3x3 stride 2, 64 -> 128 from 128 x 128 to 64 x 64 and 128 -> 256 from 64 x 64 to 32 x 32, B = 32 (level3 / level4
    .tree1.tree1.conv1 of the benched step: the 64- and the 128-row tile of the input gradient): the
    input gradient runs its four input-pixel parity classes as ONE launch on the wave-specialised 64-row tile
    (`igemm_fwd_ws_classes_kernel<64>`: blockIdx.y = class; rounds 2-5: one launch per class), against the CPU's fp64 convolution; forward and weight gradient ride along."""
    import torch.nn.functional as F
    import hip_runtime as hr
    from hip_runtime import ops
    from test_zz_kernel_coverage import short
    g = torch.Generator().manual_seed(78)
    x = torch.randn(32, C, S, S, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.05
    gy = torch.randn(32, Co, S // 2, S // 2, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    want_y = F.conv2d(xr, wr, None, 2, 1)
    want_y.backward(gy.double())
    xx, ww = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    with hr.launch_log() as log:
        y = ops.conv2d(xx, ww, None, 2, 1)
        y.backward(gy.to(DEV))
    names = sorted(short(n) for n in log.names)
    import os
    if not any(os.environ.get(v) == '0' for v in ('CNUDA_BUF', 'CNUDA_WS')):
        assert 'igemm_fwd_ws_classes_kernel<%d>' % C in names, names       # (round 6: the four parity classes in one launch)
    for name, got, want in (('y', y.detach(), want_y.detach()), ('gx', xx.grad, xr.grad), ('gw', ww.grad, wr.grad)):
        scale = want.abs().max().item()
        assert (got.double().cpu() - want).abs().max().item() <= 1e-4 * scale, name


@pytest.mark.parametrize('B,C,Co,S', [(16, 64, 64, 128), (16, 64, 27, 128), (64, 128, 27, 64), (32, 64, 256, 128),
                                      (12, 64, 27, 160), (40, 128, 27, 80)],      # (round 6: configs[4]'s 160- / 80-wide maps)
                         ids=['64to64_128sq', '64to27_128sq', '128to27_64sq', '64to256_128sq', '64to27_160sq', '128to27_80sq'])
def test_full_size_halo_tile_convolutions_match_fp64(B, C, Co, S):
    """3x3 / stride 1 layers of the benched step on the halo-tile kernels (csrc/hconv.cuh) at sizes where the launch
    plan picks the 256-pixel tiles (64 x 256 and 32 x 256: two or four image rows per tile, >= 1024 tiles) and the
    128 x 128 tile: forward and input gradient against the CPU's fp64 convolution.  The benched step runs the 27-row
    cases (DCN offset convolutions) on them; the 64- / 128-row variants are selectable (CNUDA_HCONV=1) and measured
    slower than the wave-specialised im2col kernels inside the step (csrc/conv.hip hconv_level)."""
    import os
    import torch.nn.functional as F
    import hip_runtime as hr
    from hip_runtime import ops
    from test_zz_kernel_coverage import short
    g = torch.Generator().manual_seed(79 + Co)
    x = torch.randn(B, C, S, S, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) * 0.05
    gy = torch.randn(B, Co, S, S, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    want_y = F.conv2d(xr, wr, None, 1, 1)
    want_y.backward(gy.double())
    xx, ww = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    # (the default policy gives the halo-tile kernels the 32-row GEMMs only: level 1 = every eligible layer)
    with hr.halo_conv(1, 128), hr.launch_log() as log:
        y = ops.conv2d(xx, ww, None, 1, 1)
        y.backward(gy.to(DEV))
    names = sorted(short(n) for n in log.names)
    if hr.get_matrix_mode() == 0:
        want_fwd = 'hconv_kernel<%s, HconvFwd>' % {64: '64, 256', 27: '32, 256' if S % 32 == 0 else '32, 128', 256: '128, 128'}[Co]
        assert want_fwd in names, (want_fwd, names)
        if Co % 16 == 0:
            assert 'hconv_kernel<64, 256, HconvDgrad>' in names, names
    for name, got, want in (('y', y.detach(), want_y.detach()), ('gx', xx.grad, xr.grad), ('gw', ww.grad, wr.grad)):
        scale = want.abs().max().item()
        assert (got.double().cpu() - want).abs().max().item() <= 1e-4 * scale, name


def test_full_size_3x3_convolution_matches_fp64():
    """3x3, 64 -> 256 at 64 x 64, B = 8: the 128-row forward tile (2 row tiles x 256 pixel tiles), the 64-row input
    gradient and the 128 x 64 weight-gradient tile, against the CPU's fp64 convolution."""
    import torch.nn.functional as F
    from hip_runtime import ops
    g = torch.Generator().manual_seed(77)
    x = torch.randn(8, 64, 64, 64, generator=g)
    w = torch.randn(256, 64, 3, 3, generator=g) * 0.05
    gy = torch.randn(8, 256, 64, 64, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xr, wr, None, 1, 1).backward(gy.double())
    want_y = F.conv2d(x.double(), w.double(), None, 1, 1)
    xx, ww = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = ops.conv2d(xx, ww, None, 1, 1)
    y.backward(gy.to(DEV))
    for name, got, want in (('y', y.detach(), want_y), ('gx', xx.grad, xr.grad), ('gw', ww.grad, wr.grad)):
        scale = want.abs().max().item()
        assert (got.double().cpu() - want).abs().max().item() <= 1e-4 * scale, name


def test_kernels_are_unaffected_by_work_on_another_stream():
    """One process per GPU runs its gradient all-reduce (RCCL kernels) next to the backward pass, and nothing stops a
    caller from driving two models on two streams: a kernel's result must not depend on what else is resident on the
    CU.  (A 16-byte epilogue that stored through buffer descriptors passed every single-stream test and dropped a few
    stores per 100k as soon as a second convolution ran beside it.)  A full-size 1x1 convolution and a DCN layer
    are repeated while another stream runs convolution forward / backward passes; every result must be bit-identical
    to the one computed alone (grad_input of the DCN up to the order of col2im's straggler atomics)."""
    from hip_runtime import ops
    from libs.DCNv2.dcn_v2 import DCN
    torch.manual_seed(11)
    x = torch.randn(32, 128, 64, 64, device=DEV)
    w = torch.randn(1152, 128, 1, 1, device=DEV) * 0.1
    m = DCN(64, 64, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1).to(DEV)
    with torch.no_grad():
        m.conv_offset_mask.weight.normal_(0, 0.05)
        m.conv_offset_mask.bias.normal_(0, 0.3)
    xd, gd = torch.randn(32, 64, 128, 128, device=DEV), torch.randn(32, 64, 128, 128, device=DEV)

    def dcn_once():
        xi = xd.clone().requires_grad_(True)
        for p in m.parameters():
            p.grad = None
        y = m(xi)
        y.backward(gd)
        return y.detach().clone(), xi.grad.clone(), m.weight.grad.clone(), m.conv_offset_mask.weight.grad.clone()
    ref_y = ops.conv2d_infer(x, w, None, 1, 0).clone()
    ref_d = dcn_once()
    x3 = torch.randn(32, 128, 64, 64, device=DEV, requires_grad=True)
    w3 = (torch.randn(128, 128, 3, 3, device=DEV) * 0.05).requires_grad_(True)
    g3 = torch.randn(32, 128, 64, 64, device=DEV)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for it in range(6):
        with torch.cuda.stream(side):
            for _ in range(3):
                y3 = ops.conv2d(x3, w3, None, 1, 1)
                if it % 2:
                    y3.backward(g3)
        y = ops.conv2d_infer(x, w, None, 1, 0)
        d = dcn_once()
        torch.cuda.synchronize()
        assert torch.equal(y, ref_y), it
        assert torch.equal(d[0], ref_d[0]) and torch.equal(d[2], ref_d[2]) and torch.equal(d[3], ref_d[3]), it
        assert (d[1] - ref_d[1]).abs().max().item() <= 1e-5 * ref_d[1].abs().max().item(), it


def _dcn_full_case(B, C, Co, S, off_scale, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, S, S, generator=g)
    w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(Co, generator=g)
    off = torch.randn(B, 18, S, S, generator=g) * off_scale
    m = torch.sigmoid(torch.randn(B, 9, S, S, generator=g))
    go = torch.randn(B, Co, S, S, generator=g)
    return x, w, b, off, m, go


# The DCN layer shapes of the benched step (32 images), each at the smallest batch at which the launch plan
# (csrc/dcn.hip make_plan / pick_bm) picks the kernels the bench batch runs -- asserted from the library's launch log:
#   64 -> 64 at 128 x 128 and 128 -> 64 at 64 x 64, each on either side of the 512-tile rule of the data-gradient walk:
#     the smaller batch takes coord_grad + col2im (what the 32 x 32 / 16 x 16 maps of the bench run), the larger one
#     (B * tiles = 512) the ONE-launch walk `dcn_bwd_data_kernel` on `dcn_prep_kernel`'s geometry records (what its
#     128 x 128 / 64 x 64 maps run); forward: the LDS-window kernel `dcnw_fwd_kernel` with its column side output;
#   128 -> 128 at 64 x 64: the 128-row `DcnFwdLoader` tile;
#   256 -> 256 and 256 -> 128 at 32 x 32: the two-kernel forward (dcn_sample_kernel + a 128- / 64-row GEMM over the columns).
DCN_LAYERS = {
    '64to64_128sq': dict(B=4, C=64, Co=64, S=128, kernels=['dcnw_fwd_kernel<64, 32>', 'dcn_col2im_kernel', 'dcn_coord_grad_kernel', 'igemm_fwd_shortk_kernel', 'igemm_wgrad_ws_kernel<DcnColWBufLoader, 64, 64>']),
    '64to64_128sq_one_launch': dict(B=8, C=64, Co=64, S=128, kernels=['dcnw_fwd_kernel<64, 32>', 'dcn_bwd_data_kernel', 'dcn_prep_kernel', 'igemm_fwd_shortk_kernel']),
    '128to64_64sq': dict(B=16, C=128, Co=64, S=64, kernels=['dcnw_fwd_kernel<64, 32>', 'dcn_col2im_kernel', 'dcn_coord_grad_kernel', 'igemm_wgrad_ws_kernel<DcnColWBufLoader, 64, 128>']),
    '128to64_64sq_one_launch': dict(B=32, C=128, Co=64, S=64, kernels=['dcnw_fwd_kernel<64, 32>', 'dcn_bwd_data_kernel', 'dcn_prep_kernel']),
    '32to64_100sq': dict(B=8, C=32, Co=64, S=100, kernels=['igemm_fwd_kernel<64, DcnFwdLoaderT<true>']),      # a map the window kernels do not take (100 % 16 != 0)
    # round 6: the maps of a 640 x 640 input (configs[4]) and an odd multiple of 32 -- the window forward on 5 x 32-, 5 x 16- and
    # 3 x 32-column tiles; the data-gradient walk on 8 x 32-pixel tiles (5 x 32 = 160; 3 x 32 = 96 > 80: a ragged last column tile)
    '64to64_160sq': dict(B=2, C=64, Co=64, S=160, kernels=['dcnw_fwd_kernel<64, 32>', 'dcn_col2im_kernel', 'dcn_coord_grad_kernel']),
    '64to64_160sq_one_launch': dict(B=6, C=64, Co=64, S=160, kernels=['dcnw_fwd_kernel<64, 32>', 'dcn_bwd_data_kernel', 'dcn_prep_kernel']),
    '128to64_80sq_one_launch': dict(B=18, C=128, Co=64, S=80, kernels=['dcnw_fwd_kernel<64, 16>', 'dcn_bwd_data_kernel', 'dcn_prep_kernel']),
    '32to64_96sq': dict(B=8, C=32, Co=64, S=96, kernels=['dcnw_fwd_kernel<64, 32>']),
    '128to128_64sq': dict(B=16, C=128, Co=128, S=64, kernels=['igemm_fwd_kernel<128, DcnFwdLoaderT<true>']),
    '256to256_32sq': dict(B=32, C=256, Co=256, S=32, kernels=['dcn_sample_kernel', 'igemm_fwd_ws_kernel<128, DcnColsBufLoader']),
    '256to128_32sq': dict(B=32, C=256, Co=128, S=32, kernels=['dcn_sample_kernel', 'igemm_fwd_ws_kernel<64, DcnColsBufLoader']),
}
# offset scales: 1e-3 px of either sign is the benched step after its first optimizer step (a sample at y - 1e-4 has its anchor
# one cell above its neighbour's at y + 1e-4: jittered anchors, all four corner weights non-zero, one of them ~1)
_DCN_CASES = [(n, o) for n in DCN_LAYERS for o in ((0.001, 0.3, 1.0, 6.0) if DCN_LAYERS[n]['S'] >= 64 and DCN_LAYERS[n]['Co'] == 64 and DCN_LAYERS[n]['C'] >= 64 else (1.0,))]


@pytest.mark.parametrize('layer,off_scale', _DCN_CASES,      # 0.3 px: col2im's DPP ranking path; 6 px: strays beyond the LDS window
                         ids=['%s-pm%gpx' % (n, o) for n, o in _DCN_CASES])
def test_full_size_dcn_layer_matches_the_oracle(layer, off_scale):
    _full_size_dcn_layer(layer, off_scale)


def _full_size_dcn_layer(layer, off_scale):
    """`dcn_v2_cuda.cu:42-341` at the layer shapes the bench runs, VALUES against the CPU oracle (forward, the saved
    columns, all five gradients, 1e-4 of each tensor's magnitude), through the product's autograd path (which keeps
    the columns) and through the literal `_ext` entry points (which do not) -- and again while a second stream runs
    convolutions on the same CUs: everything but grad_input (straggler atomics) must then be bit-identical."""
    import _ext
    import hip_runtime as hr
    from hip_runtime import ops
    from libs.DCNv2.dcn_v2 import dcn_v2_conv
    from oracle import dcn as od
    from test_zz_kernel_coverage import short
    B, C, Co, S = (DCN_LAYERS[layer][k] for k in ('B', 'C', 'Co', 'S'))
    x, w, b, off, m, go = _dcn_full_case(B, C, Co, S, off_scale, 1000 * C + S + int(10 * off_scale))
    geom = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    want_y = od.dcn_v2_forward(x, w, b, off, m, *geom)
    want = od.dcn_v2_backward(x, w, b, off, m, go, *geom)              # input, offset, mask, weight, bias
    # the reference's columns (rows (c, tap), dcn_v2_im2col_cuda.cu:152-186) of image 0 and B-1, as (tap, c) rows
    col = torch.empty(C * 9, S * S)
    want_cols = {}
    for bi in (0, B - 1):
        od._fn('im2col', torch.float32)(od._p(x[bi]), od._p(off[bi]), od._p(m[bi]), od._p(col),
                                        *od._ints(C, S, S, S, S, 3, 3, 1, 1, 1, 1, 1, 1, 1))
        want_cols[bi] = col.view(C, 9, S * S).permute(1, 0, 2).reshape(9 * C, S * S).clone()
    dev = [t.to(DEV) for t in (x, w, b, off, m, go)]

    def close(got, ref, what):
        scale = ref.abs().max().item()
        err = (got.double().cpu() - ref.double()).abs().max().item()
        assert err <= 1e-4 * scale, (what, err, scale)

    def autograd_path():
        leaves = [t.clone().requires_grad_(True) for t in (dev[0], dev[3], dev[4], dev[1], dev[2])]   # x, off, m, w, b
        y = dcn_v2_conv(*leaves, 1, 1, 1, 1)
        cols = y.grad_fn.saved_tensors[5]
        y.backward(dev[5])
        return [y.detach()] + [l.grad for l in leaves] + [cols.clone()]

    with hr.launch_log() as log:
        res = autograd_path()
    names = sorted(short(n) for n in log.names)
    print(names)
    import os
    if not any(os.environ.get(v) == '0' for v in ('CNUDA_BUF', 'CNUDA_WS', 'CNUDA_SHORTK', 'CNUDA_DCNW')):     # (tests/test_gpu_kernel_switches.py)
        for want_kernel in DCN_LAYERS[layer]['kernels']:
            assert any(n.startswith(want_kernel) for n in names), (want_kernel, names)
    y, gx, goff, gm, gw, gb, cols = res
    close(y, want_y, 'forward')
    for bi, wc in want_cols.items():
        close(cols[bi], wc, 'saved columns of image %d' % bi)
    for got, ref, what in zip((gx, goff, gm, gw, gb), want, ('input', 'offset', 'mask', 'weight', 'bias')):
        close(got, ref, 'grad_' + what)
    # the literal native entry points (no saved columns: the weight gradient re-samples)
    close(_ext.dcn_v2_forward(dev[0], dev[1], dev[2], dev[3], dev[4], *geom), want_y, '_ext forward')
    for got, ref, what in zip(_ext.dcn_v2_backward(*dev, *geom), want, ('input', 'offset', 'mask', 'weight', 'bias')):
        close(got, ref, '_ext grad_' + what)
    # ... and with convolutions resident on the same CUs
    x3 = torch.randn(32, 128, 64, 64, device=DEV, requires_grad=True)
    w3 = (torch.randn(128, 128, 3, 3, device=DEV) * 0.05).requires_grad_(True)
    g3 = torch.randn(32, 128, 64, 64, device=DEV)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for it in range(4):
        with torch.cuda.stream(side):
            for _ in range(3):
                y3 = ops.conv2d(x3, w3, None, 1, 1)
                if it % 2:
                    y3.backward(g3)
        r2 = autograd_path()
        torch.cuda.synchronize()
        for i, what in ((0, 'forward'), (2, 'grad_offset'), (3, 'grad_mask'), (4, 'grad_weight'), (5, 'grad_bias'),
                        (6, 'saved columns')):
            assert torch.equal(r2[i], res[i]), (it, what)
        close(r2[1], want[0], 'grad_input beside another stream')
