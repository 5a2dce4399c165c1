"""One pass over the concatenated source | target batch (hip_runtime.domain_groups, DLASeg.forward_domains,
uda.base.Model.batch_domains) against the reference's literal sequence -- two forward calls, two backward calls
(uda/entropy_minimization.py:18-19,31-32): same outputs, running statistics (Q6), losses and gradients up to
floating-point summation order."""
import ast

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import inputs as gin

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


@pytest.mark.parametrize('shape', [(4, 16, 10, 12), (6, 5, 7, 7), (2, 64, 2, 2), (8, 32, 33, 31),
                                   (16, 512, 4, 4)])             # (last: several images per workgroup, 2 groups of 8)
@pytest.mark.parametrize('relu,res', [(False, False), (True, True)])
def test_grouped_batch_norm_equals_two_calls(shape, relu, res):
    from hip_runtime import ops
    g = torch.Generator().manual_seed(5)
    B, C, H, W = shape
    x = (torch.randn(shape, generator=g) * 2 + 0.5)
    x[B // 2:] = x[B // 2:] * 1.7 - 1.0                     # the two domains have different statistics
    r = torch.randn(shape, generator=g) if res else None
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    gy = torch.randn(shape, generator=g)
    rm0, rv0 = torch.randn(C, generator=g) * 0.1, 1 + 0.3 * torch.rand(C, generator=g)

    def run(grouped):
        xs = x.to(DEV).requires_grad_(True)
        rs = r.to(DEV).requires_grad_(True) if res else None
        ga, be = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
        rm, rv, nbt = rm0.to(DEV), rv0.to(DEV), torch.tensor(0, dtype=torch.long, device=DEV)
        if grouped:
            y = ops.batch_norm_act(xs, ga, be, rm, rv, True, 0.1, 1e-5, rs, relu, nbt, groups=2)
            y.backward(gy.to(DEV))
        else:
            h = B // 2
            ys = [ops.batch_norm_act(xs[i * h:(i + 1) * h], ga, be, rm, rv, True, 0.1, 1e-5,
                                     None if rs is None else rs[i * h:(i + 1) * h], relu, nbt) for i in range(2)]
            y = torch.cat(ys)
            ys[0].backward(gy[:h].to(DEV))
            ys[1].backward(gy[h:].to(DEV))
        return y.detach(), xs.grad, None if rs is None else rs.grad, ga.grad, be.grad, rm, rv, int(nbt)
    a, b = run(True), run(False)
    assert a[7] == b[7] == 2
    for u, v, name in zip(a[:7], b[:7], ['y', 'gx', 'gres', 'ggamma', 'gbeta', 'running_mean', 'running_var']):
        if u is None:
            continue
        assert _rel(u, v) <= 2e-6, (name, _rel(u, v))
    # ... and both equal CPU torch, domain by domain
    h = B // 2
    rm, rv = rm0.clone(), rv0.clone()
    want = []
    for i in range(2):
        yy = F.batch_norm(x[i * h:(i + 1) * h], rm, rv, gamma, beta, True, 0.1, 1e-5)
        if res:
            yy = yy + r[i * h:(i + 1) * h]
        want.append(F.relu(yy) if relu else yy)
    assert _rel(a[0].cpu(), torch.cat(want)) <= 1e-5
    assert _rel(a[5].cpu(), rm) <= 1e-6 and _rel(a[6].cpu(), rv) <= 1e-6


def _plugin(golden, batched, kind='entropy'):
    import uda
    from backends import dla
    from hip_runtime import optim
    from losses.centernet import DetectionLoss
    shapes = dict(ast.literal_eval(str(golden('dla_axis')['shapes_json'])))
    model = dla.build(num_classes=6)
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, 0.1).items()})
    plugin = uda.EntropyMinimization(1e-2) if kind == 'entropy' else uda.MaxSquaresMinimization(0.3)
    plugin.batch_domains = batched
    plugin.backend = model.to(DEV)
    plugin.device = torch.device(DEV)
    plugin.optimizer = optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, weight_decay=1e-4)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0)
    plugin.init_done()
    plugin.to(DEV)
    plugin.set_phase(True)
    return plugin, model


@pytest.mark.parametrize('kind', ['entropy', 'maxsq'])
def test_batched_step_equals_the_sequential_step(golden, kind):
    B, S, M = 4, 128, 8
    res = []
    for batched in (True, False):
        plugin, model = _plugin(golden, batched, kind)
        data = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (3, 2, 7, 1), 2, 51).items()}
        data['input'] = T(gin.image_batch(B, S, S, 60))
        data['target_domain_input'] = T(gin.image_batch(B, S, S, 70))
        out = plugin.step(data)
        res.append((out, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                    {n: b.detach().clone() for n, b in model.named_buffers()}))
    (ob, gb, bb), (os_, gs, bs) = res
    assert list(ob['stats']) == list(os_['stats'])
    for k in ob['stats']:
        assert abs(float(ob['stats'][k]) - float(os_['stats'][k])) <= 1e-5 * max(abs(float(os_['stats'][k])), 1e-9), k
    for dom in ('source_domain', 'target_domain'):
        assert list(ob[dom]) == list(os_[dom]) == ['hm', 'wh', 'reg']
        for k in ob[dom]:
            assert ob[dom][k].shape == os_[dom][k].shape
            assert _rel(ob[dom][k].detach(), os_[dom][k].detach()) <= 1e-5, (dom, k)
    assert sorted(gb) == sorted(gs)                       # the same parameters receive a gradient
    for n in gs:
        if n.endswith('.conv.bias') and 'ida' in n:
            continue                                      # analytically zero (bias in front of a BatchNorm): noise
        assert _rel(gb[n], gs[n]) <= 2e-4, (n, _rel(gb[n], gs[n]))
    for n in bs:
        if bs[n].is_floating_point():
            assert _rel(bb[n], bs[n]) <= 1e-5, n
        else:
            assert int(bb[n]) == int(bs[n]) == 2, n        # Q6: two BatchNorm updates per step
    # the target heads that feed no loss are evaluated without a tape; like the reference's they still say
    # requires_grad, and a loss that does reach one fails loudly instead of training nothing
    assert ob['target_domain']['hm'].requires_grad and ob['target_domain']['wh'].requires_grad
    import pytest
    with pytest.raises(RuntimeError, match='target_grad_heads'):
        ob['target_domain']['wh'].sum().backward()


def test_batched_advent_step_equals_the_sequential_step(golden):
    """S4 with the two backbone passes batched: (detection loss + fooling loss).backward() in one pass over
    source | target against the reference's literal five-backward sequence."""
    import uda
    from backends import dla
    from hip_runtime import optim
    from losses.centernet import DetectionLoss
    g = golden('step_advent')
    shapes = dict(ast.literal_eval(str(g['shapes_json'])))
    dshapes = dict(ast.literal_eval(str(g['dshapes_json'])))

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    res = []
    B, S, M = 4, 128, 8
    for batched in (True, False):
        model = dla.build(num_classes=6, rotated_boxes=True)
        model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, 0.1).items()})
        plugin = uda.AdversarialEntropyMinimization(1e-2, optimizer=Cfg(name='Adam', params=Cfg(lr=1e-3, weight_decay=1e-4)))
        plugin.batch_domains = batched
        plugin.cfg = Cfg(max_detections=20, model=Cfg(backend=Cfg(params=Cfg(rotated_boxes=True, num_classes=6))))
        plugin.backend = model
        plugin.device = torch.device(DEV)
        plugin.optimizer = optim.Adam(model.parameters(), lr=5e-5, weight_decay=1e-4)
        plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0, periodic=True)
        plugin.init_done()
        plugin.discriminator.load_state_dict({k: T(gin.fill_value('discriminator.' + k, tuple(v))) for k, v in dshapes.items()})
        plugin.to(DEV)
        plugin.set_phase(True)
        data = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (4, 2, 1, 6), 3, 71).items()}
        data['input'] = T(gin.image_batch(B, S, S, 72))
        data['target_domain_input'] = T(gin.image_batch(B, S, S, 73))
        out = plugin.step(data)
        res.append((out['stats'], {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None},
                    {n: p.grad.detach().clone() for n, p in plugin.discriminator.named_parameters()}))
    (sb, gb, db), (ss, gs, ds) = res
    assert set(sb) == set(ss) == {'centernet_loss', 'hm_loss', 'wh_loss', 'off_loss', 'total_loss', 'dis_soruce',
                                  'dis_target', 'dis_fool'}
    for k in ss:
        assert abs(float(sb[k]) - float(ss[k])) <= 1e-5 * max(abs(float(ss[k])), 1e-9), k
    assert sorted(gb) == sorted(gs)
    for n in gs:
        if n.endswith('.conv.bias') and 'ida' in n:
            continue
        assert _rel(gb[n], gs[n]) <= 2e-4, (n, _rel(gb[n], gs[n]))
    for n in ds:
        assert _rel(db[n], ds[n]) <= 2e-4, n
