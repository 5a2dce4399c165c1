"""The reference's driver builds `torch.optim.<name>` (train.py:88-90).  A step driven by stock torch.optim.Adam
(whose zero_grad() sets every `.grad` to None) must train exactly like the fused arena Adam: same losses, same
parameters after every step -- the gradient sink stays authoritative whatever the optimizer (hip_runtime/arena.py)."""
import ast

import numpy as np
import pytest
import torch

import inputs as gin

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _plugin(golden, stock):
    import uda
    from backends import dla
    from hip_runtime import optim
    from losses.centernet import DetectionLoss
    shapes = dict(ast.literal_eval(str(golden('dla_axis')['shapes_json'])))
    model = dla.build(num_classes=6)
    model.load_state_dict({k: T(v) for k, v in gin.fill_state(shapes, 0.1).items()})
    plugin = uda.EntropyMinimization(1e-2)
    plugin.backend = model.to(DEV)
    plugin.device = torch.device(DEV)
    params = [p for p in model.parameters() if p.requires_grad]
    plugin.optimizer = (torch.optim.Adam if stock else optim.Adam)(params, lr=5e-5, weight_decay=1e-4)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0)
    plugin.init_done()
    plugin.to(DEV)
    plugin.set_phase(True)
    return plugin, model


@pytest.mark.parametrize('parallel', [False, True])
def test_stock_torch_adam_trains_like_the_fused_adam(golden, parallel):
    B, S, M = 2, 64, 8
    runs = []
    for stock in (False, True):
        plugin, model = _plugin(golden, stock)
        if parallel:                                     # the wrapper without a process group: one rank
            from hip_runtime.parallel import DataParallel
            plugin.backend = DataParallel(model)
        hist = []
        for step in range(3):
            data = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (3, 2), 2, 51 + step).items()}
            data['input'] = T(gin.image_batch(B, S, S, 60 + step))
            data['target_domain_input'] = T(gin.image_batch(B, S, S, 70 + step))
            hist.append({k: float(v) for k, v in plugin.step(data)['stats'].items()})
        runs.append((hist, {n: p.detach().clone() for n, p in model.named_parameters()},
                     {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in model.named_parameters()}))
    (h0, p0, g0), (h1, p1, g1) = runs
    for a, b in zip(h0, h1):
        for k in a:
            assert abs(a[k] - b[k]) <= 1e-4 * max(abs(a[k]), 1e-6), (k, a[k], b[k])
    assert any(g is not None for g in g1.values())
    for n in p0:
        assert (g0[n] is None or float(g0[n].abs().sum()) == 0.0) == (g1[n] is None or float(g1[n].abs().sum()) == 0.0), n
        # Adam moves every element by <= lr per step: three steps of rounding-level disagreement stay far below that
        d = (p0[n] - p1[n]).abs()
        assert d.max().item() <= 3 * 2.1 * 5e-5, n
        if not n.endswith('.conv.bias'):
            assert (d > 1e-6).float().mean().item() <= 0.02, (n, (d > 1e-6).float().mean().item())


def test_out_of_range_ind_poisons_the_loss_instead_of_touching_memory():
    """`ind` outside [0, H*W): the reference's torch.gather device-asserts (utils/tensor.py:10-18); here the loss is
    NaN and nothing out of bounds is read or written."""
    from losses.centernet import DetectionLoss
    out_np, batch_np, w = gin.loss_inputs('plain')
    for bad in (16 * 16, -1, 10 ** 12):
        out = {k: T(v).to(DEV).requires_grad_(True) for k, v in out_np.items()}
        batch = {k: T(v).clone().to(DEV) for k, v in batch_np.items()}
        batch['ind'][1, 2] = bad
        loss, stats = DetectionLoss(**w)({k: v * 1.0 for k, v in out.items()}, batch)
        assert torch.isnan(loss).item() and torch.isnan(stats['wh_loss']).item()
        assert torch.isfinite(stats['hm_loss']).item()
        loss.backward()
        torch.cuda.synchronize()
        assert all(v.grad is not None for v in out.values())
