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
    """Step 0 must agree to rounding (same gradients bit for bit, parameters within Adam's own rounding); later
    steps are compared through the losses only: at this fixture size (B = 2 at 64 x 64, 8 samples per channel in
    the deepest BatchNorm) a one-ulp parameter difference changes the next gradient by 1e-3 -- measured between
    these very two optimizers -- so element-wise parameter equality after several steps is not a property."""
    B, S, M = 2, 64, 8
    runs = []
    for stock in (False, True):
        plugin, model = _plugin(golden, stock)
        if parallel:                                     # the wrapper without a process group: one rank
            from hip_runtime.parallel import DataParallel
            plugin.backend = DataParallel(model)
        hist = []
        for step in range(3):
            before = {n: p.detach().clone() for n, p in model.named_parameters()}
            data = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (3, 2), 2, 51 + step).items()}
            data['input'] = T(gin.image_batch(B, S, S, 60 + step))
            data['target_domain_input'] = T(gin.image_batch(B, S, S, 70 + step))
            stats = {k: float(v) for k, v in plugin.step(data)['stats'].items()}
            params = {n: p.detach().clone() for n, p in model.named_parameters()}
            grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in model.named_parameters()}
            moved = sum(int((params[n] != before[n]).sum()) for n in params)
            hist.append((stats, params, grads, moved))
        runs.append(hist)
    fused, stock = runs
    # step 0: identical gradients (this step's, nothing stale, nothing missing), parameters equal to Adam's rounding
    for n, g0 in fused[0][2].items():
        g1 = stock[0][2][n]
        assert (g0 is None or float(g0.abs().sum()) == 0.0) == (g1 is None or float(g1.abs().sum()) == 0.0), n
        if g1 is not None and g0 is not None:
            assert torch.equal(g0, g1), n
        assert (fused[0][1][n] - stock[0][1][n]).abs().max().item() <= 2e-7 * (1.0 + fused[0][1][n].abs().max().item()), n
    for step in range(3):
        a, b = fused[step][0], stock[step][0]
        for k in a:
            assert abs(a[k] - b[k]) <= (1e-6 if step == 0 else 2e-3) * max(abs(a[k]), 1e-6), (step, k, a[k], b[k])
        # every step really trains: nearly every trainable element moves
        total = sum(p.numel() for n, p in stock[step][1].items() if stock[step][2][n] is not None)
        assert stock[step][3] >= 0.9 * total, (step, stock[step][3], total)


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
