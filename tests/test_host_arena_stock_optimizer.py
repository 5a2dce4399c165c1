"""The gradient sink and the data-parallel wrapper with an optimizer that does not know the arena (the
reference's driver builds `torch.optim.<name>`, train.py:88-90, whose zero_grad() sets `.grad = None`):
the arena has to stay authoritative -- `.grad` rebound to its view, stale sums cleared -- or the optimizer
would skip every parameter while `flat_grad` silently accumulates across steps.  CPU tensors; the kernels'
autograd Functions are stood in for by one that writes its parameter gradient into `grad_sink` like they do."""
import gc
import weakref

import pytest
import torch
from torch import nn


class _SunkLinear(torch.autograd.Function):
    """y = x @ w^T; the weight gradient goes to the arena's staging slot, autograd gets None for it."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, gy):
        from hip_runtime.arena import grad_sink
        x, w = ctx.saved_tensors
        gw = gy.t() @ x
        slot = grad_sink(w)
        if slot is not None:
            slot.copy_(gw)
            gw = None
        return gy @ w, gw


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.w1 = nn.Parameter(torch.randn(5, 4) * 0.3)
        self.w2 = nn.Parameter(torch.randn(3, 5) * 0.3)
        self.unused = nn.Parameter(torch.randn(2, 2))

    def forward(self, x):
        return _SunkLinear.apply(torch.relu(_SunkLinear.apply(x, self.w1)), self.w2)


def _plain_grads(net, xs):
    ref = _Net()
    ref.load_state_dict(net.state_dict())
    for x in xs:
        (torch.relu(x @ ref.w1.t()) @ ref.w2.t()).pow(2).sum().backward()
    return ref


@pytest.mark.parametrize('set_to_none', [True, False])
@pytest.mark.parametrize('wrap', [False, True])
def test_stock_torch_optimizer_sees_this_steps_gradients(set_to_none, wrap):
    from hip_runtime.arena import arena_for
    from hip_runtime.parallel import DataParallel
    torch.manual_seed(0)
    net = _Net()
    model = DataParallel(net) if wrap else net
    arena = model.arena if wrap else arena_for(net.parameters())
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    ref_net = _Net()
    ref_net.load_state_dict(net.state_dict())
    ref_opt = torch.optim.SGD(ref_net.parameters(), lr=0.1)
    for step in range(3):
        xs = [torch.randn(6, 4), torch.randn(6, 4)]
        opt.zero_grad(set_to_none=set_to_none)
        ref_opt.zero_grad()
        # two backward calls per step (uda/entropy_minimization.py:31-32); the first under no_sync when wrapped
        if wrap:
            with model.no_sync():
                model(xs[0]).pow(2).sum().backward()
        else:
            model(xs[0]).pow(2).sum().backward()
        model(xs[1]).pow(2).sum().backward()
        if wrap:
            model.finish_gradient_sync()
        for x in xs:
            (torch.relu(x @ ref_net.w1.t()) @ ref_net.w2.t()).pow(2).sum().backward()
        for n in ('w1', 'w2'):
            p, r = getattr(net, n), getattr(ref_net, n)
            assert p.grad is not None, (step, n)
            assert p.grad.data_ptr() == arena.flat_grad.data_ptr() + 4 * arena.offsets[['w1', 'w2', 'unused'].index(n)]
            torch.testing.assert_close(p.grad, r.grad, rtol=1e-5, atol=1e-6)     # this step's sum, nothing stale
        assert net.unused.grad is None or float(net.unused.grad.abs().sum()) == 0.0
        opt.step()
        ref_opt.step()
        for n in ('w1', 'w2', 'unused'):
            torch.testing.assert_close(getattr(net, n).data, getattr(ref_net, n).data, rtol=1e-5, atol=1e-6)


def test_arena_skips_frozen_parameters_and_is_shared():
    from hip_runtime.arena import ParamArena, arena_for
    from hip_runtime.parallel import DataParallel
    net = _Net()
    net.w1.requires_grad_(False)                       # freeze_base=True
    before = [p.data_ptr() for p in net.parameters()]
    with pytest.raises(RuntimeError, match='do not require grad'):
        ParamArena(list(net.parameters()))             # rejected before any parameter is re-pointed
    assert before == [p.data_ptr() for p in net.parameters()]
    a = arena_for(net.parameters())                    # what an optimizer given model.parameters() asks for
    assert [p is q for p, q in zip(a.params, (net.w2, net.unused))] == [True, True]
    assert DataParallel(net).arena is a                # the wrapper filters by requires_grad: same arena
    with pytest.raises(ValueError):
        arena_for([net.w1])


def test_registry_does_not_keep_models_alive():
    from hip_runtime import arena as A
    net = _Net()
    a = A.arena_for(net.parameters())
    ref_net, ref_arena = weakref.ref(net), weakref.ref(a)
    del net, a
    gc.collect()
    assert ref_net() is None and ref_arena() is None
    # a new tensor at a recycled address must not resolve to the dead arena
    p = nn.Parameter(torch.zeros(5, 4))
    assert A.grad_sink(p) is None


def test_average_meter_has_the_reference_signature():
    from utils.helper import AverageMeter
    m = AverageMeter(name='hm_loss')                   # train.py:164,182,243
    m.update(2.0, n=2)
    m.update(4.0)
    assert abs(m.avg - 8.0 / 3) < 1e-12 and m.val == 4.0 and m.count == 3
    assert str(m) == 'hm_loss 4.000000 (2.666667)'
    assert str(AverageMeter('t', ':.2f')) == 't 0.00 (0.00)'
    with pytest.raises(TypeError):
        AverageMeter()
