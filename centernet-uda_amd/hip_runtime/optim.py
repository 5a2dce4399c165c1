"""Optimizers on the flat arena.  `Adam` has torch.optim.Adam's constructor and
arithmetic (the driver resolves `torch.optim.<name>` from the config,
train.py:88-90; this build resolves the same names through `resolve`)."""
import torch

from . import ops
from .arena import arena_for


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("hip_runtime.optim.Adam supports a single parameter group")
        self._arena = None
        self._m = self._v = None
        self._step = 0

    def _ensure(self):
        params = self.param_groups[0]['params']
        if self._arena is None or not self._arena.valid():
            old = (self._m, self._v, self._arena)
            self._arena = arena_for(params)
            self._m = torch.zeros_like(self._arena.flat_param)
            self._v = torch.zeros_like(self._arena.flat_param)
            if old[2] is not None and old[0] is not None and old[0].numel() == self._m.numel():
                self._m.copy_(old[0])
                self._v.copy_(old[1])
            for p, o in zip(self._arena.params, self._arena.offsets):
                n = p.numel()
                self.state[p] = {'step': torch.tensor(float(self._step)),
                                 'exp_avg': self._m[o:o + n].view(p.shape),
                                 'exp_avg_sq': self._v[o:o + n].view(p.shape)}
        return self._arena

    def zero_grad(self, set_to_none=False):
        # gradients are views of the arena: zeroing is one memset, never `None`
        self._ensure().zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        a = self._ensure()
        g = self.param_groups[0]
        self._step += 1
        for start, end in a.touched_runs():
            ops.adam_step_(a.flat_param[start:end], a.flat_grad[start:end], self._m[start:end], self._v[start:end],
                           g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay'], self._step)
        for p in a.params:
            self.state[p]['step'].fill_(float(self._step))
        from . import bump_param_epoch
        # the kernel wrote the flat arena: torch's version counters did not move; cached packed weights follow
        bump_param_epoch(a.flat_param)
        return loss

    def load_state_dict(self, state_dict):
        a = self._ensure()            # before the base class fills self.state: _ensure() rebinds every entry
        super().load_state_dict(state_dict)
        steps = [int(s['step']) for s in self.state.values() if 'step' in s]
        self._step = max(steps) if steps else 0
        for p, o in zip(a.params, a.offsets):
            st, n = self.state.get(p, {}), p.numel()
            if 'exp_avg' in st:
                self._m[o:o + n].copy_(st['exp_avg'].reshape(-1))
                self._v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
            self.state[p] = {'step': torch.tensor(float(self._step)),
                             'exp_avg': self._m[o:o + n].view(p.shape),
                             'exp_avg_sq': self._v[o:o + n].view(p.shape)}


def resolve(name):
    """`optimizer.name` from the config -> class (train.py:88: torch.optim.<name>)."""
    if name == 'Adam':
        return Adam
    raise NotImplementedError("optimizer %r: only Adam (the one every reference experiment config uses) "
                              "runs on the fused arena kernel" % name)
