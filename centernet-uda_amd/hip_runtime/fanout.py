"""Gradient fan-in without the engine's elementwise adds.

Where a tensor feeds several consumers, torch's autograd engine sums their gradients with one `at::add` pass per extra
consumer (57 of them in a DLA-34 UDA step: block inputs that also are skip connections, tree nodes that also enter a
root's concatenation, a DCN input that also feeds the offset convolution).  `fork(x, n)` hands every consumer its own
alias of `x`; the aliases share a GradSlot, and the consumers' backward passes meet there instead of in the engine:

  * the first consumer to run its backward writes its share straight into the slot (its output IS the buffer);
  * a convolution adds the slot's content in the epilogue of its input-gradient GEMM (cnuda_conv2d_backward_data_add),
    a DCN lets its data-gradient atomics land on top of it (cnuda_dcn_v2_backward_acc): no extra pass;
  * anything else -- or a consumer that does not know about slots -- returns its gradient as usual and `_Fork.backward`
    adds it with the library's own `cnuda_add`, in place when the running total is a buffer the slot allocated.

A slot belongs to ONE fork.  An alias forked again gets a fresh slot that remembers its parent: the inner fork's total
is one contribution to the parent, and a convolution of the inner fork whose parent already holds an owned buffer
writes `own share + parent's content + inner content` there in one epilogue (two addends) -- the inner slot then is an
ALIAS of its parent (`GradSlot.up`), its fork hands the parent's buffer up, and the parent recognises its own storage:
nothing is counted twice, at any nesting depth.
Outside a grad-enabled graph `fork` returns `x` itself n times.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import check, f32c, lib, ptr, stream


class GradSlot:
    """Running total of one fork's gradient shares.  A slot that has merged into its parent (`up`) is from then on a pure
    alias of it: `buf`, `owned` and `included` resolve through the chain, so every consumer -- whichever level's alias
    it holds, and however far the parents have merged upwards since -- sees the ONE buffer the totals live in."""
    __slots__ = ('_buf', '_owned', '_included', 'parent', 'up')

    def __init__(self, parent=None):
        self._buf = None      # running total of the shares that went through the slot so far
        self._owned = False   # buf was allocated for this slot (nobody else holds it): adding into it in place is safe
        self.parent = parent  # the slot of the alias this fork was made from (nested forks)
        self.up = None        # the slot this one's total has been folded into (accumulate_target's parent merge)
        self._included = []   # tensors slot-aware consumers wrote and handed to autograd as their gradient: already in the
                              # total when `_Fork.backward` meets them (held here, so their storage cannot be reused meanwhile)

    def root(self):
        s = self
        while s.up is not None:
            s = s.up
        return s

    @property
    def buf(self):
        return self.root()._buf

    @buf.setter
    def buf(self, t):
        self.root()._buf = t

    @property
    def owned(self):
        return self.root()._owned

    @owned.setter
    def owned(self, v):
        self.root()._owned = v

    @property
    def included(self):
        return self.root()._included

    def has(self, g):
        p = g.data_ptr()
        return any(t.data_ptr() == p for t in self.included)


def slot_of(t):
    return getattr(t, '_cnuda_slot', None)


def add_into(slot, g):
    """slot total += g (an arbitrary gradient tensor); -> the tensor that now holds the total."""
    g = f32c(g)
    if slot.has(g):
        return slot.buf
    if slot.buf is None:
        slot.buf, slot.owned = g, False
    else:
        out = slot.buf if slot.owned else torch.empty_like(slot.buf)
        check(lib().cnuda_add(ptr(slot.buf), ptr(g), ptr(out), g.numel(), stream()), 'add')
        slot.buf, slot.owned = out, True
    return slot.buf


def claim(slot, t):
    """A consumer that produces a whole fresh gradient `t`: an empty slot takes it as its buffer (the fork then has nothing to
    add for this consumer); otherwise `t` stays private and `_Fork.backward` adds it.  -> t"""
    if slot is not None and slot.buf is None:
        slot.buf, slot.owned = t, True
        slot.included.append(t)
    return t


def first_writer(slot, like):
    return claim(slot, torch.empty_like(like))


def accumulate_target(slot, like):
    """For a consumer that can add what the slot holds while it writes (convolution input gradient):
    -> (out, addend, addend2): write `out = own share + addend + addend2` (None: nothing to add).  The slot then holds `out`."""
    if slot is None:
        return torch.empty_like(like), None, None
    slot = slot.root()
    par = None if slot.parent is None else slot.parent.root()
    if par is not None and par.buf is not None and par.owned:
        # merge into the parent's buffer: parent's content + this slot's content + the consumer's share, in place.  From
        # here on the slot is an alias of its parent (`up`): whatever it held is folded exactly once, and a later
        # consumer -- also one that arrives after the parent itself has merged further up -- adds to the live total.
        out, addend, addend2 = par.buf, par.buf, slot._buf
        par._included.extend(slot._included)
        par._included.append(out)
        slot._buf, slot._owned, slot._included, slot.up = None, False, [], par
        return out, addend, addend2
    if slot.buf is None:
        out, addend = torch.empty_like(like), None
    elif slot.owned:
        out = addend = slot.buf                       # in place
    else:
        out, addend = torch.empty_like(like), slot.buf
    slot.buf, slot.owned = out, True
    slot.included.append(out)
    return out, addend, None


class _Fork(Function):
    @staticmethod
    def forward(ctx, x, n, slot):
        ctx.slot = slot
        ctx.set_materialize_grads(False)       # an alias whose consumer never reaches a loss: None, not a tensor of zeros
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        slot = ctx.slot
        for g in gs:
            if g is not None:
                add_into(slot, g)
        if slot.up is not None:            # folded into an outer fork's total: hand that buffer up, the outer fork knows it
            total, slot.up = slot.buf, None    # (this fork's last act in the pass: a later pass over the graph starts unmerged)
            return total, None, None
        total, slot._buf = slot._buf, None
        slot._included = []
        return total, None, None


def fork(x, n=2):
    """n aliases of x, one per consumer (each alias must go to exactly ONE consumer)."""
    if n <= 1 or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    slot = GradSlot(slot_of(x))
    outs = _Fork.apply(x, n, slot)
    for o in outs:
        o._cnuda_slot = slot
    return outs
