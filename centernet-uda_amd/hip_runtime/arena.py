"""Flat parameter / gradient arena.

All trainable tensors of a module live in ONE contiguous fp32 buffer and their
gradients in another (`p.data` / `p.grad` become views).  That turns the
optimizer into a single kernel launch and the data-parallel gradient exchange
into a few large RCCL all-reduces over contiguous slices -- the layout the
288 GB HBM / xGMI design wants -- instead of 233 small tensors.

Parameters that never receive a gradient (the reference's discarded
level3/level4 `project` branches, its unused `base.fc`; Q8) are tracked with
`touched` flags so the optimizer skips them exactly like torch.optim does for
`grad is None`.

Gradient sink.  The kernels' autograd Functions do not hand parameter gradients to autograd (which would add
each of the 233 tensors into `.grad` with its own tiny kernel, twice per step): `grad_sink(param)` gives them a
slot of a second flat buffer (`staging`, same layout) to write the gradient of THIS backward pass into; they
return None for the parameter.  autograd still runs the parameter's post-accumulate hook when the Function
returns, which is where the slot is recorded; `flat_grad += staging` then happens in a few large launches over
runs of recorded slots -- per all-reduce bucket as soon as it is complete (data parallel), otherwise once at
the end of the pass (an engine callback queued from the first hook).
"""
import weakref

import torch

# Neither table keeps an arena (and through it a model's parameters) alive: the optimizer / data-parallel wrapper
# that asked for the arena owns it, and the parameters' hooks reference it for as long as the parameters live.
_REGISTRY = weakref.WeakValueDictionary()      # ids of the parameter set -> arena
_BY_PTR = {}          # data_ptr of a parameter inside some arena -> (weak reference to the arena, index)


def grad_sink(param):
    """-> a float32 view (shaped like `param`) the caller must fill with the gradient of the running backward
    pass and then NOT return to autograd; or None when `param` is not arena-managed (return the gradient as
    usual).  A parameter used twice in one pass gets None the second time (autograd accumulates that one)."""
    hit = _BY_PTR.get(param.data_ptr())
    if hit is None:
        return None
    arena, i = hit[0](), hit[1]
    if arena is None:                               # the arena is gone and the address was reused
        del _BY_PTR[param.data_ptr()]
        return None
    if not arena.valid_index(i, param) or arena.sunk[i]:
        return None
    return arena._sink(i)



class ParamArena:
    ALIGN = 64          # elements; keeps every tensor 256-byte aligned inside the arena

    def __init__(self, params):
        self.params = list(params)
        if not self.params:
            raise ValueError("ParamArena: no parameters")
        # every check comes before the first parameter is re-pointed: a half-built arena must not exist
        frozen = [i for i, p in enumerate(self.params) if not p.requires_grad]
        if frozen:
            raise RuntimeError("ParamArena: %d of %d parameters do not require grad (first at index %d); pass only "
                               "trainable parameters (arena_for() filters them)" % (len(frozen), len(self.params), frozen[0]))
        dev = self.params[0].device
        # host-side bookkeeping only: works on any device (the gloo tests run it on CPU tensors);
        # the kernels that consume the arena (Adam, RCCL) require the GPU themselves
        self.offsets, off = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("ParamArena: all parameters must be float32 on one device")
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.touched = [False] * len(self.params)
        self.staging = None             # gradients of the running backward pass (allocated on first use)
        self.sunk = [False] * len(self.params)      # staging slot written in this pass, not yet added to flat_grad
        self._flush_queued = False
        self._bulk_zeroed = False       # flat_grad was cleared for this pass because every .grad had been set to None
        self._hooks = []
        self.on_ready = None            # callback(index) used by the data-parallel wrapper
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                self.flat_param[o:o + n].copy_(p.data.reshape(-1))
                p.data = self.flat_param[o:o + n].view(p.shape)
                gview = self.flat_grad[o:o + n].view(p.shape)
                if p.grad is not None:
                    gview.copy_(p.grad)
                    self.touched[i] = True
                p.grad = gview
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
                _BY_PTR[p.data_ptr()] = (weakref.ref(self), i)

    # -- gradient sink -------------------------------------------------------------
    def valid_index(self, i, param):
        return self.params[i] is param or self.params[i].data_ptr() == param.data_ptr()

    def _sink(self, i):
        if self.staging is None:
            self.staging = torch.zeros_like(self.flat_grad)      # the alignment gaps stay zero for ever
        self.sunk[i] = True
        o, p = self.offsets[i], self.params[i]
        return self.staging[o:o + p.numel()].view(p.shape)

    def flush(self, lo=0, hi=None):
        """flat_grad += staging over the sunk parameters whose offsets lie in [lo, hi): one launch per run of
        neighbouring slots (the alignment gaps between them are included: both buffers hold zeros / don't-care
        there and the gaps are never read)."""
        hi = self.numel if hi is None else hi
        run = None
        for i, o in enumerate(self.offsets):
            if o < lo or o >= hi:
                continue
            if self.sunk[i]:
                end = o + self.params[i].numel()
                run = [o, end] if run is None else [run[0], end]
                self.sunk[i] = False
            elif run is not None:
                self.flat_grad[run[0]:run[1]].add_(self.staging[run[0]:run[1]])
                run = None
        if run is not None:
            self.flat_grad[run[0]:run[1]].add_(self.staging[run[0]:run[1]])

    def _end_of_pass(self):
        self._flush_queued = False
        self._bulk_zeroed = False
        self.flush()

    def _adopt_cleared_grad(self, i, param):
        """`param.grad is None` when its hook runs: an optimizer that does not know the arena (stock
        torch.optim.*, whose zero_grad() defaults to set_to_none=True) dropped the gradient view.  The arena
        stays authoritative: whatever `flat_grad` still holds for this parameter belongs to an earlier step and
        is cleared, then `.grad` is bound to the view again -- before this pass's staged gradient is added."""
        o, n = self.offsets[i], param.numel()
        if not self._bulk_zeroed:
            if all(p.grad is None for p in self.params):
                self.flat_grad.zero_()                          # the usual case: one memset for the whole step
                self.touched = [False] * len(self.params)
                self._bulk_zeroed = True
                if not self._flush_queued:
                    self._flush_queued = True                   # _end_of_pass() re-arms the bulk clear
                    torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)
            else:
                self.flat_grad[o:o + n].zero_()
        param.grad = self.flat_grad[o:o + n].view(param.shape)

    def _make_hook(self, i):
        wself = weakref.ref(self)       # a hook lives in the parameter's C++ autograd node: a strong reference
                                        # from there would make arena <-> parameters an uncollectable cycle

        def hook(param):
            self = wself()
            if self is None:            # the arena's owners (optimizer, wrapper) are gone
                return
            if param.grad is None:
                self._adopt_cleared_grad(i, param)
            self.touched[i] = True
            if self.sunk[i] and not self._flush_queued:
                # runs inside backward(): the engine calls this back when the pass is over
                self._flush_queued = True
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_pass)
            # autograd may rebind .grad (it does not when .grad is defined and no graph is built);
            # keep the arena authoritative
            o, n = self.offsets[i], param.numel()
            if param.grad is not None and param.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                self.flat_grad[o:o + n].view(param.shape).copy_(param.grad)
                param.grad = self.flat_grad[o:o + n].view(param.shape)
            if self.on_ready is not None:
                self.on_ready(i)
        return hook

    def valid(self):
        """False when something (e.g. module.to(), load_state_dict with assign) re-pointed a parameter."""
        base = self.flat_param.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def zero_grad(self):
        self.flat_grad.zero_()
        self.sunk = [False] * len(self.params)
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            self.touched[i] = False
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view(p.shape)

    def touched_runs(self):
        """Maximal [start, end) element ranges covering only touched parameters."""
        runs, start, end = [], None, None
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            nxt = self.offsets[i + 1] if i + 1 < len(self.params) else self.numel
            if self.touched[i]:
                if start is None:
                    start = o
                end = nxt
            elif start is not None:
                runs.append((start, end))
                start = None
        if start is not None:
            runs.append((start, end))
        return runs

    def release(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for k in [k for k, v in _BY_PTR.items() if v[0]() is self or v[0]() is None]:
            del _BY_PTR[k]


def arena_for(params):
    """One arena per set of TRAINABLE parameters: the optimizer (which may have been given frozen parameters
    too, e.g. `model.parameters()` with freeze_base=True) and the data-parallel wrapper (which filters by
    requires_grad) resolve to the same arena."""
    params = [p for p in params if p.requires_grad]
    if not params:
        raise ValueError("arena_for: none of the parameters requires grad")
    key = tuple(id(p) for p in params)
    a = _REGISTRY.get(key)
    if a is not None and not a.valid():
        a.release()
        a = None
    if a is None:
        a = ParamArena(params)
        _REGISTRY[key] = a
    return a
