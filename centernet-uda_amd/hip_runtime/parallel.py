"""Data parallelism for MI355X: one process per GPU, gradients of the flat arena
all-reduced by RCCL over xGMI in a few large buckets, launched from
gradient-ready hooks so that they overlap the remaining backward kernels.

Replaces the reference's single-process `CustomDataParallel(nn.DataParallel)`
(utils/helper.py:75-80; scatter / replicate / gather / ReduceAddCoalesced
through device 0).  Semantics: each rank owns its own per-GPU batch and its own
BatchNorm statistics (as DataParallel's replicas do); gradients are averaged
over ranks (RCCL's averaging all-reduce; sum then / world_size on gloo).  The reference's loss sees the gathered global
batch; losses.centernet.DetectionLoss.use_global_normalizers() (the default of
uda.base.Model.to(parallel=True)) scales every rank's loss so that this average
is the gradient of that global loss.

BatchNorm buffers: nn.DataParallel re-broadcasts device 0's buffers to the replicas on every forward and only
replica 0's in-place updates persist, so the running statistics the reference evaluates and checkpoints with are
those of replica 0's shard.  Here every rank updates its own statistics while training (they are not read in train
mode) and rank 0's are broadcast to all ranks by an EXPLICIT collective, `sync_buffers()`, which
`uda.base.Model.set_phase(False)` calls on the switch from training to evaluation: evaluation on any rank then sees
exactly replica 0's statistics, and rank 0 -- the rank that writes checkpoints -- holds them by construction.
`train()` / `eval()` themselves are plain nn.Module calls and communicate nothing.

Two (or more) backward() calls per step accumulate locally under `no_sync()`;
buckets fire during the last backward; `finish_gradient_sync()` launches any
bucket that did not become ready (parameters without a gradient) and makes the
compute stream wait for the collectives.
"""
import contextlib

import torch
import torch.distributed as dist
from torch import nn

from .arena import arena_for

BUCKET_BYTES = 24 << 20      # xGMI links are point-to-point: few, large messages


class DataParallel(nn.Module):
    def __init__(self, module, process_group=None, bucket_bytes=BUCKET_BYTES):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self._sync = True
        self._works = []
        # RCCL averages inside the collective (ncclAvg), gloo sums and finish_gradient_sync() scales: decided HERE from the
        # process group's backend, so that a step in which no bucket was launched cannot read an undecided flag as 'scale'
        self._avg_in_collective = bool(dist.is_available() and dist.is_initialized() and
                                       dist.get_backend(self.process_group) == 'nccl')
        # what the exchange step did since the last reset_exchange_stats(): bytes / buckets all-reduced, and the time the
        # compute stream spent waiting for collectives that had not finished under the backward pass ("exposed")
        self.measure_exchange = False
        self._xstats = {'steps': 0, 'bytes': 0, 'buckets': 0, 'events': [], 'host_wait_s': 0.0}
        params = [p for p in module.parameters() if p.requires_grad]
        self.arena = arena_for(params)
        self.arena.on_ready = self._param_ready
        # buckets in reverse parameter order (~ the order gradients become ready)
        self.buckets, cur_end, cur_start = [], self.arena.numel, self.arena.numel
        limit = max(1, bucket_bytes // 4)
        self.bucket_of = [0] * len(params)
        for i in range(len(params) - 1, -1, -1):
            cur_start = self.arena.offsets[i]
            self.bucket_of[i] = len(self.buckets)
            if cur_end - cur_start >= limit or i == 0:
                self.buckets.append([cur_start, cur_end, 0, False])     # start, end, n_params, launched
                cur_end = cur_start
        for b in self.bucket_of:
            self.buckets[b][2] += 1
        self._pending = [b[2] for b in self.buckets]
        if hasattr(module, 'forward_domains'):          # offered exactly when the wrapped backend offers it
            self.forward_domains = self._forward_domains
        if self.world_size > 1:
            # replicas start identical (DataParallel broadcasts parameters and buffers every forward)
            dist.broadcast(self.arena.flat_param, src=0, group=self.process_group)
            from . import bump_param_epoch
            bump_param_epoch()
            self.sync_buffers()

    @property
    def world_size(self):
        return dist.get_world_size(self.process_group) if dist.is_available() and dist.is_initialized() else 1

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)       # .down_ratio, .rotated_boxes, ... like CustomDataParallel

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def _forward_domains(self, *args, **kwargs):
        # explicit (not through __getattr__): anything forward() grows must be mirrored here, not silently skipped
        return self.module.forward_domains(*args, **kwargs)

    def sync_buffers(self):
        """Every rank's buffers (BatchNorm running statistics, batch counters) := rank 0's; one broadcast per
        dtype over a flattened copy (330 tensors for DLA-34).  Collective: all ranks must call it."""
        if self.world_size <= 1:
            return
        by_dtype = {}
        for buf in self.module.buffers():
            by_dtype.setdefault(buf.dtype, []).append(buf)
        for bufs in by_dtype.values():
            flat = torch.cat([b.detach().reshape(-1) for b in bufs])
            dist.broadcast(flat, src=0, group=self.process_group)
            off = 0
            with torch.no_grad():
                for b in bufs:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()

    def train(self, mode=True):
        """Plain nn.Module.train(): NO collective.  (Round 3 broadcast the buffers from inside this method; a rank that
        validated alone then hung its peers in a call that does not look like communication.)  The switch to
        evaluation with replica 0's statistics is explicit: `sync_buffers()`, which uda.base.Model.set_phase(False)
        calls on every rank."""
        return super().train(mode)

    @contextlib.contextmanager
    def no_sync(self):
        prev, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = prev

    # -- measurement ---------------------------------------------------------------
    def reset_exchange_stats(self, measure=True):
        self.measure_exchange = measure
        self._xstats = {'steps': 0, 'bytes': 0, 'buckets': 0, 'events': [], 'host_wait_s': 0.0}

    def exchange_stats(self):
        """Per step since reset_exchange_stats(): bytes and buckets all-reduced, and the exposed (not overlapped with
        backward) all-reduce time: device time between the point where the compute stream starts waiting for the
        collectives and the point where the last one has finished (events on the compute stream around the waits of
        finish_gradient_sync); `host_wait_ms` is the host side of the same waits (all of it for a host-blocking
        backend such as gloo).  Call after a device synchronisation."""
        x = self._xstats
        n = max(1, x['steps'])
        backend = dist.get_backend(self.process_group) if dist.is_available() and dist.is_initialized() else None
        rccl = None
        if backend == 'nccl':
            try:
                rccl = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
        return {'backend': backend, 'world_size': self.world_size, 'rccl_version': rccl,
                'bytes_per_step': x['bytes'] // n, 'buckets_per_step': x['buckets'] / n,
                'bucket_bytes_limit': BUCKET_BYTES,
                'exposed_allreduce_ms_per_step': (sum(a.elapsed_time(b) for a, b in x['events']) / n) if x['events'] else None,
                'host_wait_ms_per_step': 1e3 * x['host_wait_s'] / n, 'steps': x['steps']}

    # -- bucket machinery ----------------------------------------------------------
    def _reset(self):
        self._pending = [b[2] for b in self.buckets]
        for b in self.buckets:
            b[3] = False

    def _launch(self, k):
        b = self.buckets[k]
        if b[3]:
            return
        b[3] = True
        self.arena.flush(b[0], b[1])                            # this pass's sunk gradients of the bucket
        if dist.is_available() and dist.is_initialized():      # also with one rank: the collective is an identity
            chunk = self.arena.flat_grad[b[0]:b[1]]
            self._xstats['bytes'] += chunk.numel() * chunk.element_size()
            self._xstats['buckets'] += 1
            self._works.append(dist.all_reduce(chunk, op=self._reduce_op(), group=self.process_group, async_op=True))

    def _reduce_op(self):
        """RCCL averages inside the collective (ncclAvg: every rank's contribution is scaled by 1 / world as it is
        read, no pass of its own over the 78.6 MB arena); gloo has no such operator -> SUM, and
        finish_gradient_sync() scales."""
        return dist.ReduceOp.AVG if self._avg_in_collective else dist.ReduceOp.SUM

    def _param_ready(self, i):
        if not self._sync:
            return
        k = self.bucket_of[i]
        self._pending[k] -= 1
        if self._pending[k] == 0:
            self._launch(k)

    def finish_gradient_sync(self):
        for k in range(len(self.buckets)):
            self._launch(k)
        measure = self.measure_exchange and self._works and self.arena.flat_grad.is_cuda
        if measure:
            import time
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
        for w in self._works:
            w.wait()
        if measure:
            e1.record()
            self._xstats['events'].append((e0, e1))
            self._xstats['host_wait_s'] += time.perf_counter() - t0
        self._xstats['steps'] += 1
        self._works = []
        if self.world_size > 1 and not self._avg_in_collective:
            self.arena.flat_grad.mul_(1.0 / self.world_size)       # (gloo only: the CPU / shared-GPU test backends)
        self._reset()
