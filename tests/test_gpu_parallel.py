"""MI355X: the data-parallel wrapper on the real device path -- flat gradient arena on the GPU, gradient-ready
hooks fired by the HIP autograd Functions, bucketed `dist.all_reduce` over the `nccl` (= RCCL) backend while the
rest of backward is still running.  A one-GPU box can only host a one-rank group (the collective is then an
identity), so the numbers must equal the unwrapped model's bit for bit; the world-size-2 arithmetic is covered
on CPU by tests/test_parallel_gloo.py."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import inputs as gin

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


def _plugin(parallel):
    import uda
    from backends import resnet
    from hip_runtime import optim
    from losses.centernet import DetectionLoss
    torch.manual_seed(3)
    backend = resnet.build(18, num_classes=6, pretrained=False)
    plugin = uda.EntropyMinimization(1e-2)
    plugin.backend = backend
    plugin.device = torch.device(DEV)
    plugin.centernet_loss = DetectionLoss(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0,
                                          periodic=False)
    plugin.to(DEV, parallel)
    plugin.optimizer = optim.Adam([p for p in plugin.backend.parameters() if p.requires_grad], lr=1e-3)
    plugin.init_done()
    plugin.set_phase(True)
    return plugin


def _batch():
    B, S, M = 2, 64, 8
    data = {k: T(v) for k, v in gin.detection_batch(B, 6, S // 4, S // 4, M, (3, 2), 2, 81).items()}
    data['input'] = T(gin.image_batch(B, S, S, 82))
    data['target_domain_input'] = T(gin.image_batch(B, S, S, 83))
    return data


def test_one_rank_rccl_group_matches_plain_step():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        from hip_runtime.parallel import DataParallel
        wrapped, plain = _plugin(True), _plugin(False)
        assert isinstance(wrapped.backend, DataParallel) and wrapped.backend.down_ratio == 4
        assert len(wrapped.backend.buckets) >= 2                                   # 11.7 M parameters, 24 MB buckets
        for step in range(2):                                                     # the second step reuses the buckets
            a, b = wrapped.step(_batch()), plain.step(_batch())
            for k in b['stats']:
                assert float(a['stats'][k]) == float(b['stats'][k]), (step, k)
        pa = dict(wrapped.backend.module.named_parameters())
        for n, p in plain.backend.named_parameters():
            assert torch.equal(pa[n], p), n
            assert torch.equal(pa[n].grad, p.grad), n
        assert not wrapped.backend._works                                          # every async collective was waited for
    finally:
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, 'tests', 'golden'), root, os.path.join(root, 'centernet-uda_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)      # two ranks cannot share a GPU under RCCL
    try:
        plugin = _plugin(True)
        data = _batch()
        # different images per rank (the partition of section 8e): shift the batch of rank 1
        if rank == 1:
            data['input'] = data['input'].flip(0).contiguous()
            data['target_domain_input'] = data['target_domain_input'] * 0.5
        out = plugin.step(data)
        arena = plugin.backend.arena
        q.put((rank, {k: float(v) for k, v in out['stats'].items()},
               arena.flat_grad.double().abs().sum().item(), arena.flat_param.double().sum().item()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_share_one_gpu_through_gloo():
    """World size 2 on the real kernels (both ranks on cuda:0, collective over gloo): after the step both replicas
    hold the same averaged gradient and the same parameters, although they saw different images."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, s0, g0, w0), (_, s1, g1, w1) = res
    assert s0['entropy_loss'] != s1['entropy_loss']                # different shards (the UDA term is logged per rank) ...
    assert s0['centernet_loss'] == s1['centernet_loss']            # ... one detection loss over the gathered batch
    assert g0 == g1 and w0 == w1                                   # identical gradients and parameters after the exchange
    assert g0 > 0


def _loss_shards():
    """Two ranks' shards of one global batch; rank 1 holds NO object (num_pos == 0, empty masks)."""
    outs, batches = [], []
    for r, n_obj in enumerate([(3, 1), (0, 0)]):
        b = gin.detection_batch(2, 6, 16, 16, 8, n_obj, 2, 91 + r)
        rs = np.random.RandomState(191 + r)
        o = dict(hm=(rs.standard_normal((2, 6, 16, 16)) * 1.5 - 1.0).astype(np.float32),
                 wh=(rs.standard_normal((2, 2, 16, 16)) * 3.0).astype(np.float32),
                 reg=rs.standard_normal((2, 2, 16, 16)).astype(np.float32))
        outs.append(o)
        batches.append(b)
    return outs, batches


def _global_loss_worker(rank, world, port, q):
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, 'tests', 'golden'), root, os.path.join(root, 'centernet-uda_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from losses.centernet import DetectionLoss
        kw = dict(hm_weight=1.0, wh_weight=0.1, off_weight=1.0, angle_weight=1.0, periodic=False)
        outs, batches = _loss_shards()
        dev = lambda d: {k: T(v).to(DEV) for k, v in d.items()}
        crit = DetectionLoss(**kw)
        crit.use_global_normalizers()
        out = {k: v.requires_grad_(True) for k, v in dev(outs[rank]).items()}
        leaves = dict(out)
        loss, stats = crit(out, dev(batches[rank]))
        loss.backward()
        res = {'stats': {k: float(v) for k, v in stats.items()},
               'grads': {k: v.grad.double().cpu().numpy() for k, v in leaves.items()}}
        if rank == 0:
            # the reference's semantics: ONE loss over the gathered batch (single process, local normalisers)
            cat = lambda ds: {k: torch.cat([T(d[k]) for d in ds]).to(DEV) for k in ds[0]}
            gout = {k: v.requires_grad_(True) for k, v in cat(outs).items()}
            gleaves = dict(gout)
            gl, gstats = DetectionLoss(**kw)(gout, cat(batches))
            gl.backward()
            res['ref_stats'] = {k: float(v) for k, v in gstats.items()}
            res['ref_grads'] = {k: v.grad.double().cpu().numpy() for k, v in gleaves.items()}
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_global_normalizers_equal_the_gathered_batch_loss():
    """SURVEY 8e exact-match mode: with use_global_normalizers() the stats equal the single-process loss over the
    concatenated batch, and each rank's gradient is world_size times its slice of that loss's gradient (the
    data-parallel wrapper then averages) -- including a rank without any object (local num_pos == 0)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_global_loss_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref_s, ref_g = res[0]['ref_stats'], res[0]['ref_grads']
    for r in (0, 1):
        for k, v in ref_s.items():
            assert abs(res[r]['stats'][k] - v) <= 2e-6 * max(1.0, abs(v)), (r, k, res[r]['stats'][k], v)
        for k, g in res[r]['grads'].items():
            want = 2.0 * ref_g[k][2 * r:2 * r + 2]
            assert np.abs(g - want).max() <= 2e-6 * max(1e-6, np.abs(want).max()), (r, k)
    assert np.abs(res[1]['grads']['hm']).max() > 0          # the empty rank still gets the negative-term gradient
    assert np.abs(res[1]['grads']['wh']).max() == 0
