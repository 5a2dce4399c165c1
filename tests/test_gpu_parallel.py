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
    assert s0['centernet_loss'] != s1['centernet_loss']            # different shards ...
    assert g0 == g1 and w0 == w1                                   # ... identical gradients and parameters after the exchange
    assert g0 > 0
