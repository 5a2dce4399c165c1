"""World-size-2 CPU (gloo) tests of the data-parallel path: flat arena, bucketing,
gradient-ready hooks, no_sync() accumulation over two backward calls, averaging,
parameters without gradients, parameter/buffer broadcast.  The collective is the
same `dist.all_reduce` the GPU path issues over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 5)
        self.unused = nn.Linear(3, 3)          # never receives a gradient (like base.level3.project, Q8)
        self.b = nn.Linear(5, 4)
        self.c = nn.Linear(4, 2)
        self.register_buffer('stat', torch.zeros(3))

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, bucket_bytes, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, 'centernet-uda_amd'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from hip_runtime.parallel import DataParallel
        torch.manual_seed(100 + rank)                     # ranks start DIFFERENT; wrap must broadcast rank 0
        net = _Net()
        net.stat.fill_(float(rank + 1))
        ref = _Net()
        dp = DataParallel(net, bucket_bytes=bucket_bytes)
        assert len(dp.buckets) >= (3 if bucket_bytes < 200 else 1)
        # parameters and buffers now equal rank 0's
        flat0 = dp.arena.flat_param.clone()
        dist.broadcast(flat0, src=0)
        assert torch.equal(flat0, dp.arena.flat_param)
        assert float(net.stat[0]) == 1.0
        assert dp.down_ratio_probe if False else True
        ref.load_state_dict(net.state_dict())
        g = torch.Generator().manual_seed(7 + rank)
        x1, x2 = torch.randn(8, 6, generator=g), torch.randn(8, 6, generator=g)
        # two backward calls per step, the first under no_sync (entropy_minimization.py:31-32)
        dp.arena.zero_grad()
        with dp.no_sync():
            dp(x1).pow(2).mean().backward()
        assert not dp._works                                # nothing launched yet
        (dp(x2).sum() * 0.1).backward()
        dp.finish_gradient_sync()
        # expected: mean over ranks of the locally accumulated gradients
        ref(x1).pow(2).mean().backward()
        (ref(x2).sum() * 0.1).backward()
        for (n, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
            if r.grad is None:
                assert n.startswith('unused')
                assert float(p.grad.abs().sum()) == 0.0
                continue
            want = r.grad.clone()
            dist.all_reduce(want)
            want /= world
            assert torch.allclose(p.grad, want, atol=1e-6), n
        idx = [i for i, p in enumerate(dp.arena.params) if p is net.unused.weight][0]
        assert dp.arena.touched[idx] is False and dp.arena.touched[0] is True
        runs = dp.arena.touched_runs()
        assert len(runs) == 2                               # the untouched pair splits the arena in two runs
        # a second step works (buckets were reset) and gradients are views of the arena
        dp.arena.zero_grad()
        dp(x1).sum().backward()
        dp.finish_gradient_sync()
        assert net.a.weight.grad.data_ptr() == dp.arena.flat_grad.data_ptr()
        # running statistics drift apart per rank while training and are rank 0's again at the switch to eval
        # (nn.DataParallel re-broadcasts device 0's buffers every forward, utils/helper.py:75-80)
        net.stat.fill_(10.0 * (rank + 1))
        net.register_buffer('count', torch.tensor(rank + 5, dtype=torch.int64))
        dp.train()
        assert float(net.stat[0]) == 10.0 * (rank + 1)            # train -> train: nothing is exchanged
        dp.eval()                                                 # plain nn.Module call: no hidden collective
        assert float(net.stat[0]) == 10.0 * (rank + 1) and not net.training
        dp.train()
        # the plugin's set_phase(False) is where the exchange happens, explicitly (uda/base.py)
        import uda.base
        plugin = uda.base.Model()
        plugin.backend = dp
        plugin.set_phase(True)
        assert float(net.stat[0]) == 10.0 * (rank + 1)
        plugin.set_phase(False)
        assert float(net.stat[0]) == 10.0 and int(net.count) == 5 and not net.training
        plugin.set_phase(False)                                   # eval -> eval: nothing is exchanged
        plugin.set_phase(True)
        q.put((rank, 'ok'))
    except Exception as e:                                  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('bucket_bytes', [64, 1 << 20])
def test_data_parallel_two_ranks(bucket_bytes):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, msg in results:
        assert msg == 'ok', 'rank %d: %s' % (rank, msg)


def test_single_process_wrapper_is_transparent():
    import sys
    from hip_runtime.parallel import DataParallel
    net = _Net()
    net.down_ratio = 4
    dp = DataParallel(net)
    assert dp.world_size == 1 and dp.down_ratio == 4
    dp(torch.randn(2, 6)).sum().backward()
    dp.finish_gradient_sync()
    assert net.a.weight.grad is not None
