"""`python bench.py --gpus N` started plainly spawns its N ranks itself (fresh children of torch.distributed.run,
decided from the environment before anything touches the GPU) and exits non-zero when a rank fails."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=900):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    env.pop('LOCAL_RANK', None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=timeout)


def test_plain_multi_gpu_start_spawns_ranks_and_propagates_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip('covered by the GPU variant below')
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0', '--no-cpu-baseline'])
    assert r.returncode != 0
    # the refusal comes from a spawned RANK (no GPU here), not from an argument check in the parent: torchrun prefixes
    # nothing, but it reports the failed child (`rank : N (local_rank: N)`) and ends the survivor as soon as the first
    # rank fails, so the message is guaranteed once, not twice
    assert r.stderr.count('bench.py needs an MI355X') >= 1, r.stderr[-2000:]
    assert 'local_rank' in r.stderr, r.stderr[-2000:]


def test_world_size_mismatch_is_refused():
    r = _run(['--gpus', '2'], {'WORLD_SIZE': '4', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'WORLD_SIZE=4' in r.stderr


@pytest.mark.gpu
def test_plain_two_rank_start_prints_one_json_line():
    r = _run(['--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '2', '--size', '128', '--no-cpu-baseline',
              '--no-extras'], {'CNUDA_BENCH_ONE_DEVICE': '1', 'CNUDA_BENCH_BACKEND': 'gloo'})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['steps'] == 2 and line['config']['global_batch'] == 4
    assert line['value'] > 0 and line['scaling'] == 'weak'
    assert line['roofline']['kernel'] and line['roofline']['achieved'] > 0
    # the N > 1 line proves what the exchange step did: the process group really had two ranks, every parameter
    # gradient of DLA-34 went through the bucketed all-reduce, and the exposed (non-overlapped) time is reported
    c = line['collective']
    assert c['backend'] == 'gloo' and c['world_size'] == 2 and c['steps'] == 2
    assert c['bytes_per_step'] >= 4 * 18e6 and c['buckets_per_step'] >= 2          # ~18.5 M parameters, 24 MB buckets
    assert c['host_wait_ms_per_step'] >= 0 and 'exposed_allreduce_ms_per_step' in c
    assert line['ms_per_step_sd'] >= 0 and line['ms_per_step_min'] <= line['ms_per_step'] * 1.05


@pytest.mark.gpu
def test_single_gpu_run_with_the_rccl_leg_leaves_exactly_one_line_on_stdout():
    # the one-rank RCCL leg brings a communicator up; this RCCL build prints a version banner on STDOUT when such a
    # process exits -- after the bench line.  The driver reads stdout: nothing but the line may be there.
    r = _run(['--steps', '1', '--warmup', '1', '--batch', '2', '--size', '128', '--no-cpu-baseline'])
    assert r.returncode == 0, r.stderr[-3000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(out) == 1 and out[0].startswith('{'), out
    line = json.loads(out[0])
    assert line['n_gpus'] == 1 and line.get('dp1_rccl'), sorted(line)
