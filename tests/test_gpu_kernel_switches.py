"""The kernel-selection switches that remain (`CNUDA_BUF=0`: pointer-addressed loaders -- also what tensors of 2 GiB
and more take; `CNUDA_WS=0`: the 4-wave kernels on the 64- / 128-row tiles -- also what matrix mode 1 takes;
`CNUDA_SHORTK=0`: the pipelined kernel for the K = 64 column-gradient GEMM; `CNUDA_HCONV=0`: the im2col-style kernels
for the 27-row DCN offset convolutions that otherwise take the halo-tile kernels; `CNUDA_DCNW=0`: the gathering loader
for the DCN forward of the layers that otherwise sample from an LDS window; `CNUDA_SPLITK=0`: pick_bm's smaller row tiles
for the starved long-K GEMMs that otherwise cut K over the grid; `CNUDA_DCOL_QUADS=0`: the DCN column gradient in plain
[9 C][pixel] rows for the layers that otherwise interleave its rows in quads) are read once per process, so each
alternate runs the operator-level parity tests in a child process: per-operator values against the oracle at small
sizes, the full-size convolution and DCN value checks.  (The switches whose alternate lost in round 2 are gone.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUBSET = ['tests/test_gpu_ops.py::test_conv2d_fwd_bwd', 'tests/test_gpu_dcn.py::test_forward_backward_vs_oracle',
          'tests/test_gpu_fullsize.py::test_full_size_1x1_convolutions_match_fp64_and_repeat',
          'tests/test_gpu_fullsize.py::test_full_size_3x3_convolution_matches_fp64',
          'tests/test_gpu_fullsize.py::test_full_size_dcn_layer_matches_the_oracle']


@pytest.mark.parametrize('switch', ['CNUDA_BUF', 'CNUDA_WS', 'CNUDA_SHORTK', 'CNUDA_HCONV', 'CNUDA_DCNW', 'CNUDA_SPLITK', 'CNUDA_DCOL_QUADS'])
def test_alternate_kernel_paths_hold_the_same_parity(switch):
    env = dict(os.environ, **{switch: '0'})
    # (the full-size DCN layers at one offset scale and without the opt-in backward: the alternates differ in loaders and
    # tiles, not in what the offsets or that kernel exercise)
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider',
                        '-k', '(not dcn_layer) or pm1px'] + SUBSET,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
    assert ' passed' in r.stdout and 'skipped' not in r.stdout.splitlines()[-1], r.stdout[-500:]
