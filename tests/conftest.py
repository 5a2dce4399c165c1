import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'centernet-uda_amd')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (GOLDEN, ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    if os.environ.get('ABL_LIB'):      # A/B builds of the library (profiles/microbench/build_variant.sh): same tests, other .so
        import hip_runtime as hr
        hr.LIB_PATH = os.environ['ABL_LIB']


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


# ---------------------------------------------------------------------------
# Which of the library's kernels every GPU test launched (hip_runtime.launch_log: the library counts its own launches
# per kernel symbol).  tests/test_zz_kernel_coverage.py, collected last, holds this against tests/kernel_manifest.py:
# every kernel of the benched step must name an oracle-value test, and that test must really have launched it.
# CNUDA_DUMP_KERNELS=<path> writes the session's table as JSON (how the manifest was drawn up).
# ---------------------------------------------------------------------------
KERNELS_BY_TEST = {}


@pytest.fixture(autouse=True)
def _record_launched_kernels(request):
    if 'gpu' not in request.keywords:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    import hip_runtime as hr
    with hr.launch_log() as log:
        yield
    KERNELS_BY_TEST[request.node.nodeid] = dict(log.counts)


def pytest_sessionfinish(session, exitstatus):
    path = os.environ.get('CNUDA_DUMP_KERNELS')
    if path and KERNELS_BY_TEST:
        import json
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(KERNELS_BY_TEST, f, indent=0, sort_keys=True)
