"""Kernel coverage of the benched step (collected last: the file name sorts after every other test file).

1. One training step of the bench workload (bench.py's own builder: DLA-34 + 16 DCNv2, 512 x 512, 16 source + 16
   target images, entropy minimisation, Adam) runs under the library's launch log; every kernel it launches must be a
   key of tests/kernel_manifest.py.
2. Every manifest entry is checked against what its tests really launched in THIS session (tests/conftest.py records
   the launch log around every GPU test): the named tests must exist, and at least one of them must have launched the
   kernel."""
import os
import re
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """demangled kernel symbol -> manifest key: no `void`, no namespaces, no argument list, no `, false` default of
    the split-operand flag"""
    n = name.replace('void ', '').replace('cnuda::(anonymous namespace)::', '').replace('cnuda::', '')
    depth, out = 0, []
    for ch in n:                      # cut the argument list: the first '(' outside template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            break
        out.append(ch)
    return re.sub(r'\s+', ' ', ''.join(out)).strip()


def _bench_step_kernels():
    sys.path.insert(0, ROOT)
    import bench
    import hip_runtime as hr
    import torch
    dev = torch.device('cuda', 0)
    plugin = bench.build_plugin(dev, parallel=False, uda_name='entropy', backend_name='dla34')
    batch = bench.synthetic_batch(16, 512, 42, dev)
    for _ in range(2):
        plugin.step(batch)
    torch.cuda.synchronize()
    with hr.launch_log() as log:
        plugin.step(batch)
        torch.cuda.synchronize()
    del plugin, batch
    torch.cuda.empty_cache()
    return {short(k): v for k, v in log.counts.items()}


def test_every_kernel_of_the_benched_step_has_an_oracle_value_test():
    from kernel_manifest import MANIFEST
    launched = _bench_step_kernels()
    print('\n'.join('%5d  %s' % (v, k) for k, v in sorted(launched.items(), key=lambda kv: -kv[1])))
    assert len(launched) > 30, launched            # the log works and the step is the real one
    missing = sorted(k for k in launched if k not in MANIFEST)
    assert not missing, ('kernels of the benched step without an entry in tests/kernel_manifest.py '
                         '(add an oracle-value test that selects each, then name it there): %s' % missing)


def test_manifest_tests_really_launch_their_kernels(request):
    import conftest
    from kernel_manifest import MANIFEST
    ran = {nodeid: {short(k) for k in ks} for nodeid, ks in conftest.KERNELS_BY_TEST.items()}
    filtered = bool(request.config.option.keyword) or any('::' in a or a.endswith('.py') for a in request.config.args)
    collected = [it.nodeid for it in request.session.items]
    wrong, unverified = [], []
    for kernel, prefixes in sorted(MANIFEST.items()):
        assert prefixes, kernel
        hit = False
        for pre in prefixes:
            if not filtered:
                assert any(c.startswith(pre) for c in collected), ('manifest names a test that does not exist', kernel, pre)
            tests = [t for t in ran if t.startswith(pre)]
            if any(kernel in ran[t] for t in tests):
                hit = True
            elif tests:
                wrong.append((kernel, pre))
        if not hit:
            unverified.append(kernel)
    assert not wrong, 'tests that the manifest credits with a kernel they did not launch: %s' % wrong
    if unverified and filtered:
        pytest.skip('partial session: %d manifest entries not exercised' % len(unverified))
    assert not unverified, 'no named test launched: %s' % unverified
