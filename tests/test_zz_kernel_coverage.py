"""Kernel coverage of the benched step (collected last: the file name sorts after every other test file).

1. One training step of EVERY BASELINE.json config at its full size (bench.py's own builder and bench.CONFIGS: the
   headline DLA-34 + 16 DCNv2, 512 x 512, 16 source + 16 target images, entropy minimisation, Adam; ResNet-18 at
   256 x 256; no UDA; max-squares; rotated + periodic + ADVENT at 640 x 640) runs under the library's launch log; every
   kernel it launches must be a key of tests/kernel_manifest.py.
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


def _bench_step_kernels(idx):
    """One training step of BASELINE.json configs[idx] at its FULL size, built by bench.py's own builder (bench.CONFIGS:
    backend, UDA method, input size, per-GPU batch), under the library's launch log -> {kernel: launches}."""
    sys.path.insert(0, ROOT)
    import bench
    import hip_runtime as hr
    import torch
    dev = torch.device('cuda', 0)
    backend_name, uda_name, size, batch_n = bench.CONFIGS[idx]
    plugin = bench.build_plugin(dev, parallel=False, uda_name=uda_name, backend_name=backend_name)
    batch = bench.synthetic_batch(batch_n, size, 42, dev, rotated=bench.UDA_WORKLOADS[uda_name][2])
    for _ in range(2):
        plugin.step(batch)
    torch.cuda.synchronize()
    with hr.launch_log() as log:
        out = plugin.step(batch)
        torch.cuda.synchronize()
    stats = {k: float(v) for k, v in out['stats'].items()}
    del plugin, batch, out
    torch.cuda.empty_cache()
    return {short(k): v for k, v in log.counts.items()}, stats


@pytest.mark.parametrize('idx', [2, 0, 1, 3, 4], ids=lambda i: 'configs%d' % i)
def test_every_kernel_of_the_benched_step_has_an_oracle_value_test(idx):
    """configs[2] is the headline; [0] ResNet-18 256 x 256 B = 2, [1] no UDA, [3] max-squares, [4] rotated boxes +
    periodic angle loss + ADVENT discriminator at 640 x 640 -- every one at the size and batch BASELINE.json names."""
    import math
    from kernel_manifest import MANIFEST
    launched, stats = _bench_step_kernels(idx)
    print('\n'.join('%5d  %s' % (v, k) for k, v in sorted(launched.items(), key=lambda kv: -kv[1])))
    assert len(launched) > (20 if idx == 0 else 30), launched     # the log works and the step is the real one
    assert stats and all(math.isfinite(v) for v in stats.values()), stats
    missing = sorted(k for k in launched if k not in MANIFEST)
    assert not missing, ('kernels of the configs[%d] step without an entry in tests/kernel_manifest.py '
                         '(add an oracle-value test that selects each, then name it there): %s' % (idx, missing))


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
