"""The alias plan of the aggregation (backends/dla.py `_FanPlan`): every tensor of DLAUp / IDAUp is forked into exactly
as many aliases as it has readers -- found by walking the loops over tensor ids.  Host logic only: the modules are
replaced by pass-throughs, the trunk by six leaf tensors; what is checked is the bookkeeping (no alias left over, no
reader without an alias), for the configurations the reference builds (down_ratio 4 / 2 / 8, last_level 5 / 6)."""
import os
import sys
from unittest import mock

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))


@pytest.mark.parametrize('down_ratio,last_level', [(4, 5), (2, 5), (8, 5), (4, 6)])
def test_every_alias_of_the_aggregation_has_exactly_one_reader(down_ratio, last_level):
    from backends import dla
    from hip_runtime import nn as hnn
    seg = dla.DLASeg('dla34', {'hm': 3, 'wh': 2}, pretrained=False, down_ratio=down_ratio, final_kernel=1,
                     last_level=last_level, head_conv=64)
    chans = seg.base.channels
    feats = [torch.zeros(1, c, 4, 4, requires_grad=True) for c in chans]
    pools, fallbacks = [], []
    real_init, real_take = dla._Aliases.__init__, dla._Aliases.take

    def init(self, t, n):
        real_init(self, t, n)
        pools.append((self, n))

    def take(self):
        if self.pool is not None and not self.pool:
            fallbacks.append(self)
        return real_take(self)

    with mock.patch.object(dla.DeformConv, 'forward', lambda self, x: x), \
            mock.patch.object(hnn.DepthwiseConvTranspose2d, 'forward', lambda self, x, skip=None: x if skip is None else x + 0 * skip.sum()), \
            mock.patch.object(type(seg.base), 'forward', lambda self, x: list(feats)), \
            mock.patch.object(dla._Aliases, '__init__', init), mock.patch.object(dla._Aliases, 'take', take):
        y = seg.features(torch.zeros(1, 3, 16, 16))
    assert y.requires_grad
    assert not fallbacks, 'a tensor had more readers than the plan gave it aliases'
    left = [(n, len(a.pool)) for a, n in pools if a.pool]
    assert not left, 'aliases without a reader: %s' % left
    assert any(n >= 2 for _, n in pools)          # the plan does fork something: levels re-used as skip and as input
    # (the sums themselves run on the GPU: tests/test_gpu_fanout.py, tests/test_gpu_dla.py)
