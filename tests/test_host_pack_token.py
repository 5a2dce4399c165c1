"""Pack-cache identities of copied modules: a deep copy or an unpickled copy of a module owns other
weight buffers and must not share the original's pack token; a shallow copy shares the weights and keeps it."""
import copy
import pickle

import hip_runtime as hr
from hip_runtime import nn as hnn


def test_deep_copies_and_unpickled_modules_take_fresh_tokens():
    conv = hnn.Conv2d(4, 8, 3, padding=1)
    t0 = int(conv._pack_token)
    twin = copy.deepcopy(conv)
    assert int(twin._pack_token) != t0 and int(conv._pack_token) == t0
    again = pickle.loads(pickle.dumps(conv))
    assert int(again._pack_token) not in (t0, int(twin._pack_token))
    assert int(copy.copy(conv)._pack_token) == t0                     # shallow: same weights, same identity
    # inside a model: every copied layer gets its own identity
    import torch
    model = torch.nn.Sequential(hnn.Conv2d(3, 4, 1), hnn.Conv2d(4, 4, 3))
    ema = copy.deepcopy(model)
    ids = [int(m._pack_token) for m in list(model) + list(ema)]
    assert len(set(ids)) == 4, ids


def test_dcn_and_folded_blocks_use_token_objects():
    from backends import dla
    from libs.DCNv2.dcn_v2 import DCN
    d = DCN(4, 4, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)
    assert isinstance(d._pack_token, hr.PackToken)
    assert int(copy.deepcopy(d)._pack_token) != int(d._pack_token)
    blk = dla.BasicBlock(4, 4)
    blk.eval()
    blk.fold_batchnorm_()
    twin = copy.deepcopy(blk)
    assert {int(twin._fold_token), int(twin._fold_token2)}.isdisjoint({int(blk._fold_token), int(blk._fold_token2)})
