"""Host-side (no GPU) checks of the product's DLA-34 module tree: parameter /
buffer names and shapes are the reference's checkpoint wire format."""
import ast

import numpy as np
import pytest


@pytest.mark.parametrize('tag,rotated', [('axis', False), ('rot', True)])
def test_state_dict_names_and_shapes_match_reference(golden, tag, rotated):
    from backends import dla
    g = golden('dla_' + tag)
    want = dict(ast.literal_eval(str(g['shapes_json'])))
    model = dla.build(num_classes=6, rotated_boxes=rotated)
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == tuple(want[k]), k
    assert sum(p.numel() for p in model.parameters()) == int(g['n_params'])
    assert model.down_ratio == 4 and model.rotated_boxes == rotated
    assert list(model.heads) == ['hm', 'wh', 'reg']


def test_initialisation_follows_reference():
    import torch
    from backends import dla
    m = dla.build(num_classes=6)
    sd = m.state_dict()
    assert torch.all(sd['hm.2.bias'] == -2.19)                                   # dla.py:485
    assert torch.all(sd['wh.0.bias'] == 0) and torch.all(sd['reg.2.bias'] == 0)    # fill_fc_weights
    assert torch.all(sd['ida_up.proj_1.conv.conv_offset_mask.weight'] == 0)       # dcn_v2.py:114-116
    assert torch.all(sd['dla_up.ida_0.node_1.conv.bias'] == 0)
    up = sd['ida_up.up_2.weight']                                                 # f = 4 -> k = 8 bilinear
    assert up.shape == (64, 1, 8, 8) and torch.allclose(up[0], up[17])
    f, c = 4, (2 * 4 - 1 - 0) / 8.0
    want = [(1 - abs(i / f - c)) for i in range(8)]
    assert torch.allclose(up[0, 0, 3], torch.tensor([want[3] * w for w in want]))
    w = sd['dla_up.ida_1.proj_1.conv.weight']
    assert w.abs().max() <= 1.0 / (256 * 9) ** 0.5 + 1e-7


def test_build_signature_and_no_cpu_fallback():
    import inspect
    import torch
    from backends import dla
    sig = inspect.signature(dla.build)
    assert list(sig.parameters) == ['num_classes', 'num_keypoints', 'head_conv', 'down_ratio', 'freeze_base',
                                    'rotated_boxes']
    m = dla.build(num_classes=2)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64))          # CPU tensors are refused, never silently computed


def test_build_loads_local_imagenet_trunk_or_warns(tmp_path, monkeypatch):
    """The reference's build() always starts from the ImageNet dla34 checkpoint (backends/dla.py:524-526,297-309);
    this build has no download: it loads the file when it is present (and then carries the unused `base.fc`, Q8)
    and warns loudly when it is not."""
    import torch
    from backends import dla
    monkeypatch.setenv('CNUDA_DLA34_WEIGHTS', str(tmp_path / 'missing.pth'))
    with pytest.warns(RuntimeWarning, match='RANDOMLY initialised'):
        m0 = dla.build(num_classes=3)
    assert not any(k.startswith('base.fc') for k in m0.state_dict())
    with pytest.raises(RuntimeError):
        dla.dla34(pretrained=True)
    # a stand-in for dla34-ba72cf86.pth: the trunk's own keys + the classifier, last entry = fc.bias [1000]
    trunk = dla.DLA()
    sd = {k: torch.full_like(v, 0.25) if v.is_floating_point() else v.clone() for k, v in trunk.state_dict().items()}
    sd['fc.weight'] = torch.zeros(1000, 512, 1, 1)
    sd['fc.bias'] = torch.zeros(1000)
    path = tmp_path / dla.PRETRAINED_FILE
    torch.save(sd, path)
    monkeypatch.setenv('CNUDA_DLA34_WEIGHTS', str(path))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        m1 = dla.build(num_classes=3, freeze_base=True)
    got = m1.state_dict()
    assert got['base.fc.weight'].shape == (1000, 512, 1, 1) and got['base.fc.bias'].shape == (1000,)
    assert torch.all(got['base.level2.tree1.conv1.weight'] == 0.25)
    assert all(not p.requires_grad for p in m1.base.parameters())
    assert m1.hm[2].bias.requires_grad
