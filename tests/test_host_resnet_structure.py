"""Host-side (no GPU) checks of the product's CenterResNet module tree against what the reference's class
produces (tests/golden/resnet18_fwd.npz): state_dict names and shapes are the checkpoint wire format."""
import ast
import inspect

import pytest
import torch


def test_state_dict_names_shapes_and_order_match_reference(golden):
    from backends import resnet
    g = golden('resnet18_fwd')
    want = dict(ast.literal_eval(str(g['shapes_json'])))
    model = resnet.build(18, num_classes=6, pretrained=False)
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert list(got) == [str(s) for s in g['state_names']]            # same registration order (sorted heads)
    for k in want:
        assert got[k] == tuple(want[k]), k
    assert [n for n, _ in model.named_parameters()] == [str(s) for s in g['param_names']]
    assert sum(p.numel() for p in model.parameters()) == int(g['n_params'])
    assert model.down_ratio == 4 and model.rotated_boxes is False
    assert list(model.heads) == ['hm', 'wh', 'reg']                   # emission order (resnet.py:58)


@pytest.mark.parametrize('n,width', [(34, 512), (50, 2048)])
def test_deeper_variants_build(n, width):
    from backends import resnet
    m = resnet.build(n, num_classes=3, pretrained=False, rotated_boxes=True, freeze_base=True)
    sd = m.state_dict()
    assert sd['deconv_layers.0.weight'].shape == (width, 256, 4, 4)
    assert sd['wh.2.weight'].shape == (3, 64, 1, 1)
    assert not any(p.requires_grad for p in m.base.parameters())
    assert all(p.requires_grad for p in m.deconv_layers.parameters())


def test_build_signature_errors_and_no_cpu_fallback():
    from backends import resnet
    sig = inspect.signature(resnet.build)
    assert list(sig.parameters) == ['num_layers', 'num_classes', 'num_keypoints', 'pretrained', 'freeze_base',
                                    'rotated_boxes']
    assert sig.parameters['pretrained'].default is True
    with pytest.raises(AssertionError):
        resnet.build(19, num_classes=2, pretrained=False)
    with pytest.raises(RuntimeError):                                  # no silent random init for pretrained=True
        resnet.build(18, num_classes=2)
    m = resnet.build(18, num_classes=2, pretrained=False)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64))                                   # CPU tensors are refused
