"""Host-side (no GPU) checks of the product's CenterMobileNetV2 module tree against what the reference's class
produces (tests/golden/mbv2_*.npz): state_dict names, order and shapes are the checkpoint wire format."""
import ast
import inspect

import pytest
import torch


@pytest.mark.parametrize('tag,flags', [('dcn', dict(use_dcn=True, use_skip=False)), ('skip', dict(use_dcn=False, use_skip=True))])
def test_state_dict_names_shapes_and_order_match_reference(golden, tag, flags):
    from backends import mobilenetv2
    g = golden('mbv2_' + tag)
    want = dict(ast.literal_eval(str(g['shapes_json'])))
    model = mobilenetv2.build(num_classes=6, pretrained=False, **flags)
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert list(got) == [str(s) for s in g['state_names']]            # same registration order
    for k in want:
        assert got[k] == tuple(want[k]), k
    assert [n for n, _ in model.named_parameters()] == [str(s) for s in g['param_names']]
    assert sum(p.numel() for p in model.parameters()) == int(g['n_params'])
    assert model.down_ratio == 4 and model.rotated_boxes is False
    assert list(model.heads) == ['hm', 'wh', 'reg']


def test_build_signature_variants_and_no_cpu_fallback():
    from backends import mobilenetv2
    sig = inspect.signature(mobilenetv2.build)
    assert list(sig.parameters) == ['num_classes', 'num_keypoints', 'pretrained', 'freeze_base', 'use_dcn', 'use_skip',
                                    'rotated_boxes']
    assert sig.parameters['pretrained'].default is True
    with pytest.raises(RuntimeError):                                  # no silent random init for pretrained=True
        mobilenetv2.build(num_classes=2)
    m = mobilenetv2.build(num_classes=3, num_keypoints=4, pretrained=False, freeze_base=True, rotated_boxes=True)
    sd = m.state_dict()
    assert sd['base.18.0.weight'].shape == (1280, 320, 1, 1) and sd['base.2.conv.1.0.weight'].shape == (96, 1, 3, 3)
    assert sd['deconv_layers.0.weight'].shape == (1280, 256, 4, 4)
    assert sd['wh.2.weight'].shape == (3, 64, 1, 1) and sd['kps.2.weight'].shape == (8, 64, 1, 1)
    assert list(m.heads) == ['hm', 'wh', 'reg', 'kps']
    assert not any(p.requires_grad for p in m.base.parameters())
    assert all(p.requires_grad for p in m.deconv_layers.parameters())
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 64, 64))                                   # CPU tensors are refused
