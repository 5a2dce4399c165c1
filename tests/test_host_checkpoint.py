"""Host-side (no GPU): the checkpoint wire format of utils/helper.py:83-147 -- `{'epoch', 'state_dict'
[, 'optimizer', 'scheduler']}`, `module.` prefix stripping, shape-mismatch tolerance, start-epoch return
values -- and interchange with checkpoints written by the reference (plain torch modules / torch.optim.Adam)."""
import logging

import pytest
import torch
from torch import nn


def _small_model():
    from hip_runtime import nn as hnn
    torch.manual_seed(0)
    return nn.Sequential(hnn.Conv2d(3, 8, 3, padding=1, bias=True), hnn.BatchNorm2d(8), hnn.Conv2d(8, 4, 1))


def test_round_trip_and_epoch_semantics(tmp_path):
    from utils.helper import load_model, save_model
    a, b = _small_model(), _small_model()
    with torch.no_grad():
        for p in a.parameters():
            p.add_(1.0)
        a[1].running_mean.fill_(0.25)
        a[1].num_batches_tracked.fill_(7)
    path = tmp_path / 'model_best.pth'
    assert load_model(b, None, None, path) == 1                    # missing file: start at epoch 1 (helper.py:85-88)
    save_model(a, path, epoch=12)
    ckpt = torch.load(path, weights_only=False)
    assert sorted(ckpt) == ['epoch', 'state_dict'] and ckpt['epoch'] == 12
    assert list(ckpt['state_dict']) == list(a.state_dict())
    assert load_model(b, None, None, path, resume=False) == 0      # pretrained: epoch 0
    for (k, v), (k2, v2) in zip(a.state_dict().items(), b.state_dict().items()):
        assert k == k2 and torch.equal(v, v2), k
    assert load_model(b, None, None, path, resume=True) == 13      # resume: saved epoch + 1


def test_module_prefix_shape_mismatch_missing_and_extra_keys(tmp_path, caplog):
    from utils.helper import load_model
    m = _small_model()
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    sd = {'module.' + k: torch.full_like(v, 3) for k, v in ref.items()}         # nn.DataParallel checkpoint
    sd['module.2.weight'] = torch.zeros(5, 8, 1, 1)                              # other number of classes: skipped
    del sd['module.1.running_var']                                               # missing: keeps the model's value
    sd['module.base.fc.weight'] = torch.zeros(3)                                 # unknown: dropped (Q8: base.fc)
    path = tmp_path / 'dp.pth'
    torch.save({'epoch': 3, 'state_dict': sd}, path)
    with caplog.at_level(logging.INFO):
        assert load_model(m, None, None, path) == 0
    got = m.state_dict()
    assert torch.all(got['0.weight'] == 3) and torch.all(got['1.bias'] == 3)
    assert torch.equal(got['2.weight'], ref['2.weight'])                         # shape mismatch -> untouched
    assert torch.equal(got['1.running_var'], ref['1.running_var'])
    text = caplog.text
    assert 'shape mismatch' in text and 'no parameter 1.running_var' in text and 'drop parameter base.fc.weight' in text


def test_parallel_wrapper_saves_unprefixed_keys(tmp_path):
    from hip_runtime.parallel import DataParallel
    from utils.helper import load_model, save_model
    m = _small_model()
    dp = DataParallel(m)                                  # no process group: a plain wrapper
    path = tmp_path / 'w.pth'
    save_model(dp, path, epoch=1)
    assert list(torch.load(path, weights_only=False)['state_dict']) == list(m.state_dict())   # helper.py:134-137
    other = DataParallel(_small_model())
    with torch.no_grad():
        m[0].weight.fill_(2.0)
    save_model(dp, path, epoch=1)
    load_model(other, None, None, path)
    assert torch.all(other.module[0].weight == 2.0)


def test_optimizer_state_interchanges_with_torch_adam(tmp_path):
    """A checkpoint written by the reference holds torch.optim.Adam's state_dict; the fused Adam reads it and
    writes the same layout back."""
    from hip_runtime import optim
    from utils.helper import load_model, save_model
    ref_model = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.Conv2d(8, 4, 1))
    ref_opt = torch.optim.Adam(ref_model.parameters(), lr=5e-5, weight_decay=1e-4)
    ref_sched = torch.optim.lr_scheduler.MultiStepLR(ref_opt, milestones=[2, 4], gamma=0.1)
    ref_model(torch.randn(2, 3, 5, 5)).sum().backward()
    ref_opt.step()
    ref_sched.step()
    path = tmp_path / 'resume.pth'
    torch.save({'epoch': 5, 'state_dict': ref_model.state_dict(), 'optimizer': ref_opt.state_dict(),
                'scheduler': ref_sched.state_dict()}, path)
    m = _small_model()
    opt = optim.Adam(m.parameters(), lr=1.0)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[9], gamma=0.5)
    assert load_model(m, opt, sched, path, resume=True) == 6
    g = opt.param_groups[0]
    assert g['lr'] == ref_opt.param_groups[0]['lr'] and g['weight_decay'] == 1e-4
    for p, rp in zip(m.parameters(), ref_model.parameters()):
        assert torch.equal(p, rp)
        assert torch.equal(opt.state[p]['exp_avg'], ref_opt.state[rp]['exp_avg'])
        assert torch.equal(opt.state[p]['exp_avg_sq'], ref_opt.state[rp]['exp_avg_sq'])
        assert int(opt.state[p]['step']) == 1
    assert sched.state_dict()['milestones'] == ref_sched.state_dict()['milestones']
    out = tmp_path / 'again.pth'
    save_model(m, out, epoch=6, optimizer=opt, scheduler=sched)
    ck = torch.load(out, weights_only=False)
    assert sorted(ck) == ['epoch', 'optimizer', 'scheduler', 'state_dict']
    fresh = torch.optim.Adam(ref_model.parameters())
    fresh.load_state_dict(ck['optimizer'])                # and torch.optim.Adam reads what the fused Adam wrote
    for rp in ref_model.parameters():
        assert torch.equal(fresh.state[rp]['exp_avg'], ref_opt.state[rp]['exp_avg'])
