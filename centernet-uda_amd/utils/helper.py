"""Host-side helpers with the reference's names (utils/helper.py): AverageMeter
and the checkpoint wire format (`{'epoch','state_dict'[,'optimizer','scheduler']}`,
`module.` prefix stripping, shape-mismatch tolerance, start-epoch return values;
utils/helper.py:83-147)."""
import logging
from pathlib import Path

import torch

log = logging.getLogger(__name__)


class AverageMeter:
    """Running mean weighted by batch size, with the reference's constructor and text form
    (utils/helper.py:13-35; the driver builds `AverageMeter(name=k)`, train.py:164,182,243)."""

    def __init__(self, name, fmt=':f'):
        self.name = name
        self.fmt = fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        if self.count > 0:                      # the reference divides unguarded; update(x, n=0) would raise there
            self.avg = self.sum / self.count

    def __str__(self):
        fmtstr = '{name} {val' + self.fmt + '} ({avg' + self.fmt + '})'
        return fmtstr.format(**self.__dict__)


def _unwrap(model):
    return model.module if hasattr(model, 'module') and isinstance(model.module, torch.nn.Module) else model


def load_model(model, optimizer, scheduler, path, resume=False):
    """Returns the start epoch: 1 if the file is missing, 0 for `pretrained`,
    saved epoch + 1 for `resume` (utils/helper.py:85-90,128)."""
    path = Path(path)
    if not path.exists():
        log.warning("Model path %s does not exists!", path)
        return 1
    checkpoint = torch.load(path, map_location='cpu', weights_only=False)
    epoch = checkpoint["epoch"] if resume else 0
    incoming = {}
    for k, v in checkpoint['state_dict'].items():
        # DataParallel checkpoints carry a 'module.' prefix ('module_list...' is a real name)
        incoming[k[7:] if k.startswith('module') and not k.startswith('module_list') else k] = v
    target = _unwrap(model)
    own = target.state_dict()
    for k in list(incoming):
        if k not in own:
            log.info("drop parameter %s", k)
        elif incoming[k].shape != own[k].shape:
            log.warning("skip parameter %s because of shape mismatch", k)
            incoming[k] = own[k]
    for k in own:
        if k not in incoming:
            log.warning("no parameter %s available", k)
            incoming[k] = own[k]
    target.load_state_dict(incoming, strict=False)
    if resume and optimizer is not None and 'optimizer' in checkpoint:
        optimizer.load_state_dict(checkpoint['optimizer'])
        if scheduler is not None and 'scheduler' in checkpoint:
            scheduler.load_state_dict(checkpoint['scheduler'])
    return (epoch + 1) if resume else epoch


def save_model(model, path, epoch, optimizer=None, scheduler=None):
    data = {'epoch': epoch, 'state_dict': _unwrap(model).state_dict()}
    if optimizer is not None:
        data['optimizer'] = optimizer.state_dict()
        if scheduler is not None:
            data['scheduler'] = scheduler.state_dict()
    torch.save(data, path)
