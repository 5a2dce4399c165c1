"""UDA step plugins.  The driver resolves them by class name from the config (`uda.<ClassName>`, train.py:104-106);
the classes are imported on first use.  FDA (uda/fda.py) is not part of this build: it needs `torch.rfft`, which
current PyTorch no longer has."""
import importlib

_PLUGINS = {
    'Model': 'uda.base',
    'EntropyMinimization': 'uda.entropy_minimization',
    'MaxSquaresMinimization': 'uda.max_squares_minimization',
    'AdversarialEntropyMinimization': 'uda.adversarial_entropy_minimization',
}
__all__ = sorted(_PLUGINS)


def __getattr__(name):
    if name in _PLUGINS:
        cls = getattr(importlib.import_module(_PLUGINS[name]), name)
        globals()[name] = cls
        return cls
    if name == 'FDA':
        raise AttributeError("uda.FDA is outside this build (uda/fda.py uses torch.rfft, removed from PyTorch)")
    raise AttributeError("module 'uda' has no attribute %r" % name)


def __dir__():
    return sorted(list(globals()) + list(_PLUGINS))
