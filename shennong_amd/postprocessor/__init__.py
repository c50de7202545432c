"""Features post-processors, loaded on first use (see shennong_amd.processor)"""

import importlib

_HOME = {
    'CmvnPostProcessor': 'cmvn',
    'SlidingWindowCmvnPostProcessor': 'cmvn',
    'apply_cmvn': 'cmvn',
    'DeltaPostProcessor': 'delta',
    'VadPostProcessor': 'vad',
}
__all__ = sorted(_HOME)


def __getattr__(name):
    if name in _HOME:
        return getattr(importlib.import_module(f'{__name__}.{_HOME[name]}'), name)
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')


def __dir__():
    return __all__
