"""Features extraction processors, loaded on first use

``from shennong_amd.processor import MfccProcessor`` works as in the reference package; the class
is imported from its module when it is first asked for.
"""

import importlib

_HOME = {
    'EnergyProcessor': 'energy',
    'FilterbankProcessor': 'filterbank',
    'MfccProcessor': 'mfcc',
    'PlpProcessor': 'plp',
    'SpectrogramProcessor': 'spectrogram',
    'KaldiPitchProcessor': 'pitch_kaldi',
    'KaldiPitchPostProcessor': 'pitch_kaldi',
}
__all__ = sorted(_HOME)


def __getattr__(name):
    if name in _HOME:
        return getattr(importlib.import_module(f'{__name__}.{_HOME[name]}'), name)
    raise AttributeError(f'module {__name__!r} has no attribute {name!r}')


def __dir__():
    return __all__
