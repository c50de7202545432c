"""Window functions: rectangular, hanning, hamming, povey, blackman

Counterpart of reference shennong/window.py:59-114; the coefficients come from the C ABI
(snf_window_function, in place of kaldi.feat.window.FeatureWindowFunction at window.py:107-114).
"""

import numpy as np

from shennong_amd import _abi, _backend

_ONES_AT_LENGTH_TWO = ('povey', 'blackman', 'hanning')  # cos(0) and cos(2 pi): both ends are 0 -> Kaldi
#                                                        would return zeros; the reference returns ones


def types():
    """Returns the supported window functions as a list"""
    return sorted(_abi.WINDOW_TYPES)


def window(length, type='povey', blackman_coeff=0.42):
    """Returns a window of the given `type` and `length` (float32; the two degenerate cases below are
    float64 ones, as in the reference, window.py:97-105)"""
    if int(length) <= 0:
        raise ValueError(f'length must be strictly positive but is {length}')
    if type not in types():
        raise ValueError(f'type must be in {types()} but is {type}')
    if length == 1 or (length == 2 and type in _ONES_AT_LENGTH_TWO):
        return np.ones((int(length),))
    # a 1 kHz "signal" makes the frame length in milliseconds the length in samples
    opts = _abi.default_frame_options()
    opts.samp_freq, opts.frame_length_ms = 1000, length
    opts.window_type, opts.blackman_coeff = _abi.WINDOW_TYPES[type], blackman_coeff
    return _backend.window_function(opts)
