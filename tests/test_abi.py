"""The C-ABI library loads on a CPU-only box and exports every symbol include/shennong_amd.h
declares; host-only entry points work; compute entry points fail loudly without a GPU."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from shennong_amd import _abi, _backend


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'shennong_amd.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(snf_[a-z0-9_]+)\s*\(', text)))


def test_exports_match_header():
    lib = _backend.lib()
    names = _header_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_backend.EXPORTS) == names


def test_struct_layout_matches_header():
    """ctypes mirror and C struct agree on size (checked through a C probe compiled on the fly
    would need a compiler at test time; sizes are fixed by the header's field list instead)"""
    assert C.sizeof(_abi.FrameOptions) == 40
    assert C.sizeof(_abi.MelOptions) == 20
    assert C.sizeof(_abi.PitchOptions) == 64
    assert C.sizeof(_abi.PitchPostOptions) == 52
    assert C.sizeof(_abi.Options) % 8 == 0


def test_host_entry_points():
    fo = _abi.default_frame_options()
    assert _backend.window_size(fo) == 400
    assert _backend.window_shift(fo) == 160
    assert _backend.padded_window_size(fo) == 512
    assert _backend.num_frames(fo, 22713) == 140
    assert _backend.first_sample_of_frame(fo, 3) == 480
    fo.snip_edges = 0
    assert _backend.num_frames(fo, 22713) == 142
    assert _backend.first_sample_of_frame(fo, 0) == -120
    assert _backend.pitch_num_frames(_abi.default_pitch_options(), 22713) == 140
    assert _backend.lib().snf_version().startswith(b'shennong_amd')


def test_host_tables_equal_oracle():
    """The product's window equals the independently written oracle's bit for bit"""
    from oracle import oracle as orc
    for kind in _abi.WINDOW_TYPES.values():
        for length_ms in (25, 5, 64):
            fo = _abi.default_frame_options()
            fo.window_type = kind
            fo.frame_length_ms = length_ms
            assert np.array_equal(_backend.window_function(fo), orc.window_function(fo))
    for n in (0, 399, 400, 401, 22713, 48000, 10**9):
        for snip in (0, 1):
            fo = _abi.default_frame_options()
            fo.snip_edges = snip
            assert _backend.num_frames(fo, n) == orc.num_frames(fo, n)
    po = _abi.default_pitch_options()
    for n in (0, 100, 1599, 1600, 22713, 48000, 96001):
        assert _backend.pitch_num_frames(po, n) == orc.pitch_num_frames(po, n)


def test_no_silent_cpu_fallback():
    """Without a GPU a plan cannot be created: the product never routes through the oracle"""
    if _backend.device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(RuntimeError) as err:
        _backend.Plan(_abi.default_options(_abi.KIND_FBANK))
    assert 'no HIP device' in str(err.value)
    import shennong_amd
    import sys
    assert not any(m.startswith('oracle') for m in sys.modules
                   if sys.modules[m] is not None and
                   getattr(sys.modules[m], '__file__', None) and
                   'shennong_amd' in (getattr(sys.modules[m], '__file__') or ''))
    src = os.path.join(ROOT, 'shennong_amd')
    for dirpath, _, files in os.walk(src):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.h')):
                text = open(os.path.join(dirpath, f), errors='replace').read()
                assert 'liboracle' not in text and 'from oracle' not in text, f
