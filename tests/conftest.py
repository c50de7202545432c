"""Shared fixtures.  GPU tests are marked ``@pytest.mark.gpu`` and call the HIP path through the
C ABI; everything else runs on CPU (oracle pins, host logic, ABI symbol checks)."""

import os
import sys

import numpy as np
import pytest
import scipy.io.wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def wav_file():
    return os.path.join(GOLDEN, 'test.wav')


@pytest.fixture(scope='session')
def audio(wav_file):
    """The reference's test clip: 16 kHz int16 mono, 22 713 samples -> 140 frames
    (reference test/conftest.py:28-35)"""
    from shennong_amd import Audio
    return Audio.load(wav_file)


@pytest.fixture(scope='session')
def audio_8k():
    from shennong_amd import Audio
    return Audio.load(os.path.join(GOLDEN, 'test.8k.wav'))


@pytest.fixture(scope='session')
def wave(wav_file):
    return scipy.io.wavfile.read(wav_file)[1]


@pytest.fixture(scope='session')
def synth_waves():
    from shennong_amd import synth
    return synth.ragged_utterances(1000, 6, min_s=0.3, max_s=1.2)


@pytest.fixture(scope='session')
def _gpu_backend():
    from shennong_amd import _backend
    if _backend.device_count() < 1:
        pytest.skip('no HIP device visible')
    return _backend


@pytest.fixture()
def gpu(_gpu_backend):
    """Skips when no device is visible; on the GPU box a missing library is an error.  Every GPU
    test starts from LDS filled with NaN bit patterns: LDS is not cleared between kernels, and a
    kernel that multiplies a word it never wrote by a zero weight is correct only as long as the
    previous tenant of that CU left something finite there (this bit the last mel bin once)."""
    _gpu_backend.check(_gpu_backend.lib().snf_debug_fill_lds(0xFFFFFFFF))
    return _gpu_backend


def assert_close(got, want, rtol=1e-4, atol=1e-4, what=''):
    """Parity tolerance of the float32 features: 1e-4 relative (BASELINE.json north_star) plus 1e-4
    absolute for coefficients that cross zero (an MFCC is a signed sum of 23 log energies of
    magnitude ~15: float32 round-off of either side reaches 6e-5 there; measured worst cases:
    profiles/r02_parity_errors.txt).  The measured worst case of every call is appended to the file named by
    SNF_PARITY_LOG (tools/parity_errors.py summarises it; the committed summary is in
    profiles/r02_parity_errors.txt)."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert got.dtype == want.dtype == np.float32, (what, got.dtype, want.dtype)
    log = os.environ.get('SNF_PARITY_LOG')
    if log and got.size:
        import json
        err = np.abs(got.astype(np.float64) - want.astype(np.float64))
        excess = err - rtol * np.abs(want.astype(np.float64))
        with open(log, 'a') as fh:
            fh.write(json.dumps({
                'test': os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0], 'what': str(what),
                'max_abs': float(err.max()), 'max_rel': float((err / np.maximum(np.abs(want), 1e-30)).max()),
                'needed_atol_at_rtol': float(max(excess.max(), 0.0)), 'rtol': rtol, 'atol': atol,
                'size': int(got.size)}) + '\n')
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=what)
