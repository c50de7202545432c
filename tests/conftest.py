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
    previous tenant of that CU left something finite there (this bit the last mel bin once).  The same
    goes for HBM now that freed device buffers are pooled and reused."""
    _gpu_backend.check(_gpu_backend.lib().snf_debug_fill_lds(0xFFFFFFFF))
    # (and every device buffer that comes back from the pool is handed out full of NaN bit patterns)
    _gpu_backend.DEVICE_POOL.poison = True
    return _gpu_backend


# Absolute tolerance per feature family on top of the 1e-4 relative tolerance of BASELINE.json's north_star:
# TWICE the largest excess over rtol * |want| measured over the GPU suite (profiles/r03_parity_errors.txt;
# tools/parity_errors.py).  A log-domain coefficient that crosses zero has no meaningful relative error: an
# MFCC is a signed sum of 23 log energies of magnitude ~15 (float32 round-off of either side reaches 6e-5
# there), a spectrogram bin at a spectral null is the log of a difference of large numbers.
# measured needs: fbank 0 (the relative term covers it: log energies sit far from zero), MFCC 7.5e-5, PLP
# 2.5e-6, spectrogram 6.4e-4, delta 1.1e-6 (at rtol 1e-5), pitch post-processing 3.3e-6
# Round 4: the spectrogram no longer gets a wide absolute term for its spectral nulls.  A bin more
# than 60 dB below the strongest bin of its frame is compared in the LINEAR domain, where the error of a
# float32 transform lives - |P_got - P_want| <= 1e-4 P_want + 1e-9 P_max(frame) -, every other bin at the
# north_star's 1e-4 relative + 1e-4 absolute in the log domain (`_spectrogram_close`).
# Round 5: every parity assertion of the mel families now uses the north_star's 1e-4
# relative term (14 of them carried 2e-4).  Measured over the whole GPU suite at rtol 1e-4
# (profiles/r05_parity_errors.txt): fbank needs no absolute term, MFCC 8.1e-5, PLP 5.8e-6 (one case: VTLN warp
# 1.2, a cepstrum near zero - the Durbin recursion amplifies the float32 round-off of the 23 mel energies), hence
# PLP's absolute term goes from 5e-6 to 1.2e-5 (twice the measured need, the rule above) instead of its
# relative term staying at 2e-4.
FAMILY_ATOL = {'fbank': 1e-5, 'mfcc': 1.6e-4, 'plp': 1.2e-5, 'spectrogram': 1e-4, 'delta': 2.5e-6,
               'pitch_post': 7e-6, None: 1e-4}
NULL_DB = 60.0


def family_of(what):
    text = str(what).lower()
    for key, names in (('spectrogram', ('spectrogram',)), ('mfcc', ('mfcc',)), ('plp', ('plp',)),
                       ('fbank', ('fbank', 'filterbank')), ('delta', ('delta',))):
        if any(n in text for n in names):
            return key
    return None


def _spectrogram_close(got, want, rtol, atol, what):
    """log power spectra [frames, bins] (column 0 may hold the log energy: it is far above any null)"""
    g, w = got.astype(np.float64), want.astype(np.float64)
    peak = w.max(axis=1, keepdims=True)
    null = w < peak - NULL_DB / 10.0 * np.log(10.0)
    err = np.abs(g - w)
    loud_bad = ~null & (err > atol + rtol * np.abs(w))
    pg, pw = np.exp(g - peak), np.exp(w - peak)            # powers relative to the frame's peak
    null_bad = null & (np.abs(pg - pw) > 1e-4 * pw + 1e-9)
    assert not loud_bad.any() and not null_bad.any(), (
        '%s: %d bins above the -%g dB line outside rtol %g + atol %g (worst %.3g), %d spectral nulls outside '
        'the linear bound' % (what, int(loud_bad.sum()), NULL_DB, rtol, atol,
                              float(np.where(~null, err, 0).max()), int(null_bad.sum())))
    return float(null.mean())


def assert_close(got, want, rtol=1e-4, atol=None, what='', family=None):
    """Parity tolerance of the float32 features: 1e-4 relative (BASELINE.json north_star) plus the
    absolute tolerance of the feature family (FAMILY_ATOL; `family`, or read from `what`).  The measured
    worst case of every call is appended to the file named by SNF_PARITY_LOG (tools/parity_errors.py
    summarises it; the committed summary is profiles/r03_parity_errors.txt)."""
    family = family or family_of(what)
    if atol is None:
        atol = FAMILY_ATOL[family]
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
                'family': family or 'other',
                'max_abs': float(err.max()), 'max_rel': float((err / np.maximum(np.abs(want), 1e-30)).max()),
                'needed_atol_at_rtol': float(max(excess.max(), 0.0)), 'rtol': rtol, 'atol': atol,
                'inside_1e-4_rel': int((err <= 1e-4 * np.abs(want.astype(np.float64))).sum()),
                'size': int(got.size)}) + '\n')
    if family == 'spectrogram' and got.ndim == 2 and got.shape[0] and got.shape[1] > 64:
        _spectrogram_close(got, want, rtol, atol, what)
        return
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol, err_msg=what)
