"""Closed-form known answers for the hot path: inputs whose features follow from the published
definition of the algorithm alone (window formula, DFT, triangular mel banks, DCT-II, delta
regression), with no Kaldi run and no code of this repository in the expected values.  Every case
is checked twice: against the CPU oracle (tests/test_oracle_pins.py) and against the HIP path
(tests/test_parity_gpu.py), so a transcription error shared by both sides would have to reproduce
these numbers as well.

A case is ``(name, make_processor, wave, check)``; ``check(out)`` asserts on the [frames, dims] matrix.
"""

import math

import numpy as np

from shennong_amd.processor import (FilterbankProcessor, MfccProcessor, PlpProcessor,
                                    SpectrogramProcessor)

LOG_EPS = math.log(np.finfo(np.float32).eps)   # log(FLT_EPSILON) = -15.9424...


def povey(n, length=400):
    """(0.5 - 0.5 cos(2 pi n / (L - 1))) ** 0.85, reference shennong/window.py:6-38"""
    return (0.5 - 0.5 * math.cos(2.0 * math.pi * n / (length - 1))) ** 0.85


# ---- 1. a unit impulse: flat spectrum at (A w[n0])^2 ---------------------------------------------------
def _impulse_case():
    amp, pos = 1000, 500
    wave = np.zeros(1200, dtype=np.int16)
    wave[pos] = amp

    def check(out):
        assert out.shape == (6, 257)
        for frame in range(6):
            n0 = pos - 160 * frame
            if 0 <= n0 < 400:
                want = math.log((amp * povey(n0)) ** 2)
                np.testing.assert_allclose(out[frame, 1:], want, rtol=0, atol=2e-5, err_msg=f'frame {frame}')
                # column 0 is the raw log-energy: log(A^2)
                assert abs(out[frame, 0] - math.log(amp ** 2)) < 1e-5
            else:
                np.testing.assert_allclose(out[frame], LOG_EPS, rtol=0, atol=2e-6)
    return ('impulse: flat spectrum', lambda: SpectrogramProcessor(
        dither=0, preemph_coeff=0, remove_dc_offset=False), wave, check)


# ---- 2. a constant: DC removal leaves exact zeros -> every bin sits on the floor --------------------
def _constant_cases():
    wave = np.full(2000, 1000, dtype=np.int16)

    def check_fbank(out):
        np.testing.assert_allclose(out, LOG_EPS, rtol=0, atol=2e-6)   # (one ulp of the log)

    def check_mfcc(out):
        # DCT-II of a constant vector: c0 = v sqrt(N), every other cepstrum 0; energy = floor
        assert out.shape[1] == 13
        np.testing.assert_allclose(out[:, 0], LOG_EPS * math.sqrt(23), rtol=2e-6)
        np.testing.assert_allclose(out[:, 1:], 0.0, atol=2e-4)

    def check_mfcc_energy(out):
        np.testing.assert_allclose(out[:, 0], LOG_EPS, rtol=0, atol=2e-6)

    return [('constant: fbank floor', lambda: FilterbankProcessor(num_bins=40, dither=0), wave, check_fbank),
            ('constant: spectrogram floor', lambda: SpectrogramProcessor(dither=0), wave, check_fbank),
            ('constant: mfcc c0', lambda: MfccProcessor(dither=0, use_energy=False), wave, check_mfcc),
            ('constant: mfcc energy', lambda: MfccProcessor(dither=0), wave, check_mfcc_energy)]


# ---- 3. a bin-centred sinusoid under a full-length rectangular window: one line at (A N / 2)^2 ------
def _sinusoid_case():
    amp, k0, n = 8000.0, 37, 512
    t = np.arange(4 * n)
    wave = np.round(amp * np.cos(2 * np.pi * k0 * t / n)).astype(np.int16)

    def check(out):
        assert out.shape[1] == 257 and out.shape[0] >= 3
        peak = math.log((amp * n / 2) ** 2)
        # rounding the samples to integers adds white noise of power N / 12 per bin: 1e-11 of the line
        np.testing.assert_allclose(out[:, k0], peak, rtol=0, atol=1e-4)
        others = np.delete(out[:, 1:], k0 - 1, axis=1)
        assert others.max() < peak - 20.0          # > 85 dB below the line
    return ('sinusoid: single spectral line', lambda: SpectrogramProcessor(
        dither=0, preemph_coeff=0, remove_dc_offset=False, window_type='rectangular',
        frame_length=0.032, frame_shift=0.016), wave, check)


# ---- 4. triangular mel banks are a partition of unity between the outer centre frequencies ----------
def mel_partition_check(first, weights, num_fft_bins=256):
    """first[b], weights[b] (support of bin b) -> asserts sum_b w_b(k) == 1 for every FFT bin between
    the centres of the first and the last filter"""
    total = np.zeros(num_fft_bins)
    for f, w in zip(first, weights):
        total[f:f + len(w)] += w
    peaks = [f + int(np.argmax(w)) for f, w in zip(first, weights)]
    inside = slice(peaks[0] + 1, peaks[-1])
    np.testing.assert_allclose(total[inside], 1.0, atol=2e-6)
    assert np.all(total[:peaks[0] + 1] <= 1.0 + 1e-6) and np.all(total[peaks[-1]:] <= 1.0 + 1e-6)


def _mel_partition_case():
    # a flat power spectrum P (impulse) through the LINEAR filterbank: energy_b = P sum_k w_b(k);
    # summed over the filters: P (number of FFT bins between the outer centres + the two half slopes)
    amp, pos = 1000, 200
    wave = np.zeros(400, dtype=np.int16)
    wave[pos] = amp

    def check(out):
        assert out.shape == (1, 40)
        power = (amp * povey(pos)) ** 2
        # mel(f) = 1127 ln(1 + f / 700); 42 equally spaced mel points between 20 Hz and 8 kHz
        mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
        lo, hi = mel(20.0), mel(8000.0)
        delta = (hi - lo) / 41
        total = 0.0
        for k in range(256):
            m = mel(k * 16000.0 / 512)
            x = (m - lo) / delta          # position in filter spacings: filters peak at 1 .. 40
            if 1.0 <= x <= 40.0:
                total += 1.0              # partition of unity
            elif 0.0 < x < 1.0:
                total += x                # rising slope of the first filter
            elif 40.0 < x < 41.0:
                total += 41.0 - x         # falling slope of the last filter
        np.testing.assert_allclose(out.sum(dtype=np.float64), power * total, rtol=2e-6)
    return ('mel banks: partition of unity', lambda: FilterbankProcessor(
        num_bins=40, dither=0, preemph_coeff=0, remove_dc_offset=False, use_log_fbank=False), wave, check)


# ---- 5. DCT-II orthonormality: a full-size MFCC (23 of 23, no lifter) keeps the norm of the log-mel ---
def parseval_check(fbank, mfcc):
    a = np.sum(np.asarray(fbank, np.float64) ** 2, axis=1)
    b = np.sum(np.asarray(mfcc, np.float64) ** 2, axis=1)
    np.testing.assert_allclose(b, a, rtol=2e-6)


# ---- 6. delta of a linear ramp -------------------------------------------------------------------------
def ramp_delta_check(deltas_fn):
    """deltas_fn(matrix[T, d], order, window) -> [T, d (order + 1)]; x[t] = a t: first delta = a in the
    interior, 0.8 a / 0.5 a at the clamped edges, second delta 0 in the interior"""
    slope = np.array([1.0, -2.5, 0.125], dtype=np.float32)
    x = (np.arange(40, dtype=np.float32)[:, None] * slope[None, :]).astype(np.float32)
    out = deltas_fn(x, 2, 2)
    assert out.shape == (40, 9)
    np.testing.assert_array_equal(out[:, :3], x)
    np.testing.assert_allclose(out[2:-2, 3:6], np.broadcast_to(slope, (36, 3)), rtol=1e-6)
    np.testing.assert_allclose(out[0, 3:6], 0.5 * slope, rtol=1e-6)
    np.testing.assert_allclose(out[1, 3:6], 0.8 * slope, rtol=1e-6)
    np.testing.assert_allclose(out[-1, 3:6], 0.5 * slope, rtol=1e-6)
    np.testing.assert_allclose(out[4:-4, 6:9], 0.0, atol=1e-5)


CASES = [_impulse_case()] + _constant_cases() + [_sinusoid_case(), _mel_partition_case()]
