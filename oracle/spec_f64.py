"""Float64 numpy restatement of the fbank / MFCC / spectrogram recipe.  TEST INFRASTRUCTURE ONLY.

An independent second implementation (vectorised, numpy rfft, everything in double) used to catch
transcription errors in oracle/kaldi_oracle.c and to tell float32 round-off from real divergence: the
C oracle and the HIP path must both sit within ~1e-5 of it.  Follows the same sources as the oracle
([KALDI-UPSTREAM] feature-window.cc / feature-fbank.cc / feature-mfcc.cc / mel-computations.cc; the
in-tree restatement of the per-frame recipe is reference shennong/processor/plp.py:171-260).
"""

import numpy as np

EPS32 = float(np.finfo(np.float32).eps)


def frame_geometry(sample_rate, frame_shift, frame_length, round_pow2=True):
    # float32 option fields, double arithmetic (Kaldi: samp_freq * 0.001 * frame_shift_ms)
    shift = int(np.float64(np.float32(sample_rate)) * 0.001
                * np.float64(np.float32(frame_shift * 1000.0)))
    length = int(np.float64(np.float32(sample_rate)) * 0.001
                 * np.float64(np.float32(frame_length * 1000.0)))
    padded = length
    if round_pow2:
        padded = 1
        while padded < length:
            padded *= 2
    return shift, length, padded


def num_frames(n, shift, length, snip_edges=True):
    if snip_edges:
        return 0 if n < length else 1 + (n - length) // shift
    return (n + shift // 2) // shift


def window_function(length, kind='povey', blackman_coeff=0.42):
    a = 2 * np.pi / (length - 1)
    i = np.arange(length, dtype=np.float64)
    if kind == 'hanning':
        return 0.5 - 0.5 * np.cos(a * i)
    if kind == 'hamming':
        return 0.54 - 0.46 * np.cos(a * i)
    if kind == 'povey':
        return (0.5 - 0.5 * np.cos(a * i)) ** 0.85
    if kind == 'rectangular':
        return np.ones(length)
    if kind == 'blackman':
        return (blackman_coeff - 0.5 * np.cos(a * i)
                + (0.5 - blackman_coeff) * np.cos(2 * a * i))
    raise ValueError(kind)


def extract_frames(wave, shift, length, snip_edges=True):
    """[nframes, length] float64 with Kaldi's reflection at the edges"""
    n = len(wave)
    nf = num_frames(n, shift, length, snip_edges)
    idx = np.arange(nf)[:, None] * shift + np.arange(length)[None, :]
    if not snip_edges:
        idx = idx + shift // 2 - length // 2
        for _ in range(4):
            idx = np.where(idx < 0, -idx - 1, idx)
            idx = np.where(idx >= n, 2 * n - 1 - idx, idx)
    return np.asarray(wave, dtype=np.float64)[idx]


def mel_scale(f):
    return 1127.0 * np.log(1.0 + f / 700.0)


def mel_banks(num_bins, sample_rate, padded, low_freq=20.0, high_freq=0.0):
    """Dense [num_bins, padded/2] triangular weights (no VTLN), float64"""
    nyquist = 0.5 * sample_rate
    high = high_freq if high_freq > 0 else nyquist + high_freq
    mlow, mhigh = mel_scale(low_freq), mel_scale(high)
    delta = (mhigh - mlow) / (num_bins + 1)
    nfft = padded // 2
    mel = mel_scale(np.arange(nfft) * sample_rate / padded)
    w = np.zeros((num_bins, nfft))
    for b in range(num_bins):
        left, center, right = mlow + b * delta, mlow + (b + 1) * delta, mlow + (b + 2) * delta
        up = (mel - left) / (center - left)
        down = (right - mel) / (right - center)
        inside = (mel > left) & (mel < right)
        w[b] = np.where(inside, np.where(mel <= center, up, down), 0.0)
    return w


def features(wave, kind='fbank', sample_rate=16000, frame_shift=0.01, frame_length=0.025,
             preemph=0.97, remove_dc=True, window='povey', snip_edges=True, num_bins=23,
             low_freq=20.0, high_freq=0.0, use_energy=None, raw_energy=True, num_ceps=13,
             cepstral_lifter=22.0, use_log_fbank=True, use_power=True):
    shift, length, padded = frame_geometry(sample_rate, frame_shift, frame_length)
    x = extract_frames(wave, shift, length, snip_edges)
    if remove_dc:
        x = x - x.mean(axis=1, keepdims=True)
    raw_log_energy = np.log(np.maximum((x * x).sum(axis=1), EPS32))
    if preemph != 0:
        y = x.copy()
        y[:, 1:] = x[:, 1:] - preemph * x[:, :-1]
        y[:, 0] = x[:, 0] - preemph * x[:, 0]
        x = y
    x = x * window_function(length, window)[None, :]
    post_log_energy = np.log(np.maximum((x * x).sum(axis=1), EPS32))
    log_energy = raw_log_energy if raw_energy else post_log_energy
    spec = np.fft.rfft(x, n=padded, axis=1)
    power = spec.real ** 2 + spec.imag ** 2
    if kind == 'spectrogram':
        out = np.log(np.maximum(power, EPS32))
        out[:, 0] = log_energy
        return out
    w = mel_banks(num_bins, sample_rate, padded, low_freq, high_freq)
    p = power[:, :padded // 2]
    if kind == 'fbank' and not use_power:
        p = np.sqrt(p)
    mel = p @ w.T
    if kind == 'fbank':
        out = np.log(np.maximum(mel, EPS32)) if use_log_fbank else mel
        if use_energy:
            out = np.hstack((log_energy[:, None], out))
        return out
    logmel = np.log(np.maximum(mel, EPS32))
    n = np.arange(num_bins)
    dct = np.sqrt(2.0 / num_bins) * np.cos(
        np.pi / num_bins * (n[None, :] + 0.5) * np.arange(num_ceps)[:, None])
    dct[0] = np.sqrt(1.0 / num_bins)
    out = logmel @ dct.T
    if cepstral_lifter:
        out = out * (1 + 0.5 * cepstral_lifter * np.sin(
            np.pi * np.arange(num_ceps) / cepstral_lifter))[None, :]
    if use_energy is None or use_energy:
        out[:, 0] = log_energy
    return out
