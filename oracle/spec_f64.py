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
             cepstral_lifter=22.0, use_log_fbank=True, use_power=True, energy_floor=0.0, htk_compat=False,
             blackman_coeff=0.42, round_pow2=True):
    shift, length, padded = frame_geometry(sample_rate, frame_shift, frame_length, round_pow2)
    x = extract_frames(wave, shift, length, snip_edges)
    if remove_dc:
        x = x - x.mean(axis=1, keepdims=True)
    raw_log_energy = np.log(np.maximum((x * x).sum(axis=1), EPS32))
    if preemph != 0:
        y = x.copy()
        y[:, 1:] = x[:, 1:] - preemph * x[:, :-1]
        y[:, 0] = x[:, 0] - preemph * x[:, 0]
        x = y
    x = x * window_function(length, window, blackman_coeff)[None, :]
    post_log_energy = np.log(np.maximum((x * x).sum(axis=1), EPS32))
    log_energy = raw_log_energy if raw_energy else post_log_energy
    if energy_floor > 0:   # the log-energy column is never below log(energy_floor)
        log_energy = np.maximum(log_energy, np.log(energy_floor))
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
        if use_energy:   # energy first; last with htk_compat
            out = np.hstack((out, log_energy[:, None])) if htk_compat else np.hstack((log_energy[:, None], out))
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
    if htk_compat:   # C0 (or the energy) moves to the end; a C0 that is not the energy is scaled by sqrt(2)
        c0 = out[:, :1] if (use_energy is None or use_energy) else out[:, :1] * np.sqrt(2.0)
        out = np.hstack((out[:, 1:], c0))
    return out


# ---------------------------------------------------------------------------------------------------------
# Kaldi pitch tracker in float64 (round 3).  An independent restatement of [KALDI-UPSTREAM] pitch-functions.cc
# / resample.cc - LinearResample, ComputeCorrelation / ComputeNccf with the ballast, ArbitraryResample to the
# log-spaced lags, and the Viterbi recursion as a FULL search over all previous states - with every sum in
# double and numpy's own summation orders.  oracle/kaldi_oracle.c fixes float32 summation orders that a GPU
# wavefront reproduces bit for bit (chain_dot / tree16), so "GPU == oracle" says nothing about how far those
# orders sit from the exact arithmetic: this restatement does (tests/test_oracle_pins.py reports
# max |oracle - f64| of the NCCF and the fraction of frames whose Viterbi state differs).
# Reached by the reference at shennong/processor/pitch_kaldi.py:296-299.
# ---------------------------------------------------------------------------------------------------------
def _filter_func(t, cutoff, num_zeros):
    """resample.cc FilterFunc: Hann-windowed sinc, t in seconds"""
    t = np.asarray(t, dtype=np.float64)
    window = np.where(np.abs(t) < num_zeros / (2.0 * cutoff),
                      0.5 * (1 + np.cos(2 * np.pi * cutoff / num_zeros * t)), 0.0)
    with np.errstate(divide='ignore', invalid='ignore'):
        filt = np.where(t != 0, np.sin(2 * np.pi * cutoff * t) / (np.pi * t), 2.0 * cutoff)
    return filt * window


def linear_resample(wave, rate_in=16000, rate_out=4000, cutoff=1000.0, num_zeros=1, flush=True):
    """LinearResample of the whole signal (samples outside it are zero); returns the output samples"""
    from math import gcd
    wave = np.asarray(wave, dtype=np.float64)
    n = wave.shape[0]
    base = gcd(rate_in, rate_out)
    in_unit, out_unit = rate_in // base, rate_out // base
    tick = rate_in // base * rate_out
    interval = n * (tick // rate_in)
    if not flush:
        # (an index decision, not arithmetic: Kaldi floors the FLOAT32 product of a float32 width and the tick rate)
        interval -= int(np.floor(np.float32(np.float32(num_zeros / (2.0 * cutoff)) * np.float32(tick))))
    if interval <= 0:
        return np.zeros(0)
    per_out = tick // rate_out
    last = interval // per_out
    if last * per_out == interval:
        last -= 1
    n_out = last + 1
    width = num_zeros / (2.0 * cutoff)
    out = np.zeros(n_out)
    k = np.arange(n_out)
    for i in range(out_unit):
        t_out = i / rate_out
        lo = int(np.ceil((t_out - width) * rate_in))
        hi = int(np.floor((t_out + width) * rate_in))
        taps = np.arange(lo, hi + 1)
        w = _filter_func(taps / rate_in - t_out, cutoff, num_zeros) / rate_in
        sel = k[k % out_unit == i]
        idx = (sel // out_unit)[:, None] * in_unit + taps[None, :]
        ok = (idx >= 0) & (idx < n)
        out[sel] = (np.where(ok, wave[np.clip(idx, 0, n - 1)], 0.0) * w[None, :]).sum(axis=1)
    return out


def pitch(wave, samp_freq=16000, frame_shift_ms=10.0, frame_length_ms=25.0, min_f0=50.0, max_f0=400.0,
          soft_min_f0=10.0, penalty_factor=0.1, lowpass_cutoff=1000.0, resample_freq=4000,
          delta_pitch=0.005, nccf_ballast=7000.0, lowpass_filter_width=1, upsample_filter_width=5,
          recompute_frame=500):
    """-> dict(out [T, 2] (POV NCCF, pitch), nccf [T, S] (resampled, with ballast), states [T], lags [S])
    snip_edges = True, the offline single-chunk call (frames before / after the resampler flush see the
    signal statistics of their phase; RecomputeBacktraces as in kaldi_oracle.c)."""
    samp_freq, resample_freq = int(samp_freq), int(resample_freq)
    down1 = linear_resample(wave, samp_freq, resample_freq, lowpass_cutoff, lowpass_filter_width, flush=False)
    down = linear_resample(wave, samp_freq, resample_freq, lowpass_cutoff, lowpass_filter_width, flush=True)
    n1, n2 = down1.shape[0], down.shape[0]
    outer_min = 1.0 / max_f0 - upsample_filter_width / (2.0 * resample_freq)
    outer_max = 1.0 / min_f0 + upsample_filter_width / (2.0 * resample_freq)
    first_lag, last_lag = int(np.ceil(resample_freq * outer_min)), int(np.floor(resample_freq * outer_max))
    L = last_lag + 1 - first_lag
    W = int(resample_freq * frame_length_ms / 1000.0)
    shift = int(resample_freq * frame_shift_ms / 1000.0)
    full = W + last_lag
    T1 = 0 if n1 < full else (n1 - full) // shift + 1
    T = 0 if n2 < W else (n2 - W) // shift + 1
    T1 = min(T1, T)
    # SelectLags runs in BaseFloat in Kaldi; the float32 lags are part of the algorithm's definition
    lags = []
    lag = np.float32(1.0 / max_f0)
    while lag <= np.float32(1.0 / min_f0):
        lags.append(float(lag))
        lag = np.float32(float(lag) * (1.0 + delta_pitch))
    lags = np.array(lags)
    S = lags.shape[0]
    if T <= 0:
        return dict(out=np.zeros((0, 2)), nccf=np.zeros((0, S)), states=np.zeros(0, int), lags=lags)
    ms1 = (down[:n1] ** 2).sum() / n1 - (down[:n1].sum() / n1) ** 2 if n1 > 0 else 0.0
    ms2 = (down ** 2).sum() / n2 - (down.sum() / n2) ** 2
    # ArbitraryResample: weights of the (<= 2 * width + 1) integer lags around every log-spaced lag
    up_cut = 0.5 * resample_freq
    fw = upsample_filter_width / (2.0 * up_cut)
    tq = lags - first_lag / resample_freq
    ar = np.zeros((S, L))
    for s in range(S):
        lo = max(int(np.ceil(resample_freq * (tq[s] - fw))), 0)
        hi = min(int(np.floor(resample_freq * (tq[s] + fw))), L - 1)
        j = np.arange(lo, hi + 1)
        ar[s, j] = _filter_func(tq[s] - j / resample_freq, up_cut, upsample_filter_width) / resample_freq
    padded = np.concatenate([down, np.zeros(full + shift)])
    frames = padded[np.arange(T)[:, None] * shift + np.arange(full)[None, :]]
    frames = frames - frames[:, :W].mean(axis=1, keepdims=True)
    e1 = (frames[:, :W] ** 2).sum(axis=1)
    inner = np.zeros((T, L))
    norm = np.zeros((T, L))
    for li, lg in enumerate(range(first_lag, last_lag + 1)):
        seg = frames[:, lg:lg + W]
        inner[:, li] = (frames[:, :W] * seg).sum(axis=1)
        norm[:, li] = e1 * (seg ** 2).sum(axis=1)
    ms = np.where(np.arange(T) < T1, ms1, ms2)
    ballast = (ms * W) ** 2 * nccf_ballast
    den = np.sqrt(norm + ballast[:, None])
    nccf_pitch = np.where(den != 0, inner / np.where(den != 0, den, 1.0), 0.0)
    den0 = np.sqrt(norm)
    nccf_pov = np.where(den0 != 0, inner / np.where(den0 != 0, den0, 1.0), 0.0)
    res = nccf_pitch @ ar.T
    res_pov = nccf_pov @ ar.T
    differ = T1 > 0 and not abs(ms1 - ms2) <= 0.01 * (abs(ms1) + abs(ms2))
    if (T < recompute_frame or T1 < recompute_frame) and differ:
        anp = norm.sum(axis=1) / L
        new_b = (ms2 * W) ** 2 * nccf_ballast
        res = res * np.sqrt((ballast + anp) / (new_b + anp))[:, None]
    # Viterbi: full search (Kaldi's bounded two-sweep search finds the same argmin; ties -> lowest index)
    iff = np.log(1.0 + delta_pitch) ** 2 * penalty_factor
    trans = (np.arange(S)[None, :] - np.arange(S)[:, None]) ** 2 * iff   # [i (this), j (previous)]
    fwd = np.zeros(S)
    bp = np.zeros((T, S), dtype=np.int64)
    for t in range(T):
        cost = trans + fwd[None, :]
        bp[t] = cost.argmin(axis=1)
        fwd = cost[np.arange(S), bp[t]] + (1.0 - res[t] + soft_min_f0 * lags * res[t])
        fwd -= fwd.min()
    states = np.zeros(T, dtype=np.int64)
    best = int(fwd.argmin())
    out = np.zeros((T, 2))
    for t in range(T - 1, -1, -1):
        states[t] = best
        out[t] = (res_pov[t, best], 1.0 / lags[best])
        best = bp[t, best]
    return dict(out=out, nccf=res, states=states, lags=lags)


# ---------------------------------------------------------------------------------------------------------
# Round 4: float64 restatements of the remaining families - PLP (+ RASTA), VTLN-warped mel banks, delta,
# CMVN, sliding-window CMVN and the pitch post-processing -, written from the algorithm descriptions
# ([KALDI-UPSTREAM] mel-computations.cc / feature-plp.cc / feature-functions.cc /
# cmvn.cc / pitch-functions.cc, and the reference's in-tree recipe plp.py:64-168, :548-626) with numpy's own
# operations and summation orders; no code is shared with oracle/kaldi_oracle.c.  tests/test_spec_f64.py
# bounds |C oracle - float64| for each; tools/f64_report.py writes the table (profiles/r04_f64_report.txt).
# ---------------------------------------------------------------------------------------------------------
def inverse_mel_scale(m):
    return 700.0 * (np.exp(m / 1127.0) - 1.0)


def vtln_warp_freq(vtln_low, vtln_high, low_freq, high_freq, warp, freq):
    """mel-computations.cc VtlnWarpFreq: piecewise-linear warp of [low_freq, high_freq] onto itself"""
    freq = np.asarray(freq, dtype=np.float64)
    l = vtln_low * max(1.0, warp)
    h = vtln_high * min(1.0, warp)
    scale = 1.0 / warp
    fl, fh = scale * l, scale * h
    scale_left = (fl - low_freq) / (l - low_freq)
    scale_right = (high_freq - fh) / (high_freq - h)
    out = np.where(freq < l, low_freq + scale_left * (freq - low_freq),
                   np.where(freq < h, scale * freq, high_freq + scale_right * (freq - high_freq)))
    return np.where((freq < low_freq) | (freq > high_freq), freq, out)


def mel_banks_vtln(num_bins, sample_rate, padded, low_freq=20.0, high_freq=0.0, vtln_low=100.0,
                   vtln_high=-500.0, warp=1.0):
    """(dense [num_bins, padded / 2] weights, centre frequencies) with the bin edges warped in the
    mel domain (reference processor/base.py:376-406 -> MelBanks(..., vtln_warp))"""
    nyquist = 0.5 * sample_rate
    high = high_freq if high_freq > 0 else nyquist + high_freq
    vhigh = vtln_high if vtln_high >= 0 else nyquist + vtln_high
    mlow, mhigh = mel_scale(low_freq), mel_scale(high)
    delta = (mhigh - mlow) / (num_bins + 1)
    edges = mlow + delta * np.arange(num_bins + 2)
    if warp != 1.0:
        edges = mel_scale(vtln_warp_freq(vtln_low, vhigh, low_freq, high, warp, inverse_mel_scale(edges)))
    nfft = padded // 2
    mel = mel_scale(np.arange(nfft) * sample_rate / padded)
    w = np.zeros((num_bins, nfft))
    for b in range(num_bins):
        left, center, right = edges[b], edges[b + 1], edges[b + 2]
        inside = (mel > left) & (mel < right)
        w[b] = np.where(inside, np.where(mel <= center, (mel - left) / (center - left),
                                         (right - mel) / (right - center)), 0.0)
    return w, inverse_mel_scale(edges[1:-1])


def equal_loudness(center_freqs):
    fsq = np.asarray(center_freqs, dtype=np.float64) ** 2
    fsub = fsq / (fsq + 1.6e5)
    return fsub * fsub * ((fsq + 1.44e6) / (fsq + 9.61e6))


def idft_bases(n_bases, dim):
    """feature-functions.cc InitIdftBases: cosine bases of the inverse DFT of a symmetric spectrum"""
    angle = np.pi / (dim - 1)
    scale = 1.0 / (2.0 * (dim - 1))
    i = np.arange(n_bases)[:, None]
    j = np.arange(dim)[None, :]
    m = 2.0 * scale * np.cos(angle * i * j)
    m[:, 0] = scale
    m[:, dim - 1] = scale * np.cos(angle * np.arange(n_bases) * (dim - 1))
    return m


def durbin(autocorr):
    """(lpc [order], residual energy): Levinson-Durbin with Kaldi's 1e-5 floor on 1 - k^2"""
    order = len(autocorr) - 1
    lpc = np.zeros(order)
    e = float(autocorr[0])
    for i in range(order):
        ki = (autocorr[i + 1] + np.dot(lpc[:i], autocorr[i:0:-1])) / e
        e *= max(1.0 - ki * ki, 1.0e-5)
        new = lpc.copy()
        new[i] = -ki
        new[:i] = lpc[:i] - ki * lpc[:i][::-1]
        lpc = new
    return lpc, e


def lpc_to_cepstrum(lpc):
    order = len(lpc)
    cep = np.zeros(order)
    for i in range(order):
        s = sum((i - j) * lpc[j] * cep[i - j - 1] for j in range(i))
        cep[i] = -lpc[i] - s / (i + 1)
    return cep


def rasta(mel, do_log=True):
    """RASTA over the frames of one utterance, [n, bins] -> [n, bins] (reference plp.py:64-146: numerator
    -[-2 .. 2] / 10, denominator [1, -0.94]; the first four frames emit zeros in the log domain - ones after
    the exponential - and prime the FIR delay line)"""
    import scipy.signal
    x = np.asarray(mel, dtype=np.float64)
    if do_log:
        x = np.log(x + np.finfo(np.float64).eps)
    numer = -np.arange(-2, 3) / np.sum(np.arange(-2, 3) ** 2)
    denom = np.array([1.0, -0.94])
    out = np.zeros_like(x)
    n = x.shape[0]
    head = min(4, n)
    # the FIR part sees every frame from the first; the IIR part starts at frame 4 from a zero state
    fir_state = np.zeros((4, x.shape[1]))
    if head:
        _, fir_state = scipy.signal.lfilter(numer, [1.0], x[:head], axis=0, zi=fir_state)
    if n > 4:
        # one filter with both parts, started with the FIR delay line primed and no recursive memory: in
        # direct form II transposed that is exactly the state a pure FIR run leaves behind
        y, _ = scipy.signal.lfilter(numer, denom, x[4:], axis=0, zi=fir_state)
        out[4:] = y
    return np.exp(out) if do_log else out


def plp(wave, sample_rate=16000, frame_shift=0.01, frame_length=0.025, preemph=0.97, remove_dc=True,
        window='povey', snip_edges=True, num_bins=23, low_freq=20.0, high_freq=0.0, vtln_low=100.0,
        vtln_high=-500.0, warp=1.0, lpc_order=12, num_ceps=13, cepstral_lifter=22.0, cepstral_scale=1.0,
        compress_factor=1.0 / 3.0, use_energy=True, raw_energy=True, energy_floor=0.0, htk_compat=False,
        use_rasta=False, blackman_coeff=0.42, round_pow2=True):
    """PlpProcessor (reference plp.py:510-626) in float64"""
    shift, length, padded = frame_geometry(sample_rate, frame_shift, frame_length, round_pow2)
    x = extract_frames(wave, shift, length, snip_edges)
    if x.shape[0] == 0:
        return np.zeros((0, num_ceps))
    if remove_dc:
        x = x - x.mean(axis=1, keepdims=True)
    eps64 = np.finfo(np.float64).eps
    raw_log_energy = np.log(np.maximum((x * x).sum(axis=1), eps64))
    if preemph != 0:
        y = x.copy()
        y[:, 1:] = x[:, 1:] - preemph * x[:, :-1]
        y[:, 0] = x[:, 0] - preemph * x[:, 0]
        x = y
    x = x * window_function(length, window, blackman_coeff)[None, :]
    post_log_energy = np.log(np.maximum((x * x).sum(axis=1), eps64))
    spec = np.fft.rfft(x, n=padded, axis=1)
    power = (spec.real ** 2 + spec.imag ** 2)[:, :padded // 2]
    w, centers = mel_banks_vtln(num_bins, sample_rate, padded, low_freq, high_freq, vtln_low, vtln_high, warp)
    mel = power @ w.T
    if use_rasta:
        mel = rasta(mel, do_log=True)
    mel = (mel * equal_loudness(centers)[None, :]) ** float(np.float32(compress_factor))
    dup = np.concatenate([mel[:, :1], mel, mel[:, -1:]], axis=1)
    ac = dup @ idft_bases(lpc_order + 1, num_bins + 2).T
    out = np.zeros((x.shape[0], num_ceps))
    for t in range(x.shape[0]):
        lpc, e = durbin(ac[t])
        out[t, 0] = max(np.log(e), eps64)
        out[t, 1:] = lpc_to_cepstrum(lpc)[:num_ceps - 1]
    if cepstral_lifter:
        out = out * (1 + 0.5 * cepstral_lifter * np.sin(np.pi * np.arange(num_ceps) / cepstral_lifter))[None, :]
    out = out * cepstral_scale
    if use_energy:
        le = raw_log_energy if raw_energy else post_log_energy
        if energy_floor > 0:
            le = np.maximum(le, np.log(energy_floor))
        out[:, 0] = le
    if htk_compat:
        out = np.concatenate([out[:, 1:], out[:, :1]], axis=1)
    return out


def delta_scales(order, window):
    """scales[i] of the i-th derivative: i-fold convolution of [-w..w] / sum(j^2)"""
    scales = [np.array([1.0])]
    base = np.arange(-window, window + 1, dtype=np.float64) / float(sum(j * j for j in range(-window, window + 1)))
    for _ in range(order):
        scales.append(np.convolve(scales[-1], base))
    return scales


def delta(feats, order=2, window=2):
    """DeltaPostProcessor (reference delta.py:113-136): [n, d] -> [n, d (order + 1)], frames clamped"""
    x = np.asarray(feats, dtype=np.float64)
    n = x.shape[0]
    blocks = []
    for s in delta_scales(order, window):
        half = (len(s) - 1) // 2
        acc = np.zeros_like(x)
        for k, c in enumerate(s):
            idx = np.clip(np.arange(n) + k - half, 0, n - 1)
            acc += c * x[idx]
        blocks.append(acc)
    return np.concatenate(blocks, axis=1)


def cmvn_stats(feats, weights=None):
    """[2, d + 1]: row 0 = weighted sums and the count, row 1 = weighted sums of squares"""
    x = np.asarray(feats, dtype=np.float64)
    w = np.ones(x.shape[0]) if weights is None else np.asarray(weights, dtype=np.float64)
    stats = np.zeros((2, x.shape[1] + 1))
    stats[0, :-1] = (w[:, None] * x).sum(axis=0)
    stats[1, :-1] = (w[:, None] * x * x).sum(axis=0)
    stats[0, -1] = w.sum()
    return stats


def cmvn_apply(feats, stats, norm_vars=True, reverse=False):
    x = np.asarray(feats, dtype=np.float64)
    count = stats[0, -1]
    mean = stats[0, :-1] / count
    if not norm_vars:
        return x + mean if reverse else x - mean
    var = np.maximum(stats[1, :-1] / count - mean * mean, 1.0e-20)
    if reverse:
        return x * np.sqrt(var) + mean
    return (x - mean) / np.sqrt(var)


def sliding_cmvn(feats, center=True, cmn_window=600, min_window=100, normalize_variance=False):
    """SlidingWindowCmvnPostProcessor -> [KALDI-UPSTREAM] SlidingWindowCmn"""
    x = np.asarray(feats, dtype=np.float64)
    n = x.shape[0]
    out = np.zeros_like(x)
    for t in range(n):
        if center:
            begin = t - cmn_window // 2
            end = begin + cmn_window
        else:
            begin, end = t - cmn_window, t + 1
        if begin < 0:
            end -= begin
            begin = 0
        if not center:
            if end > t:
                end = max(t + 1, min_window)
        if end > n:
            begin -= end - n
            end = n
            if begin < 0:
                begin = 0
        seg = x[begin:end]
        mean = seg.mean(axis=0)
        out[t] = x[t] - mean
        if normalize_variance:
            if end - begin == 1:
                out[t] = 0.0
            else:
                var = np.maximum((seg * seg).mean(axis=0) - mean * mean, 1.0e-10)
                out[t] = out[t] / np.sqrt(var)
    return out


def nccf_to_pov_feature(n):
    n = np.clip(n, -1.0, 1.0)
    return (1.0001 - n) ** 0.15 - 1.0


def nccf_to_pov(n):
    nd = np.minimum(np.abs(n), 1.0)
    r = -5.2 + 5.4 * np.exp(7.5 * (nd - 1.0)) + 4.8 * nd - 2.0 * np.exp(-10.0 * nd) + 4.2 * np.exp(20.0 * (nd - 1.0))
    return 1.0 / (1.0 + np.exp(-r))


def process_pitch(raw, pitch_scale=2.0, pov_scale=2.0, pov_offset=0.0, delta_pitch_scale=10.0,
                  left_context=75, right_context=75, delta_window=2, add_pov_feature=True,
                  add_normalized_log_pitch=True, add_delta_pitch=True, add_raw_log_pitch=False):
    """KaldiPitchPostProcessor (reference pitch_kaldi.py:497-540 -> ProcessPitch) without the noise term:
    [n, 2] (NCCF, pitch) -> [n, k]"""
    raw = np.asarray(raw, dtype=np.float64)
    nccf, log_pitch = raw[:, 0], np.log(raw[:, 1])
    n = raw.shape[0]
    cols = []
    if add_pov_feature:
        cols.append(pov_scale * nccf_to_pov_feature(nccf) + pov_offset)
    if add_normalized_log_pitch:
        pov = nccf_to_pov(nccf)
        norm = np.zeros(n)
        for t in range(n):
            lo, hi = max(0, t - left_context), min(n, t + right_context + 1)
            norm[t] = log_pitch[t] - np.sum(pov[lo:hi] * log_pitch[lo:hi]) / np.sum(pov[lo:hi])
        cols.append(pitch_scale * norm)
    if add_delta_pitch:
        cols.append(delta_pitch_scale * delta(log_pitch[:, None], 1, delta_window)[:, 1])
    if add_raw_log_pitch:
        cols.append(log_pitch)
    return np.stack(cols, axis=1)


def vad_energy(feats, energy_threshold=5.0, energy_mean_scale=0.5, frames_context=0, proportion_threshold=0.6):
    """VadPostProcessor (reference postprocessor/vad.py:84-113 -> [KALDI-UPSTREAM] ComputeVadEnergy): column 0 of the
    features is the log energy; a frame is voiced when at least `proportion_threshold` of the frames within
    `frames_context` of it (inside the utterance) lie above energy_threshold + energy_mean_scale * mean(log energy).
    -> ([n] of 0 / 1, the margin of the closest frame to the threshold)"""
    e = np.asarray(feats, dtype=np.float64)[:, 0]
    n = e.shape[0]
    threshold = energy_threshold + (energy_mean_scale * e.sum() / n if energy_mean_scale != 0 else 0.0)
    above = e > threshold
    out = np.zeros(n)
    for t in range(n):
        lo, hi = max(0, t - frames_context), min(n, t + frames_context + 1)
        out[t] = 1.0 if above[lo:hi].sum() >= (hi - lo) * proportion_threshold else 0.0
    return out, float(np.abs(e - threshold).min()) if n else 0.0


def energy(wave, sample_rate=16000, frame_shift=0.01, frame_length=0.025, preemph=0.97, remove_dc=True,
           window='povey', snip_edges=True, raw_energy=True, compression='log', blackman_coeff=0.42):
    """EnergyProcessor (reference processor/energy.py:150-183): Kaldi's ExtractWindow (DC removal, pre-emphasis,
    window; raw_energy = no pre-emphasis and a rectangular window), then the float64 sum of squares of the frame,
    floored at the smallest normal double, compressed -> [n, 1]"""
    shift, length, _ = frame_geometry(sample_rate, frame_shift, frame_length)
    x = extract_frames(wave, shift, length, snip_edges)
    if x.shape[0] == 0:
        return np.zeros((0, 1))
    if remove_dc:
        x = x - x.mean(axis=1, keepdims=True)
    if raw_energy:
        preemph, window = 0.0, 'rectangular'
    if preemph != 0:
        y = x.copy()
        y[:, 1:] = x[:, 1:] - preemph * x[:, :-1]
        y[:, 0] = x[:, 0] - preemph * x[:, 0]
        x = y
    x = x * window_function(length, window, blackman_coeff)[None, :]
    e = np.maximum((x * x).sum(axis=1), np.finfo(np.float64).tiny)
    e = {'log': np.log, 'sqrt': np.sqrt, 'off': lambda v: v}[compression](e)
    return e[:, None]
