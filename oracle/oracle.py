"""ctypes front-end of the CPU oracle (``oracle/liboracle.so``).  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import this
module; the product package ``shennong_amd`` never does (see kaldi_oracle.c header).
"""

import ctypes as C
import os
import subprocess

import numpy as np

from shennong_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    """Compile liboracle.so with the committed Makefile (gcc only)"""
    subprocess.run(['make', '-C', _HERE], check=True, capture_output=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, 'liboracle.so')
    src = os.path.join(_HERE, 'kaldi_oracle.c')
    if (not os.path.exists(path) or
            os.path.getmtime(path) < os.path.getmtime(src)):
        build()
    L = C.CDLL(path)
    i64, i32, f32 = C.c_int64, C.c_int32, C.c_float
    pf = C.POINTER(C.c_float)
    FO, MO = C.POINTER(_abi.FrameOptions), C.POINTER(_abi.MelOptions)
    OP = C.POINTER(_abi.Options)
    PO, PPO = C.POINTER(_abi.PitchOptions), C.POINTER(_abi.PitchPostOptions)
    L.orc_last_error.restype = C.c_char_p
    L.orc_window_shift.argtypes = [FO]
    L.orc_window_size.argtypes = [FO]
    L.orc_padded_window_size.argtypes = [FO]
    L.orc_num_frames.argtypes = [FO, i64]
    L.orc_num_frames.restype = i64
    L.orc_first_sample_of_frame.argtypes = [FO, i64]
    L.orc_first_sample_of_frame.restype = i64
    L.orc_window_function.argtypes = [FO, pf]
    L.orc_extract_window.argtypes = [FO, pf, i64, i64, pf, pf]
    L.orc_mel_banks.argtypes = [
        MO, FO, f32, C.POINTER(i32), C.POINTER(i32), pf, pf]
    L.orc_dct_matrix.argtypes = [pf, i32, i32]
    L.orc_dct_matrix.restype = None
    L.orc_lifter_coeffs.argtypes = [f32, pf, i32]
    L.orc_lifter_coeffs.restype = None
    L.orc_lpc2cepstrum.argtypes = [i32, pf, pf]
    L.orc_lpc2cepstrum.restype = None
    L.orc_rasta.argtypes = [pf, i64, i32, pf, i32]
    L.orc_ndims.argtypes = [OP]
    L.orc_compute.argtypes = [OP, C.POINTER(C.c_int16), i64, f32, pf]
    L.orc_compute_batch.argtypes = [
        OP, C.POINTER(C.c_int16), C.POINTER(i64), C.POINTER(i64), i64, pf, i32]
    L.orc_delta_scales.argtypes = [i32, i32, pf]
    L.orc_deltas.argtypes = [i32, i32, pf, i64, i32, pf]
    L.orc_pitch_num_frames.argtypes = [PO, i64]
    L.orc_pitch_num_frames.restype = i64
    L.orc_pitch.argtypes = [PO, C.POINTER(C.c_int16), i64, pf]
    L.orc_pitch_debug.argtypes = [PO, C.POINTER(C.c_int16), i64, pf, pf, pf, pf, C.POINTER(C.c_int32)]
    L.orc_pitch_lags.argtypes = [PO, pf, C.POINTER(i32), C.POINTER(i32)]
    L.orc_linear_resample.argtypes = [i32, i32, f32, i32, pf, i64, pf, i32]
    L.orc_linear_resample.restype = i64
    L.orc_vad_energy.argtypes = [f32, f32, i32, f32, pf, i64, i32, pf]
    L.orc_cmvn_accumulate.argtypes = [pf, i64, i32, pf, C.POINTER(C.c_double)]
    L.orc_cmvn_apply.argtypes = [C.POINTER(C.c_double), i32, i32, i32, pf, i64]
    L.orc_sliding_cmn.argtypes = [i32, i32, i32, i32, pf, i64, i32, pf]
    L.orc_process_pitch_ndims.argtypes = [PPO]
    L.orc_process_pitch.argtypes = [PPO, pf, i64, pf]
    _LIB = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().orc_last_error().decode())


def num_frames(frame_opts, nsamples):
    return int(lib().orc_num_frames(C.byref(frame_opts), int(nsamples)))


def first_sample_of_frame(frame_opts, frame):
    return int(lib().orc_first_sample_of_frame(C.byref(frame_opts), frame))


def window_size(frame_opts):
    return int(lib().orc_window_size(C.byref(frame_opts)))


def window_shift(frame_opts):
    return int(lib().orc_window_shift(C.byref(frame_opts)))


def padded_window_size(frame_opts):
    return int(lib().orc_padded_window_size(C.byref(frame_opts)))


def window_function(frame_opts):
    out = np.zeros(window_size(frame_opts), dtype=np.float32)
    _check(lib().orc_window_function(C.byref(frame_opts), _fp(out)))
    return out


def extract_window(frame_opts, wave, frame):
    """Kaldi ExtractWindow+ProcessWindow of one frame -> (padded window, raw log energy)"""
    wave = np.ascontiguousarray(wave, dtype=np.float32)
    out = np.zeros(padded_window_size(frame_opts), dtype=np.float32)
    energy = C.c_float(0)
    _check(lib().orc_extract_window(
        C.byref(frame_opts), _fp(wave), wave.shape[0], frame, _fp(out),
        C.byref(energy)))
    return out, energy.value


def mel_banks(mel_opts, frame_opts, vtln_warp=1.0):
    nb = mel_opts.num_bins
    nfft = padded_window_size(frame_opts) // 2
    first = np.zeros(max(nb, 1), dtype=np.int32)
    size = np.zeros(max(nb, 1), dtype=np.int32)
    weights = np.zeros((max(nb, 1), nfft), dtype=np.float32)
    center = np.zeros(max(nb, 1), dtype=np.float32)
    _check(lib().orc_mel_banks(
        C.byref(mel_opts), C.byref(frame_opts), vtln_warp,
        first.ctypes.data_as(C.POINTER(C.c_int32)),
        size.ctypes.data_as(C.POINTER(C.c_int32)), _fp(weights), _fp(center)))
    return first, size, weights, center


def dct_matrix(nrows, ncols):
    out = np.zeros((nrows, ncols), dtype=np.float32)
    lib().orc_dct_matrix(_fp(out), nrows, ncols)
    return out


def lifter_coeffs(q, n):
    out = np.zeros(n, dtype=np.float32)
    lib().orc_lifter_coeffs(q, _fp(out), n)
    return out


def lpc2cepstrum(lpc):
    lpc = np.ascontiguousarray(lpc, dtype=np.float32)
    out = np.zeros_like(lpc)
    lib().orc_lpc2cepstrum(lpc.shape[0], _fp(lpc), _fp(out))
    return out


def rasta(mel, do_log=True):
    mel = np.ascontiguousarray(mel, dtype=np.float32)
    out = np.zeros_like(mel)
    _check(lib().orc_rasta(
        _fp(mel), mel.shape[0], mel.shape[1], _fp(out), int(do_log)))
    return out


def ndims(opts):
    return int(lib().orc_ndims(C.byref(opts)))


def compute(opts, wave, vtln_warp=1.0):
    """Audio (int16 array) -> float32 [nframes, ndims] for spectrogram/fbank/mfcc/plp/energy"""
    wave = np.ascontiguousarray(wave, dtype=np.int16)
    nframes = num_frames(opts.frame, wave.shape[0])
    d = ndims(opts)
    out = np.zeros((nframes, max(d, 0)), dtype=np.float32)
    _check(lib().orc_compute(
        C.byref(opts), wave.ctypes.data_as(C.POINTER(C.c_int16)),
        wave.shape[0], vtln_warp, _fp(out)))
    if nframes == 0:
        return np.zeros((0, 0), dtype=np.float32)
    return out


def compute_batch(opts, waves, nthreads=1):
    """[n_utts, nsamples] int16 -> concatenated float32 [total_frames, ndims], one utterance per
    task over `nthreads` POSIX threads (CPU baseline driver)"""
    waves = np.ascontiguousarray(waves, dtype=np.int16)
    n, ns = waves.shape
    nf = num_frames(opts.frame, ns)
    soff = np.arange(n + 1, dtype=np.int64) * ns
    foff = np.arange(n + 1, dtype=np.int64) * nf
    out = np.zeros((n * nf, ndims(opts)), dtype=np.float32)
    _check(lib().orc_compute_batch(
        C.byref(opts), waves.ctypes.data_as(C.POINTER(C.c_int16)),
        soff.ctypes.data_as(C.POINTER(C.c_int64)),
        foff.ctypes.data_as(C.POINTER(C.c_int64)), n, _fp(out), int(nthreads)))
    return out


def delta_scales(order, window):
    total = sum(2 * i * window + 1 for i in range(order + 1))
    out = np.zeros(total, dtype=np.float32)
    _check(lib().orc_delta_scales(order, window, _fp(out)))
    res, pos = [], 0
    for i in range(order + 1):
        n = 2 * i * window + 1
        res.append(out[pos:pos + n].copy())
        pos += n
    return res


def deltas(data, order=2, window=2):
    data = np.ascontiguousarray(data, dtype=np.float32)
    out = np.zeros(
        (data.shape[0], data.shape[1] * (order + 1)), dtype=np.float32)
    _check(lib().orc_deltas(
        order, window, _fp(data), data.shape[0], data.shape[1], _fp(out)))
    return out


def pitch_num_frames(pitch_opts, nsamples):
    return int(lib().orc_pitch_num_frames(C.byref(pitch_opts), int(nsamples)))


def pitch(pitch_opts, wave):
    wave = np.ascontiguousarray(wave, dtype=np.int16)
    nframes = pitch_num_frames(pitch_opts, wave.shape[0])
    if nframes < 0:
        raise RuntimeError(lib().orc_last_error().decode())
    out = np.zeros((max(nframes, 0), 2), dtype=np.float32)
    _check(lib().orc_pitch(
        C.byref(pitch_opts), wave.ctypes.data_as(C.POINTER(C.c_int16)),
        wave.shape[0], _fp(out)))
    if nframes == 0:
        return np.zeros((0, 0), dtype=np.float32)
    return out


def pitch_debug(pitch_opts, wave):
    """(out, down, nccf_res [T, S], pov_nccf [T, L], states [T]): the intermediates of `pitch`"""
    wave = np.ascontiguousarray(wave, dtype=np.int16)
    T = pitch_num_frames(pitch_opts, wave.shape[0])
    lags, first, last = pitch_lags(pitch_opts)
    S, L = lags.shape[0], last - first + 1
    n_down = lib().orc_linear_resample(
        int(pitch_opts.samp_freq), int(pitch_opts.resample_freq), pitch_opts.lowpass_cutoff,
        int(pitch_opts.lowpass_filter_width), None, wave.shape[0], None, 1)
    out = np.zeros((max(T, 0), 2), dtype=np.float32)
    down = np.zeros(max(n_down, 1), dtype=np.float32)
    res = np.zeros((max(T, 1), S), dtype=np.float32)
    pov = np.zeros((max(T, 1), L), dtype=np.float32)
    states = np.zeros(max(T, 1), dtype=np.int32)
    _check(lib().orc_pitch_debug(
        C.byref(pitch_opts), wave.ctypes.data_as(C.POINTER(C.c_int16)), wave.shape[0], _fp(out),
        _fp(down), _fp(res), _fp(pov), states.ctypes.data_as(C.POINTER(C.c_int32))))
    return out, down[:n_down], res[:T], pov[:T], states[:T]


def pitch_lags(pitch_opts):
    lags = np.zeros(4096, dtype=np.float32)
    first, last = C.c_int32(0), C.c_int32(0)
    n = lib().orc_pitch_lags(
        C.byref(pitch_opts), _fp(lags), C.byref(first), C.byref(last))
    if n < 0:
        raise RuntimeError(lib().orc_last_error().decode())
    return lags[:n].copy(), first.value, last.value


def linear_resample(wave, rate_in, rate_out, cutoff, num_zeros, flush=True):
    wave = np.ascontiguousarray(wave, dtype=np.float32)
    n = lib().orc_linear_resample(
        rate_in, rate_out, cutoff, num_zeros, _fp(wave), wave.shape[0], None,
        int(flush))
    out = np.zeros(max(n, 0), dtype=np.float32)
    lib().orc_linear_resample(
        rate_in, rate_out, cutoff, num_zeros, _fp(wave), wave.shape[0],
        _fp(out), int(flush))
    return out


def process_pitch(post_opts, raw):
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    d = int(lib().orc_process_pitch_ndims(C.byref(post_opts)))
    out = np.zeros((raw.shape[0], max(d, 1)), dtype=np.float32)
    _check(lib().orc_process_pitch(
        C.byref(post_opts), _fp(raw), raw.shape[0], _fp(out)))
    return out


def vad_energy(feats, energy_threshold=5.0, energy_mean_scale=0.5, frames_context=0,
               proportion_threshold=0.6):
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    out = np.zeros(feats.shape[0], dtype=np.float32)
    _check(lib().orc_vad_energy(
        energy_threshold, energy_mean_scale, frames_context, proportion_threshold,
        _fp(feats), feats.shape[0], feats.shape[1], _fp(out)))
    return out


def cmvn_accumulate(feats, weights=None, stats=None):
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    d = feats.shape[1]
    if stats is None:
        stats = np.zeros((2, d + 1), dtype=np.float64)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
    _check(lib().orc_cmvn_accumulate(
        _fp(feats), feats.shape[0], d, _fp(w) if w is not None else None,
        stats.ctypes.data_as(C.POINTER(C.c_double))))
    return stats


def cmvn_apply(feats, stats, norm_vars=True, reverse=False):
    out = np.array(feats, dtype=np.float32, order='C', copy=True)
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    _check(lib().orc_cmvn_apply(
        stats.ctypes.data_as(C.POINTER(C.c_double)), out.shape[1], int(norm_vars),
        int(reverse), _fp(out), out.shape[0]))
    return out


def sliding_cmn(feats, center=True, cmn_window=600, min_window=100, normalize_variance=False):
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    out = np.zeros_like(feats)
    _check(lib().orc_sliding_cmn(
        int(center), cmn_window, min_window, int(normalize_variance), _fp(feats),
        feats.shape[0], feats.shape[1], _fp(out)))
    return out
