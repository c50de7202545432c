"""Pins the CPU oracle (oracle/kaldi_oracle.c) against everything the reference's own tests hold for
the hot path, against fixtures generated from the reference's runnable numpy
code (tests/golden/make_golden.py) and against the independent float64 restatement.

Coefficient VALUES of fbank / MFCC / PLP / pitch are "parity unpinned" by the reference (no golden
files, Kaldi not runnable here); what is pinned is listed test by test below."""

import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as orc, spec_f64
from shennong_amd import synth
from shennong_amd import _abi


@pytest.fixture(scope='module')
def golden():
    return np.load(os.path.join(GOLDEN, 'reference_numpy.npz'))


def _frame_opts(**kw):
    o = _abi.default_frame_options()
    for k, v in kw.items():
        setattr(o, k, v)
    return o


# ---- frame counts: every shape the reference tests assert -----------------------------------------
@pytest.mark.parametrize('nsamples, shift_ms, length_ms, snip, sr, expected', [
    (22713, 10, 25, True, 16000, 140),    # test_filterbank.py:61-62, test_mfcc.py:114
    (22713, 20, 25, True, 16000, 70),     # test_filterbank.py:63-64
    (22713, 20, 50, True, 16000, 69),     # test_filterbank.py:65-66
    (22713, 10, 25, False, 16000, 142),   # test_plp.py:70-74
    (16000, 10, 25, True, 16000, 98),     # test_pipeline.py:302-308 (1 s)
    (3200, 10, 25, True, 16000, 18),      # test_pipeline.py (0.2 s)
    (11356, 10, 25, True, 8000, 140),     # test_mfcc.py:131-137 (8 kHz resample of the clip)
    (10, 10, 25, True, 16000, 0),
])
def test_num_frames(nsamples, shift_ms, length_ms, snip, sr, expected):
    o = _frame_opts(samp_freq=sr, frame_shift_ms=shift_ms, frame_length_ms=length_ms,
                    snip_edges=int(snip))
    assert orc.num_frames(o, nsamples) == expected


@pytest.mark.parametrize('length, shift, n, snip, expected', [
    (1, 1, 10, True, 10), (1, 1, 10, False, 10), (2, 1, 10, True, 9), (2, 1, 10, False, 10),
    (2, 2, 10, True, 5), (2, 2, 10, False, 5), (1, 2, 10, True, 5), (1, 2, 10, False, 5),
    (3, 1, 10, True, 8), (3, 1, 10, False, 10), (5, 3, 9, True, 2), (5, 3, 9, False, 3)])
def test_num_frames_frames_literals(length, shift, n, snip, expected):
    """reference test/test_frames.py:18-138 (sample_rate=1, shift/length in seconds)"""
    o = _frame_opts(samp_freq=1, frame_shift_ms=shift * 1000.0, frame_length_ms=length * 1000.0,
                    snip_edges=int(snip))
    assert orc.num_frames(o, n) == expected


def test_first_sample_of_frame():
    o = _frame_opts()
    assert [orc.first_sample_of_frame(o, f) for f in range(3)] == [0, 160, 320]
    o.snip_edges = 0
    assert [orc.first_sample_of_frame(o, f) for f in range(3)] == [-120, 40, 200]


# ---- window known answers (reference shennong/window.py:43-49, test/test_window.py:12-28) ----------
def test_window_known_answers():
    o = _frame_opts(samp_freq=1000, frame_length_ms=5,
                    window_type=_abi.WINDOW_TYPES['hamming'])
    assert np.array_equal(orc.window_function(o),
                          np.array([0.08, 0.54, 1.0, 0.54, 0.08], dtype=np.float32))
    o.window_type = _abi.WINDOW_TYPES['povey']
    assert orc.window_function(o).tolist() == [
        0.0, 0.5547847151756287, 1.0, 0.5547847151756287, 0.0]
    o.window_type = _abi.WINDOW_TYPES['rectangular']
    assert np.all(orc.window_function(o) == 1.0)


@pytest.mark.parametrize('kind', ['hamming', 'hanning', 'povey', 'rectangular', 'blackman'])
@pytest.mark.parametrize('length', [3, 10, 100, 400])
def test_window_properties(kind, length):
    o = _frame_opts(samp_freq=1000, frame_length_ms=length, window_type=_abi.WINDOW_TYPES[kind])
    win = orc.window_function(o)
    assert win.shape == (length,) and not np.any(np.isnan(win))
    assert win.max() <= 1.0 and win.min() >= -1e-7
    if kind == 'povey':
        assert win[0] == win[-1] == 0.0
    np.testing.assert_allclose(win, spec_f64.window_function(length, kind), rtol=1e-6, atol=1e-7)


# ---- ExtractWindow reflection (reference plp.py:242-254) --------------------------------------------
def test_extract_window_reflection():
    o = _frame_opts(samp_freq=1000, frame_shift_ms=3, frame_length_ms=5, snip_edges=0,
                    remove_dc_offset=0, preemph_coeff=0.0, dither=0.0,
                    window_type=_abi.WINDOW_TYPES['rectangular'], round_to_power_of_two=0)
    wave = np.arange(9, dtype=np.float32)
    # frame 0 starts at 3*0 + 1 - 2 = -1: reflect -1 -> 0
    win, _ = orc.extract_window(o, wave, 0)
    assert win.tolist() == [0, 0, 1, 2, 3]
    # last frame (2) starts at 5, runs to 9: reflect 9 -> 8
    win, _ = orc.extract_window(o, wave, 2)
    assert win.tolist() == [5, 6, 7, 8, 8]


# ---- energy identity (reference test/processor/test_energy.py:36-44) --------------------------------
@pytest.mark.parametrize('raw_energy', [True, False])
def test_energy_identity(wave, raw_energy):
    """MFCC[:, 0] == PLP[:, 0] == EnergyProcessor == log sum (x - mean)^2 (raw) computed in numpy"""
    x = spec_f64.extract_frames(wave, 160, 400)
    x = x - x.mean(axis=1, keepdims=True)
    if not raw_energy:
        y = x.copy()
        y[:, 1:] = x[:, 1:] - np.float32(0.97) * x[:, :-1]
        y[:, 0] = x[:, 0] - np.float32(0.97) * x[:, 0]
        x = y * spec_f64.window_function(400, 'povey')
    want = np.log((x * x).sum(axis=1))
    cols = []
    for kind in (_abi.KIND_MFCC, _abi.KIND_PLP, _abi.KIND_ENERGY):
        o = _abi.default_options(kind)
        o.frame.dither = 0
        o.raw_energy = int(raw_energy)
        cols.append(orc.compute(o, wave)[:, 0])
        assert cols[-1].shape == (140,)
        np.testing.assert_allclose(cols[-1], want, rtol=2e-6)
    assert np.allclose(cols[0], cols[1]) and np.allclose(cols[0], cols[2])


def test_plp_energy_floor_known_answer(wave):
    """reference test_plp.py:77-81: energy_floor=exp(50), raw_energy=False -> column 0 == 50"""
    o = _abi.default_options(_abi.KIND_PLP)
    o.frame.dither = 0
    o.raw_energy = 0
    o.energy_floor = np.exp(50)
    feat = orc.compute(o, wave)
    assert feat.shape == (140, 13)
    assert np.all(feat[:, 0] == 50)


def test_htk_compat_rules(wave):
    """reference test_mfcc.py:100-111 and test_plp.py:50-61"""
    def run(kind, **kw):
        o = _abi.default_options(kind)
        o.frame.dither = 0
        for k, v in kw.items():
            setattr(o, k, v)
        return orc.compute(o, wave)
    for kind, c0_scale in ((_abi.KIND_MFCC, 2 ** 0.5), (_abi.KIND_PLP, 1.0)):
        a = run(kind, use_energy=1, htk_compat=0)
        b = run(kind, use_energy=1, htk_compat=1)
        assert a[:, 0] == pytest.approx(b[:, -1])
        assert np.array_equal(a[:, 1:], b[:, :-1])
        a = run(kind, use_energy=0, htk_compat=0)
        b = run(kind, use_energy=0, htk_compat=1)
        assert a[:, 0] * c0_scale == pytest.approx(b[:, -1])
    a = run(_abi.KIND_FBANK, use_energy=1, htk_compat=0)
    b = run(_abi.KIND_FBANK, use_energy=1, htk_compat=1)
    assert np.array_equal(a[:, 0], b[:, -1]) and np.array_equal(a[:, 1:], b[:, :-1])


def test_option_errors():
    """Kaldi KALDI_ERR cases surfaced as RuntimeError (reference test_filterbank.py:46-50,
    test_mfcc.py:72-98)"""
    wave = np.zeros(16000, dtype=np.int16)
    for nb in (0, 1, 2):
        o = _abi.default_options(_abi.KIND_FBANK)
        o.frame.dither = 0
        o.mel.num_bins = nb
        with pytest.raises(RuntimeError):
            orc.compute(o, wave)
    o = _abi.default_options(_abi.KIND_MFCC)
    o.frame.dither = 0
    o.num_ceps = 25
    with pytest.raises(RuntimeError):
        orc.compute(o, wave)


def test_when_the_mel_banks_are_built():
    """Kaldi's Fbank / Mfcc computers build the banks of warp factor 1 in their constructors: their option errors
    come with or without frames, whatever the utterance's own warp factor.  The reference's PLP is its own recipe and
    builds the banks of a warp factor when a frame asks for them (shennong/processor/plp.py:482-494, :559): no frames,
    no error; a warped utterance never sees the errors of the unwarped banks.  8 ms frames at 8 kHz leave bin 0 of 23
    unwarped bins empty; the banks of warp factor 0.85 are complete."""
    wave = synth.utterances(7, 1, 9670, 8000)[0]

    def options(kind):
        o = _abi.default_options(kind)
        o.frame.samp_freq, o.frame.frame_length_ms, o.frame.frame_shift_ms = 8000, 8.0, 20.0
        o.frame.dither, o.frame.snip_edges, o.frame.window_type = 0, 0, _abi.WINDOW_TYPES['rectangular']
        o.mel.num_bins, o.mel.low_freq, o.mel.high_freq, o.mel.vtln_low, o.mel.vtln_high = 23, 0, -200, 100, -500
        return o
    plp = options(_abi.KIND_PLP)
    plp.num_ceps = 7
    assert orc.compute(plp, wave, 0.85).shape == (60, 7)
    with pytest.raises(RuntimeError, match='num_bins too large'):
        orc.compute(plp, wave, 1.0)
    assert orc.compute(plp, wave[:40], 1.0).size == 0          # no frames: no banks
    for kind in (_abi.KIND_FBANK, _abi.KIND_MFCC):
        o = options(kind)
        for length, warp in ((len(wave), 0.85), (len(wave), 1.0), (40, 1.0), (40, 0.85)):
            with pytest.raises(RuntimeError, match='num_bins too large'):
                orc.compute(o, wave[:length], warp)


# ---- float64 restatement -------------------------------------------------------------------------------
@pytest.mark.parametrize('kind, code, kw', [
    ('fbank', _abi.KIND_FBANK, dict(num_bins=40)),
    ('fbank', _abi.KIND_FBANK, dict(num_bins=23, snip_edges=False)),
    ('mfcc', _abi.KIND_MFCC, dict()),
    ('spectrogram', _abi.KIND_SPECTROGRAM, dict()),
])
def test_against_float64_spec(wave, kind, code, kw):
    o = _abi.default_options(code)
    o.frame.dither = 0
    o.mel.num_bins = kw.get('num_bins', 23)
    o.frame.snip_edges = int(kw.get('snip_edges', True))
    got = orc.compute(o, wave)
    want = spec_f64.features(
        wave, kind, num_bins=o.mel.num_bins, snip_edges=bool(o.frame.snip_edges),
        use_energy=(None if kind == 'mfcc' else False))
    assert got.shape == want.shape
    atol = {'fbank': 1e-4, 'mfcc': 5e-4, 'spectrogram': 5e-3}[kind]
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=atol)


def test_mel_banks_structure():
    """strict (left, right) support, Nyquist bin never used, two non-zeros per FFT bin at most"""
    first, size, w, center = orc.mel_banks(
        _abi.MelOptions(num_bins=40, low_freq=20, high_freq=0, vtln_low=100, vtln_high=-500),
        _abi.default_frame_options())
    assert w.shape == (40, 256)
    assert np.all((w > 0).sum(axis=0) <= 2)
    assert np.all(np.diff(center) > 0) and first[0] >= 1
    for b in range(40):
        nz = np.nonzero(w[b])[0]
        assert nz[0] == first[b] and len(nz) == size[b] and w[b].max() <= 1.0
    np.testing.assert_allclose(w, spec_f64.mel_banks(40, 16000.0, 512), atol=2e-5)


# ---- deltas (scales derivable; reference test_delta.py:26-35) -------------------------------------------
def test_delta_scales():
    s = orc.delta_scales(2, 2)
    np.testing.assert_allclose(s[1], np.array([-2, -1, 0, 1, 2]) / 10, rtol=1e-7)
    np.testing.assert_allclose(
        s[2], np.array([4, 4, 1, -4, -10, -4, 1, 4, 4]) / 100, rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('order, window', [(0, 1), (1, 2), (2, 2), (5, 5)])
def test_deltas(order, window):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((30, 4)).astype(np.float32)
    d = orc.deltas(x, order, window)
    assert d.shape == (30, 4 * (order + 1))
    assert np.array_equal(d[:, :4], x)
    if order >= 1:
        idx = np.clip(np.arange(30)[:, None] + np.arange(-window, window + 1)[None, :], 0, 29)
        sc = np.arange(-window, window + 1) / (2 * sum(j * j for j in range(1, window + 1)))
        want = (x[idx] * sc[None, :, None]).sum(axis=1)
        np.testing.assert_allclose(d[:, 4:8], want, rtol=1e-5, atol=1e-6)


# ---- fixtures generated from the reference's runnable numpy code ----------------------------------------
def test_rasta_golden(golden):
    """RastaFilter.filter (reference plp.py:101-146) incl. the 4-frame warm-up"""
    got = orc.rasta(golden['rasta_in'], do_log=True)
    np.testing.assert_allclose(got, golden['rasta_out'], rtol=3e-6)
    assert np.all(got[:4] == 1.0)
    got = orc.rasta(golden['rasta_in'][:3], do_log=True)
    assert np.array_equal(got, golden['rasta_short_out'])
    got = orc.rasta(golden['rasta_nolog_in'], do_log=False)
    np.testing.assert_allclose(got, golden['rasta_nolog_out'], rtol=1e-6, atol=1e-7)


def test_lpc2cepstrum_golden(golden):
    for lpc, cep in zip(golden['lpc_in'], golden['lpc_out']):
        assert np.array_equal(orc.lpc2cepstrum(lpc), cep)


# ---- pitch: structure pinned by the reference (shapes) and by Kaldi's constants ---------------------------
def test_pitch_structure(wave):
    po = _abi.default_pitch_options()
    lags, first_lag, last_lag = orc.pitch_lags(po)
    assert (len(lags), first_lag, last_lag) == (417, 8, 82)
    assert orc.pitch_num_frames(po, 22713) == 140             # test_pitch_kaldi.py:39-47
    po2 = _abi.default_pitch_options()
    po2.frame_shift_ms = 20
    assert orc.pitch_num_frames(po2, 22713) == 70
    po2.frame_length_ms = 50
    assert orc.pitch_num_frames(po2, 22713) == 69
    down = orc.linear_resample(wave.astype(np.float32), 16000, 4000, 1000.0, 1)
    assert down.shape == (5679,)
    raw = orc.pitch(po, wave)
    assert raw.shape == (140, 2)
    assert np.all((raw[:, 1] >= 50) & (raw[:, 1] <= 400)) and np.all(np.abs(raw[:, 0]) <= 1.01)


def _click_utterance(n, seed=3):
    """a quiet utterance that ends in a loud click: the two samples the resampler emits at the flush change
    the mean square of the whole signal by far more than 1 %"""
    rng = np.random.default_rng(seed)
    t = np.arange(n)
    w = (rng.normal(0, 12, n) + 40 * np.sin(2 * np.pi * 140 * t / 16000)).astype(np.int16)
    w[-24:] = (20000 * np.sin(2 * np.pi * 300 * np.arange(24) / 16000 + 0.5)).astype(np.int16)
    return w


@pytest.mark.parametrize('nsamples, nframes', [(80240, 500), (80400, 501), (80560, 502), (80720, 503)])
def test_pitch_recompute_backtraces_corner(nsamples, nframes):
    """[KALDI-UPSTREAM] pitch-functions.cc: RecomputeBacktraces runs when frame recompute_frame - 1 (499) has
    been processed.  In the offline call the first AcceptWaveform covers the frames available before the flush
    (T1 = T - 3) and InputFinished adds the rest, so for utterances of 500 - 502 frames frame 499 arrives in
    the SECOND call, with the final signal statistics: frames 0 .. T1 - 1 are rescaled exactly as
    InputFinished rescales a shorter utterance.  From 503 frames on frame 499 arrives in the first call and
    nothing is recomputed."""
    wave = _click_utterance(nsamples)
    po = _abi.default_pitch_options()
    assert po.recompute_frame == 500 and orc.pitch_num_frames(po, nsamples) == nframes
    default = orc.pitch(po, wave)
    po.recompute_frame = 100000   # every utterance is "short": InputFinished recomputes all frames
    all_frames = orc.pitch(po, wave)
    po.recompute_frame = 1        # frame 0 arrives in the first call: never recomputed
    never = orc.pitch(po, wave)
    assert not np.array_equal(all_frames, never)      # the click matters
    if nframes <= 502:
        np.testing.assert_array_equal(default, all_frames)
    else:
        np.testing.assert_array_equal(default, never)


def pitch_oracle_vs_f64(waves):
    """(max |resampled NCCF oracle - float64|, frames whose Viterbi state differs, frames, largest state
    distance) of the C oracle against oracle/spec_f64.pitch over `waves`"""
    po = _abi.default_pitch_options()
    worst, differ, total, step = 0.0, 0, 0, 0
    for w in waves:
        ref = spec_f64.pitch(w)
        out, _, res, _, states = orc.pitch_debug(po, w)
        worst = max(worst, float(np.abs(res - ref['nccf']).max()))
        differ += int((states != ref['states']).sum())
        step = max(step, int(np.abs(states - ref['states']).max()))
        total += states.shape[0]
        same = states == ref['states']
        # where the path is the same the outputs are the same quantities in two precisions
        np.testing.assert_allclose(out[same, 1], ref['out'][same, 1], rtol=1e-6)
        np.testing.assert_allclose(out[same, 0], ref['out'][same, 0], atol=2e-5)
    return worst, differ, total, step


def test_pitch_against_float64_restatement(wave):
    """THE parity statement of the pitch tracker (the GPU tracker equals the C oracle bit for bit because the
    oracle fixes summation orders a wavefront reproduces - that equality says nothing about the distance to
    the exact arithmetic; this test does).  oracle/spec_f64.pitch restates LinearResample, the NCCF with
    ballast, ArbitraryResample and a FULL-search Viterbi in float64.  Measured (profiles/r03_pitch_f64.txt):
    test.wav 2.9e-7 / 0 of 140 frames; 50 synthetic 2 s utterances 9.0e-8 / 5 of 9 900 frames, each one state
    (0.5 % of the pitch value) off where two paths tie within float32 round-off."""
    worst, differ, total, step = pitch_oracle_vs_f64([wave])
    assert worst < 1e-6 and differ == 0 and total == 140
    waves = [synth.utterances(900 + i, 1, 32000)[0] for i in range(50)]
    worst, differ, total, step = pitch_oracle_vs_f64(waves)
    assert total == 9900
    assert worst < 5e-7, worst                 # measured 9.0e-8
    assert differ <= total // 200, differ      # measured 5 (0.05 %); bound 0.5 %
    assert step <= 2, step                     # measured 1


def test_pitch_tracks_a_tone():
    """A 5-harmonic 150 Hz tone in light noise must be tracked within one lag step"""
    from shennong_amd import synth
    rng = np.random.default_rng(0)
    t = np.arange(32000) / 16000
    x = sum(np.sin(2 * np.pi * h * 150.0 * t) / h for h in range(1, 6)) * 6000
    wave = np.clip(x + rng.normal(0, 200, t.shape), -32767, 32767).astype(np.int16)
    raw = orc.pitch(_abi.default_pitch_options(), wave)
    assert np.median(np.abs(raw[5:-5, 1] / 150.0 - 1)) < 0.01
    assert np.median(raw[5:-5, 0]) > 0.9
    post = _abi.default_pitch_post_options()
    post.delta_pitch_noise_stddev = 0
    feats = orc.process_pitch(post, raw)
    assert feats.shape == (raw.shape[0], 3) and np.all(np.isfinite(feats))


# ---- energy, VAD, CMVN -----------------------------------------------------------
def _mfcc_oracle(wave, **kw):
    o = _abi.default_options(_abi.KIND_MFCC)
    o.frame.dither = 0.0
    for k, v in kw.items():
        setattr(o, k, v)
    return orc.compute(o, wave)


def _energy_oracle(wave, raw_energy=True, compression='log'):
    o = _abi.default_options(_abi.KIND_ENERGY)
    o.frame.dither = 0.0
    o.raw_energy = int(raw_energy)
    o.compression = _abi.COMPRESSION[compression]
    return orc.compute(o, wave)


def test_vad_known_answer(wave):
    """reference postprocessor/vad.py:55-56 doctest: 119 voiced frames out of 140 on test.wav"""
    mfcc = _mfcc_oracle(wave)
    vad = orc.vad_energy(mfcc)
    assert vad.shape == (140,) and int(vad.sum()) == 119
    assert set(np.unique(vad)) <= {0.0, 1.0}
    # reference test/postprocessor/test_vad.py:54-62
    assert np.all(orc.vad_energy(mfcc, energy_threshold=0) == 1)
    assert not np.any(orc.vad_energy(mfcc, energy_threshold=1e10))


@pytest.mark.parametrize('raw_energy', [True, False])
def test_energy_is_first_cepstral_coefficient(wave, raw_energy):
    """reference test/processor/test_energy.py:40-48 (np.allclose defaults) and
    test/postprocessor/test_vad.py:70-77"""
    mfcc = _mfcc_oracle(wave, raw_energy=int(raw_energy))
    plp = orc.compute(_plp_opts(raw_energy), wave)
    energy = _energy_oracle(wave, raw_energy)
    assert energy.shape == (140, 1)
    assert np.allclose(mfcc[:, 0], energy[:, 0])
    assert np.allclose(plp[:, 0], energy[:, 0])
    assert np.array_equal(orc.vad_energy(energy), orc.vad_energy(mfcc))


def _plp_opts(raw_energy):
    o = _abi.default_options(_abi.KIND_PLP)
    o.frame.dither = 0.0
    o.raw_energy = int(raw_energy)
    return o


def test_energy_compression(wave):
    """reference processor/energy.py:30-33 doctest: log(off) == log"""
    off = _energy_oracle(wave, compression='off')
    np.testing.assert_allclose(np.log(off), _energy_oracle(wave), rtol=1e-6)
    np.testing.assert_allclose(np.sqrt(off), _energy_oracle(wave, compression='sqrt'), rtol=1e-6)


@pytest.mark.parametrize('norm_vars', [True, False])
def test_cmvn_oracle(wave, norm_vars):
    """reference test/postprocessor/test_cmvn.py:41-84 and the cmvn.py:38-43 doctest"""
    mfcc = _mfcc_oracle(wave)
    stats = orc.cmvn_accumulate(mfcc)
    assert stats.shape == (2, 14) and stats[0, -1] == 140 and stats[1, -1] == 0
    np.testing.assert_allclose(stats[0, :13], mfcc.astype(np.float64).sum(axis=0), rtol=1e-12)
    np.testing.assert_allclose(stats[1, :13], (mfcc.astype(np.float64) ** 2).sum(axis=0), rtol=1e-7)
    out = orc.cmvn_apply(mfcc, stats, norm_vars=norm_vars)
    assert np.all(np.isclose(out.mean(axis=0), 0, atol=1e-6))
    if norm_vars:
        assert np.all(np.isclose(out.var(axis=0), 1, atol=1e-6))
    else:
        assert out.var(axis=0) == pytest.approx(mfcc.var(axis=0))
    back = orc.cmvn_apply(out, stats, norm_vars=norm_vars, reverse=True)
    assert back == pytest.approx(mfcc, abs=1e-5)
    twice = orc.cmvn_accumulate(mfcc, stats=stats.copy())
    assert twice == pytest.approx(stats * 2)
    # weights (test_cmvn.py:101-122)
    assert orc.cmvn_accumulate(mfcc, weights=np.zeros(140))[0, -1] == 0
    assert orc.cmvn_accumulate(mfcc, weights=np.ones(140) * 0.5)[0, -1] == 70
    w = np.zeros(140)
    w[:2] = 0.1
    assert orc.cmvn_accumulate(mfcc, weights=w)[0, -1] == pytest.approx(0.2)


@pytest.mark.parametrize('norm_vars, center', [(s, v) for s in (True, False) for v in (True, False)])
def test_sliding_cmvn_oracle(wave, norm_vars, center):
    """reference test/postprocessor/test_cmvn.py:223-263"""
    mfcc = _mfcc_oracle(wave)
    ws = 40
    out = orc.sliding_cmn(mfcc, center=center, cmn_window=ws, min_window=ws,
                          normalize_variance=norm_vars)
    frame = 70
    a, b = (frame - ws // 2, frame + ws // 2) if center else (frame - ws, frame + 1)
    want = mfcc[frame] - mfcc[a:b].mean(axis=0)
    if norm_vars:
        want = want / mfcc[a:b].std(axis=0)
    assert np.all(np.isclose(out[frame], want, atol=1e-6))


# ---- closed-form known answers (tests/known_answers.py): no Kaldi, no code of this repository in
# the expected values; the same cases run against the HIP path in tests/test_parity_gpu.py ----------
import known_answers  # noqa: E402


@pytest.mark.parametrize('case', known_answers.CASES, ids=[c[0] for c in known_answers.CASES])
def test_known_answer(case):
    _, make, wave, check = case
    check(orc.compute(make()._build_options(), wave))


def test_known_answer_mel_partition_of_unity():
    for nb, low, high in ((40, 20.0, 0.0), (23, 20.0, 0.0), (13, 100.0, -500.0)):
        mo = _abi.default_mel_options()
        mo.num_bins, mo.low_freq, mo.high_freq = nb, low, high
        first, size, weights, _ = orc.mel_banks(mo, _frame_opts())
        known_answers.mel_partition_check(
            first, [weights[b, first[b]:first[b] + size[b]] for b in range(nb)])


def test_known_answer_dct_orthonormal():
    for n in (13, 23, 40):
        m = orc.dct_matrix(n, n).astype(np.float64)
        np.testing.assert_allclose(m @ m.T, np.eye(n), atol=3e-7)


def test_known_answer_parseval():
    from shennong_amd.processor import FilterbankProcessor, MfccProcessor
    wave = scipy_wave()
    fbank = orc.compute(FilterbankProcessor(num_bins=23, dither=0)._build_options(), wave)
    mfcc = orc.compute(MfccProcessor(num_ceps=23, cepstral_lifter=0, use_energy=False,
                                     dither=0)._build_options(), wave)
    known_answers.parseval_check(fbank, mfcc)


def test_known_answer_delta_ramp():
    known_answers.ramp_delta_check(orc.deltas)


def scipy_wave():
    import scipy.io.wavfile
    return scipy.io.wavfile.read(os.path.join(GOLDEN, 'test.wav'))[1]


def test_linear_resample_window_in_ticks():
    """[KALDI-UPSTREAM] resample.cc LinearResample::GetNumOutputSamples: `BaseFloat window_width = num_zeros_ /
    (2.0 * filter_cutoff_); int32 window_width_ticks = floor(window_width * tick_freq);` - the product is a FLOAT
    product.  A cutoff of 800 Hz at 16 kHz -> 4 kHz (tick rate 16 000): the width is 0.000624999986 as a float, times
    16 000 = 9.99999978 exactly and 10.0f as a float: 10 ticks.  16 002 input samples without flush: 15 992 ticks = 4 x
    3 998 exactly -> outputs 0 .. 3 997 (with 9 ticks there would be 3 999).  Found by
    tests/tools/fuzz_oracle_f64_pitch.py in round 4: until then the oracle and the product formed that product in
    double; the default cutoff (1 000 Hz: 8.0000004 -> 8 either way) never showed it."""
    wave = synth.utterances(3, 1, 16002)[0]
    assert orc.linear_resample(wave, 16000, 4000, 800.0, 1, False).shape == (3998,)
    assert spec_f64.linear_resample(wave, 16000, 4000, 800.0, 1, False).shape == (3998,)
    assert orc.linear_resample(wave, 16000, 4000, 1000.0, 1, False).shape == (3999,)


def test_random_pitch_option_sets(monkeypatch):
    """tests/tools/fuzz_oracle_f64_pitch.py: random option sets of the pitch tracker, the C oracle against the float64
    restatement with a full-search Viterbi (800 cases in profiles/r04_f64_report.txt; a short run here)"""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'fuzz_oracle_f64_pitch.py')
    spec = importlib.util.spec_from_file_location('fuzz_oracle_f64_pitch', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr('sys.argv', ['fuzz_oracle_f64_pitch.py', '12', '21'])
    assert mod.main() == 0
