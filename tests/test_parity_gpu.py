"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Tolerance: rtol 1e-4 (BASELINE.json north_star) + atol 2e-3 on log-domain outputs, frame counts and
shapes bit-exact; delta is compared at 1e-5; pitch frame-by-frame (see test_pitch)."""

import itertools
import os

import numpy as np
import pytest

from conftest import assert_close
from oracle import oracle as orc
from shennong_amd import Audio, _abi, _backend, synth
from shennong_amd.processor import (
    FilterbankProcessor, MfccProcessor, PlpProcessor, SpectrogramProcessor,
    KaldiPitchProcessor, KaldiPitchPostProcessor)
from shennong_amd.postprocessor import DeltaPostProcessor

pytestmark = pytest.mark.gpu


def _oracle(proc, wave, warp=1.0):
    return orc.compute(proc._build_options(), np.asarray(wave, np.int16), warp)


@pytest.mark.parametrize('num_bins', [23, 40])
@pytest.mark.parametrize('use_energy', [False, True])
def test_fbank_testwav(gpu, audio, wave, num_bins, use_energy):
    proc = FilterbankProcessor(num_bins=num_bins, use_energy=use_energy, dither=0)
    got = proc.process(audio)
    want = _oracle(proc, wave)
    assert got.shape == (140, num_bins + use_energy)
    assert_close(got.data, want, what='fbank')


@pytest.mark.parametrize('opts', [
    dict(raw_energy=False, use_energy=True),
    dict(htk_compat=True, use_energy=True),
    dict(use_log_fbank=False),
    dict(use_power=False),
    dict(energy_floor=1e9, use_energy=True),
    dict(window_type='hamming'), dict(window_type='hanning'),
    dict(window_type='rectangular'), dict(window_type='blackman'),
    dict(preemph_coeff=0.0), dict(remove_dc_offset=False),
    dict(snip_edges=False),
    dict(frame_shift=0.02, frame_length=0.05),
    dict(low_freq=100, high_freq=-200),
    dict(round_to_power_of_two=False),
])
def test_fbank_options(gpu, audio, wave, opts):
    proc = FilterbankProcessor(dither=0, **opts)
    got = proc.process(audio)
    want = _oracle(proc, wave)
    rtol = 1e-4
    # (linear mel energies are sums of squares of int16-scale samples, ~1e7: the absolute term scales with them)
    assert_close(got.data, want, rtol=rtol, atol=None if opts.get('use_log_fbank', True) else 1.0,
                 what=str(opts), family='fbank')


@pytest.mark.parametrize('opts', [
    dict(), dict(use_energy=False), dict(raw_energy=False),
    dict(htk_compat=True), dict(htk_compat=True, use_energy=False),
    dict(num_ceps=5), dict(num_ceps=23), dict(cepstral_lifter=0.0),
    dict(num_bins=40, num_ceps=20), dict(snip_edges=False),
])
def test_mfcc(gpu, audio, wave, opts):
    proc = MfccProcessor(dither=0, **opts)
    got = proc.process(audio)
    assert_close(got.data, _oracle(proc, wave), what=str(opts), family='mfcc')


@pytest.mark.parametrize('opts', [dict(), dict(raw_energy=False),
                                  dict(energy_floor=1e9), dict(snip_edges=False)])
def test_spectrogram(gpu, audio, wave, opts):
    proc = SpectrogramProcessor(dither=0, **opts)
    got = proc.process(audio)
    want = _oracle(proc, wave)
    assert got.shape[1] == 257
    # single-bin log power: deep spectral nulls amplify float32 FFT round-off of both sides
    assert_close(got.data, want, rtol=1e-4, what=str(opts), family='spectrogram')  # (spectral nulls)
    assert np.mean(np.abs(got.data - want) < 1e-3) > 0.999


@pytest.mark.parametrize('opts', [
    dict(), dict(use_energy=False), dict(raw_energy=False), dict(htk_compat=True),
    dict(rasta=True), dict(rasta=True, snip_edges=False), dict(num_ceps=5),
    dict(cepstral_lifter=0, cepstral_scale=0.9), dict(lpc_order=8, num_ceps=9),
])
def test_plp(gpu, audio, wave, opts):
    proc = PlpProcessor(dither=0, **opts)
    got = proc.process(audio)
    want = _oracle(proc, wave)
    assert_close(got.data, want, rtol=1e-4, what=str(opts), family='plp')


@pytest.mark.parametrize('warp', [0.85, 1.0, 1.2])
@pytest.mark.parametrize('cls', [FilterbankProcessor, MfccProcessor, PlpProcessor])
def test_vtln_warp(gpu, audio, wave, cls, warp):
    proc = cls(dither=0)
    got = proc.process(audio, vtln_warp=warp)
    assert_close(got.data, _oracle(proc, wave, warp), rtol=1e-4, what=f'{proc.name} warp {warp}')
    assert got.properties[proc.name]['vtln_warp'] == warp


@pytest.mark.parametrize('sample_rate, frame_length', [
    (8000, 0.025), (16000, 0.064), (44100, 0.025), (16000, 0.005)])
def test_other_fft_sizes(gpu, sample_rate, frame_length):
    n = int(0.4 * sample_rate)
    wave = synth.utterances(7, 1, n, sample_rate)[0]
    proc = MfccProcessor(sample_rate=sample_rate, frame_length=frame_length, dither=0)
    got = proc.process(Audio(wave, sample_rate))
    assert_close(got.data, _oracle(proc, wave), what=f'mfcc {sample_rate} {frame_length}')


def test_batch_ragged(gpu, synth_waves):
    """process_all = one launch over ragged utterances, including one too short for any frame"""
    from shennong_amd import Utterances
    waves = list(synth_waves) + [np.zeros(100, np.int16)]
    utts = Utterances([(f'u{i}', Audio(w, 16000)) for i, w in enumerate(waves)])
    warps = {f'u{i}': [1.0, 0.9, 1.1][i % 3] for i in range(len(waves))}
    proc = FilterbankProcessor(num_bins=40, dither=0)
    feats = proc.process_all(utts, vtln_warp=warps)
    assert list(feats.keys()) == [f'u{i}' for i in range(len(waves))]
    for i, w in enumerate(waves):
        f = feats[f'u{i}']
        want = _oracle(proc, w, warps[f'u{i}'])
        assert f.shape == want.shape
        if want.size:
            assert_close(f.data, want, rtol=1e-4, what=f'{proc.name} utt {i}')
    assert feats[f'u{len(waves) - 1}'].shape == (0, 0)


@pytest.mark.parametrize('order, window', itertools.product([0, 1, 2, 5], [1, 2, 5]))
def test_delta(gpu, audio, order, window):
    mfcc = MfccProcessor(dither=0).process(audio)
    got = DeltaPostProcessor(order=order, window=window).process(mfcc)
    want = orc.deltas(mfcc.data, order, window)
    assert got.shape == (140, 13 * (order + 1))
    assert_close(got.data, want, rtol=1e-5, family='delta')
    assert np.array_equal(got.data[:, :13], mfcc.data)


def test_delta_batch_edges(gpu):
    rng = np.random.default_rng(3)
    mats = [rng.standard_normal((n, 7)).astype(np.float32) for n in (1, 2, 3, 9, 40)]
    from shennong_amd import Features
    feats = [Features(m, np.arange(m.shape[0], dtype=np.float64)) for m in mats]
    outs = DeltaPostProcessor()._process_batch(feats)
    for m, o in zip(mats, outs):
        assert_close(o.data, orc.deltas(m, 2, 2), rtol=1e-5, family='delta')


def _pitch_close(got, want):
    """The tracker implements the oracle's summation orders (oracle/kaldi_oracle.c chain_dot / tree16):
    every frame - Viterbi state, pitch and POV NCCF - is bit-identical"""
    assert got.shape == want.shape
    np.testing.assert_array_equal(got[:, 1], want[:, 1])
    np.testing.assert_array_equal(got[:, 0], want[:, 0])


@pytest.fixture(params=[1, 2, 4])
def vit_team(request, monkeypatch):
    """every form of the Viterbi kernel: one wave per utterance (what large batches run on), or a team of two /
    four waves per utterance (small batches; kernels_pitch.hip viterbi_forward_team) - all bit-identical"""
    monkeypatch.setenv('SNF_PITCH_TEAM', str(request.param))
    return request.param


@pytest.mark.parametrize('opts', [dict(), dict(frame_shift=0.02),
                                  dict(frame_shift=0.02, frame_length=0.05),
                                  dict(min_f0=60, max_f0=350, penalty_factor=0.2)])
def test_pitch(gpu, audio, wave, opts, vit_team):
    proc = KaldiPitchProcessor(**opts)
    got = proc.process(audio)
    want = orc.pitch(proc._options, wave)
    _pitch_close(got.data, want)


def test_pitch_corpus_longer_than_one_call(gpu, synth_waves, monkeypatch):
    """a pitch batch whose tracker scratch (2.3 GB per hour of audio, per plan) would not fit goes through whole
    calls one after the other - here with the limit lowered to one second of audio: arrays and a pinned corpus,
    utterances longer than the limit included; the rows are those of one call"""
    from shennong_amd import Utterances
    waves = list(synth_waves)          # six utterances of 0.3 .. 1.2 s
    proc = KaldiPitchProcessor()
    index = Utterances([(f'u{i}', Audio(w, 16000, validate=False)) for i, w in enumerate(waves)])
    want = proc.process_all(index)
    monkeypatch.setattr(_backend, '_MAX_TRACKER_HOURS', 1.0 / 3600.0)
    plan = _backend.get_plan(proc._build_options())
    cuts = plan._tracker_chunks(np.concatenate([[0], np.cumsum([w.shape[0] for w in waves])]).astype(np.int64))
    assert cuts is not None and len(cuts) > 3 and cuts[0] == 0 and cuts[-1] == len(waves)
    for got in (proc.process_all(index), proc.process_all(index.pin())):
        assert list(got) == list(want)
        assert all(np.array_equal(got[k].data, want[k].data) for k in want)


def test_pitch_negative_penalty_is_refused(gpu, audio):
    """the searches order costs by their bit pattern (non-negative floats only): a transition cost
    that could go negative is an error at plan creation, not a silently different track"""
    with pytest.raises(ValueError, match='penalty_factor must be >= 0'):
        KaldiPitchProcessor(penalty_factor=-0.1).process(audio)
    assert KaldiPitchProcessor(penalty_factor=0.0).process(audio).shape[1] == 2


@pytest.mark.parametrize('opts', [
    dict(min_f0=40),                   # 461 states (the 8-slice register form), 95 lags: two correlation
                                       # passes, NCCF beside the window instead of over it
    dict(delta_pitch=0.004),           # 521 states: the rows are read where they are used (no register form)
    dict(upsample_filter_width=7),     # 14 sinc taps per state: 16-lag quad windows (run-time step count)
    dict(min_f0=70, max_f0=300, delta_pitch=0.01, lowpass_cutoff=800, resample_freq=3200),
])
def test_pitch_option_paths(gpu, synth_waves, opts, vit_team):
    """option sets that leave the instantiations the default configuration runs on (kernels_pitch.hip:
    viterbi_forward<7> / <8> / <0>, the overlaid NCCF, the straight-line 12-lag resampling)"""
    proc = KaldiPitchProcessor(**opts)
    waves = list(synth_waves)[:3]
    outs = proc._process_batch([Audio(w, 16000) for w in waves])
    for w, o in zip(waves, outs):
        _pitch_close(o.data, orc.pitch(proc._options, w))


def test_pitch_recompute_backtraces_corner(gpu, vit_team):
    """utterances of 500 - 503 pitch frames whose last samples move the mean square (the T1 < 500 <= T corner
    of Kaldi's RecomputeBacktraces, tests/test_oracle_pins.py): every frame equals the oracle, in one batch
    and for the settings of recompute_frame that force / forbid the second pass"""
    from test_oracle_pins import _click_utterance
    waves = [_click_utterance(n) for n in (80240, 80400, 80560, 80720, 40000)]
    for rf in (500, 100000, 1):
        opts = KaldiPitchProcessor()._build_options()
        opts.pitch.recompute_frame = rf
        outs = _backend.get_plan(opts).run(waves)
        for w, o in zip(waves, outs):
            _pitch_close(o, orc.pitch(opts.pitch, w))


def test_pitch_batch(gpu, synth_waves, vit_team):
    proc = KaldiPitchProcessor()
    outs = proc._process_batch([Audio(w, 16000) for w in synth_waves])
    for w, o in zip(synth_waves, outs):
        _pitch_close(o.data, orc.pitch(proc._options, w))


@pytest.mark.parametrize('flags', [(1, 1, 1, 1), (1, 1, 1, 0), (0, 0, 1, 1), (0, 1, 0, 0)])
def test_pitch_post(gpu, wave, flags):
    raw = orc.pitch(_abi.default_pitch_options(), wave)
    from shennong_amd import Features
    proc = KaldiPitchPostProcessor(
        delta_pitch_noise_stddev=0, add_pov_feature=flags[0],
        add_normalized_log_pitch=flags[1], add_delta_pitch=flags[2],
        add_raw_log_pitch=flags[3])
    times = KaldiPitchProcessor().times(raw.shape[0])
    feats = Features(raw, times, properties={
        'pipeline': [{'name': 'pitch', 'columns': [0, 1]}], 'pitch': {}})
    got = proc.process(feats)
    want = orc.process_pitch(proc._options, raw)
    assert_close(got.data, want, rtol=1e-4, family='pitch_post')


def test_full_size_properties(gpu):
    """At BASELINE.json's full utterance size: bit stability run-to-run, batch == single, and
    linearity of the (linear) mel energies: fbank(2 x) = 4 fbank(x)."""
    waves = synth.utterances(42, 8, 48000)
    proc = FilterbankProcessor(num_bins=40, dither=0, use_log_fbank=False,
                               remove_dc_offset=False)
    a = proc._process_batch([Audio(w, 16000) for w in waves])
    b = proc._process_batch([Audio(w, 16000) for w in waves])
    for x, y in zip(a, b):
        assert x.shape == (298, 40)
        assert x == y
    single = proc.process(Audio(waves[3], 16000))
    assert np.array_equal(single.data, a[3].data)
    half = (waves[0] // 2).astype(np.int16)
    f1 = proc.process(Audio(half, 16000)).data
    f2 = proc.process(Audio((half * 2).astype(np.int16), 16000)).data
    np.testing.assert_allclose(f2, 4 * f1, rtol=2e-5)


@pytest.fixture(scope='module')
def full_workload():
    """BASELINE.json configs[1]: 10 000 unique synthetic 16 kHz 3 s utterances (the bench batch)"""
    from concurrent.futures import ProcessPoolExecutor
    jobs = [(1000 + i, 625, 48000) for i in range(0, 10000, 625)]
    with ProcessPoolExecutor(min(16, os.cpu_count() or 1)) as pool:
        parts = list(pool.map(_synth_job, jobs))
    return np.concatenate(parts, axis=0)


def _synth_job(args):
    return synth.utterances(*args)


@pytest.mark.parametrize('kind', ['fbank40', 'mfcc13', 'mfcc13_delta', 'plp13', 'spectrogram257'])
def test_full_workload_every_frame(gpu, full_workload, kind):
    """The whole bench workload (2.98 M frames), EVERY frame against the oracle, through the same
    batched entry point the bench times; then order independence: the batch reversed gives the same
    rows bit for bit.  `spectrogram257` (the other half of BASELINE config 2; 3.06 GB of output) is compared
    500 utterances at a time."""
    waves = full_workload
    proc = (FilterbankProcessor(num_bins=40, dither=0) if kind == 'fbank40'
            else PlpProcessor(dither=0) if kind == 'plp13'
            else SpectrogramProcessor(dither=0) if kind == 'spectrogram257' else MfccProcessor(dither=0))
    opts = proc._build_options()
    plan = _backend.get_plan(opts)
    got = plan.run(list(waves))
    assert len(got) == 10000 and all(g.shape == (298, plan.ndims) for g in got[::997])
    if kind == 'spectrogram257':
        assert plan.ndims == 257
        for a in range(0, 10000, 500):
            want = orc.compute_batch(opts, waves[a:a + 500], os.cpu_count() or 1)
            assert_close(np.concatenate(got[a:a + 500]), want, what=f'spectrogram utterances {a}..{a + 499}')
        rev = plan.run(list(waves[:2000][::-1]))[::-1]   # (order independence on a fifth of it)
        assert all(np.array_equal(r, g) for r, g in zip(rev, got[:2000]))
        return
    got = np.concatenate(got)
    want = orc.compute_batch(opts, waves, os.cpu_count() or 1)
    if kind == 'mfcc13_delta':
        foff = np.arange(10001, dtype=np.int64) * 298
        dproc = DeltaPostProcessor(order=2)
        dplan = _backend.get_plan(dproc._build_options())
        got39 = np.concatenate(dplan.run_post(list(got.reshape(10000, 298, 13))))
        # the delta kernel on the GPU's own MFCC vs the oracle's delta of the same matrix: exact
        # up to float32 summation order; edge clamping is per utterance
        for u in (0, 1, 4999, 9999):
            ref = orc.deltas(got[foff[u]:foff[u + 1]], 2, 2)
            assert_close(got39[foff[u]:foff[u + 1]], ref, rtol=1e-5, what=f'delta {u}')
        assert np.array_equal(got39[:, :13], got)
        return
    assert got.shape == want.shape == (2980000, plan.ndims)
    assert_close(got, want, what=kind)
    rev = np.concatenate(plan.run(list(waves[::-1]))).reshape(10000, 298, -1)[::-1]
    assert np.array_equal(rev.reshape(got.shape), got)


@pytest.mark.parametrize('snip_edges', [True, False])  # fast 512-point kernel / generic kernel
def test_dither_statistics(gpu, snip_edges):
    """Dither cannot be bit-compared (Kaldi draws from C rand()); it must be N(0, dither^2) per
    sample and per frame: on a silent signal the raw frame energy is (L-1) chi2-distributed."""
    wave = np.zeros(48000, dtype=np.int16)
    proc = MfccProcessor(dither=1.0, snip_edges=snip_edges)
    a = proc.process(Audio(wave, 16000)).data
    e = a[:, 0]  # log sum (x - mean)^2 over 400 samples of unit-variance noise
    assert abs(e.mean() - np.log(399.0)) < 0.03, e.mean()
    assert 0.04 < e.std() < 0.11, e.std()
    # overlapping frames draw independent noise (Kaldi dithers every extracted window)
    assert abs(np.corrcoef(e[:-1], e[1:])[0, 1]) < 0.25
    # every call draws its own noise (the reference's rand() stream never repeats either): same
    # statistics, different samples, and no correlation between two calls on the same signal
    b = proc.process(Audio(wave, 16000)).data
    assert not np.array_equal(a, b)
    assert abs(b[:, 0].mean() - np.log(399.0)) < 0.03
    assert abs(np.corrcoef(e, b[:, 0])[0, 1]) < 0.25
    # a loud signal is barely affected (reference test_parallel.py: is_close(atol=10))
    loud = synth.utterances(3, 1, 16000)[0]
    clean = MfccProcessor(dither=0, snip_edges=snip_edges).process(Audio(loud, 16000)).data
    noisy = proc.process(Audio(loud, 16000)).data
    assert np.abs(clean - noisy).max() < 0.05


def test_dither_is_gaussian(gpu):
    """The generator behind the default dither (round 5: a multiply-fold hash + Box-Muller on mantissa-built
    uniforms, csrc/device_fft.h) against what N(0, 1) predicts on a long silent signal.  With a rectangular
    window, no pre-emphasis and no DC removal the raw energy of a frame is the sum of 400 squared normals:
    mean 400, variance 400 (kurtosis - 1) = 800, and log-energy std sqrt(2 / 400); the spectrogram of white
    noise is flat: every bin of the power spectrum has the same expectation (400 x the window's energy)."""
    wave = np.zeros(16000 * 60, dtype=np.int16)
    proc = SpectrogramProcessor(dither=1.0, window_type='rectangular', preemph_coeff=0.0,
                                remove_dc_offset=False, raw_energy=True)
    spec = proc.process(Audio(wave, 16000)).data
    n = spec.shape[0]
    assert n > 5900
    energy = np.exp(spec[:, 0].astype(np.float64))          # sum of 400 squared samples
    assert abs(energy.mean() / 400.0 - 1.0) < 0.004          # variance of the samples: 1 (+- 3 sigma of the mean)
    kurt = energy.var() / 400.0 + 1.0                        # E x^4 / sigma^4 of the samples
    assert 2.8 < kurt < 3.2, kurt                            # (std of this estimate: ~0.04)
    assert abs(np.corrcoef(energy[:-1], energy[1:])[0, 1]) < 0.05   # overlapping frames: independent draws
    power = np.exp(spec[:, 1:].astype(np.float64)).mean(axis=0)     # bins 1..256: E |X_k|^2 = 400
    assert np.abs(power / 400.0 - 1.0).max() < 0.08, (power.min(), power.max())
    # two halves of the spectrum and even / odd bins agree (no periodic structure in the stream)
    assert abs(power[:128].mean() / power[128:].mean() - 1.0) < 0.01
    assert abs(power[0::2].mean() / power[1::2].mean() - 1.0) < 0.01


@pytest.mark.parametrize('sample_rate, kernel', [(32000, 'fbank1024x2_kernel'), (22050, 'fbank1024x2_kernel'),
                                                 (44100, 'fbank2048_kernel')])
def test_dither_is_gaussian_long_frames(gpu, sample_rate, kernel):
    """the same generator in the long-frame kernels (the 1024-sample kernel draws the normals of a frame pairwise
    over its rows): unit variance, kurtosis 3, independent frames, a flat spectrum - and two frames of a pair
    draw different noise"""
    wave = np.zeros(sample_rate * 20, dtype=np.int16)
    proc = SpectrogramProcessor(sample_rate=sample_rate, dither=1.0, window_type='rectangular', preemph_coeff=0.0,
                                remove_dc_offset=False, raw_energy=True)
    spec = proc.process(Audio(wave, sample_rate)).data
    assert _backend.get_plan(proc._build_options()).kernel_name(1) == kernel
    L = int(0.025 * sample_rate)
    energy = np.exp(spec[:, 0].astype(np.float64))          # sum of L squared samples
    assert abs(energy.mean() / L - 1.0) < 0.006, energy.mean() / L
    kurt = energy.var() / L + 1.0
    assert 2.7 < kurt < 3.3, kurt
    assert abs(np.corrcoef(energy[:-1], energy[1:])[0, 1]) < 0.08
    assert abs(np.corrcoef(energy[0:-1:2], energy[1::2])[0, 1]) < 0.1     # the two frames of a pair
    power = np.exp(spec[:, 1:].astype(np.float64)).mean(axis=0)
    assert np.abs(power / L - 1.0).max() < 0.12, (power.min() / L, power.max() / L)
    half = power.shape[0] // 2
    assert abs(power[:half].mean() / power[half:2 * half].mean() - 1.0) < 0.015
    assert abs(power[0::2].mean() / power[1::2].mean() - 1.0) < 0.015


# ---- energy, VAD, CMVN, sliding-window CMVN -----------------------------------
from shennong_amd import Features, FeaturesCollection  # noqa: E402
from shennong_amd.processor import EnergyProcessor  # noqa: E402
from shennong_amd.postprocessor import (  # noqa: E402
    CmvnPostProcessor, SlidingWindowCmvnPostProcessor, VadPostProcessor, apply_cmvn)


@pytest.mark.parametrize('opts', [
    dict(), dict(raw_energy=False), dict(compression='off'), dict(compression='sqrt'),
    dict(raw_energy=False, window_type='hanning', preemph_coeff=0.5),
    dict(snip_edges=False), dict(frame_shift=0.02, frame_length=0.05),
    dict(round_to_power_of_two=False), dict(remove_dc_offset=False)])
def test_energy(gpu, audio, wave, opts):
    proc = EnergyProcessor(dither=0, **opts)
    got = proc.process(audio)
    want = _oracle(proc, wave)
    assert got.shape == want.shape and got.shape[1] == 1
    np.testing.assert_allclose(got.data, want, rtol=1e-6)
    assert got.properties['energy']['raw_energy'] == opts.get('raw_energy', True)


@pytest.mark.parametrize('raw_energy', [True, False])
def test_energy_matches_c0(gpu, audio, raw_energy):
    """reference test/processor/test_energy.py:40-48"""
    p = {'raw_energy': raw_energy, 'dither': 0}
    mfcc = MfccProcessor(**p).process(audio).data[:, 0]
    plp = PlpProcessor(**p).process(audio).data[:, 0]
    energy = EnergyProcessor(**p).process(audio).data[:, 0]
    assert np.allclose(mfcc, energy) and np.allclose(plp, energy)


def test_energy_shapes(gpu, audio):
    """reference test/processor/test_energy.py:51-56"""
    assert EnergyProcessor(frame_shift=0.01).process(audio).shape == (140, 1)
    assert EnergyProcessor(frame_shift=0.02).process(audio).shape == (70, 1)
    assert EnergyProcessor(frame_shift=0.02, frame_length=0.05).process(audio).shape == (69, 1)


@pytest.mark.parametrize('opts', [
    dict(), dict(frames_context=2), dict(frames_context=5, proportion_threshold=0.3),
    dict(energy_mean_scale=0.0, energy_threshold=12.0), dict(energy_mean_scale=1.0, energy_threshold=0.5)])
def test_vad(gpu, audio, opts):
    mfcc = MfccProcessor(dither=0).process(audio)
    got = VadPostProcessor(**opts).process(mfcc)
    want = orc.vad_energy(mfcc.data, **opts)
    assert got.shape == (140, 1) and got.dtype == np.uint8
    assert np.array_equal(got.data[:, 0], want.astype(np.uint8))
    assert np.array_equal(got.times, mfcc.times)


def test_vad_reference_behaviour(gpu, audio):
    """reference vad.py:55-56 doctest (119 of 140) and test/postprocessor/test_vad.py:43-77"""
    mfcc = MfccProcessor(dither=0).process(audio)
    p = VadPostProcessor()
    vad = p.process(mfcc)
    assert int(vad.data.sum()) == 119
    p.energy_threshold = 0
    assert np.all(p.process(mfcc).data)
    p.energy_threshold = 1e10
    assert not np.any(p.process(mfcc).data)
    vad1 = VadPostProcessor().process(EnergyProcessor(dither=0).process(audio))
    assert Features(vad1.data, vad1.times) == Features(vad.data, vad.times)


def test_vad_batch(gpu, synth_waves):
    proc = MfccProcessor(dither=0)
    feats = proc._process_batch([Audio(w, 16000) for w in synth_waves])
    outs = VadPostProcessor(frames_context=3)._process_batch(feats)
    for f, o in zip(feats, outs):
        want = orc.vad_energy(f.data, frames_context=3)
        assert np.array_equal(o.data[:, 0], want.astype(np.uint8))


@pytest.mark.parametrize('norm_vars', [True, False])
def test_cmvn(gpu, audio, norm_vars):
    """parity with the oracle + the reference's own checks (test/postprocessor/test_cmvn.py:41-84)"""
    mfcc = MfccProcessor(dither=0).process(audio)
    backup = mfcc.data.copy()
    proc = CmvnPostProcessor(mfcc.ndims)
    proc.accumulate(mfcc)
    assert proc.count == mfcc.nframes
    want_stats = orc.cmvn_accumulate(mfcc.data)
    np.testing.assert_allclose(proc.stats, want_stats, rtol=1e-13)
    cmvn1 = proc.process(mfcc, norm_vars=norm_vars)
    # same statistics -> the apply step must be bit-exact
    assert np.array_equal(cmvn1.data, orc.cmvn_apply(mfcc.data, proc.stats, norm_vars=norm_vars))
    assert np.array_equal(backup, mfcc.data)
    assert cmvn1.shape == mfcc.shape and cmvn1.dtype == mfcc.dtype
    assert cmvn1.data.mean() == pytest.approx(0, abs=1e-6)
    if norm_vars:
        assert cmvn1.data.var(axis=0) == pytest.approx(np.ones(cmvn1.ndims))
    else:
        assert cmvn1.data.var(axis=0) == pytest.approx(mfcc.data.var(axis=0))
    cmvn2 = proc.process(cmvn1, norm_vars=norm_vars, reverse=True)
    assert np.array_equal(
        cmvn2.data, orc.cmvn_apply(cmvn1.data, proc.stats, norm_vars=norm_vars, reverse=True))
    assert cmvn2.data == pytest.approx(mfcc.data, abs=1e-5)
    stats = proc.stats.copy()
    proc.accumulate(mfcc)
    assert proc.stats == pytest.approx(stats * 2)
    assert 'cmvn' not in mfcc.properties and cmvn2.properties['cmvn']['stats'].shape == (2, 14)


def test_cmvn_weights_and_skip_dims(gpu, audio):
    """reference test/postprocessor/test_cmvn.py:101-161"""
    mfcc = MfccProcessor(dither=0).process(audio)
    rng = np.random.default_rng(0)
    w = rng.random(mfcc.nframes)
    w[::7] = 0.0
    proc = CmvnPostProcessor(mfcc.ndims)
    proc.accumulate(mfcc, weights=w)
    np.testing.assert_allclose(proc.stats, orc.cmvn_accumulate(mfcc.data, weights=w), rtol=1e-12)
    for weights, count in ((np.zeros(140), 0), (np.ones(140), 140), (np.ones(140) * 0.5, 70)):
        p = CmvnPostProcessor(dim=mfcc.ndims)
        p.accumulate(mfcc, weights=weights)
        assert p.count == count
    proc = CmvnPostProcessor(mfcc.ndims)
    proc.accumulate(mfcc)
    cmvn1 = proc.process(mfcc, skip_dims=None)
    assert cmvn1 == proc.process(mfcc, skip_dims=[])
    cmvn3 = proc.process(mfcc, skip_dims=[0, 1, 2])
    assert np.array_equal(cmvn3.data[:, :3], mfcc.data[:, :3])
    assert np.array_equal(cmvn3.data[:, 3:], cmvn1.data[:, 3:])
    assert proc.process(mfcc, skip_dims=[1, 2, 0]) == cmvn3
    assert np.array_equal(proc.process(mfcc, skip_dims=list(range(13))).data, mfcc.data)
    for d in ([-1], [-1, 2, 3], [100], [100, -1, 5]):
        with pytest.raises(ValueError):
            proc.process(mfcc, skip_dims=d)


@pytest.mark.parametrize('by_collection', [True, False])
def test_apply_cmvn(gpu, synth_waves, by_collection):
    """reference test/postprocessor/test_cmvn.py:164-220; one stats launch + one apply launch"""
    proc = MfccProcessor(dither=0)
    feats = proc._process_batch([Audio(w, 16000) for w in synth_waves])
    coll = FeaturesCollection({str(i): f for i, f in enumerate(feats)})
    cmvns = apply_cmvn(coll, by_collection=by_collection)
    assert list(cmvns.keys()) == list(coll.keys())
    if by_collection:
        stats = np.zeros((2, 14))
        for f in feats:
            orc.cmvn_accumulate(f.data, stats=stats)
        allc = np.concatenate([f.data for f in cmvns.values()], axis=0)
        # (the synthetic C0 has |mean| / std ~ 400: Kaldi's float32 x*x and offset lose ~1e-3 there,
        # identically in the oracle - the bit-exact comparison below covers that column)
        assert allc[:, 1:].mean(axis=0) == pytest.approx(0, abs=1e-5)
        assert allc[:, 1:].var(axis=0) == pytest.approx(1, abs=1e-5)
    for k, f in coll.items():
        st = stats if by_collection else orc.cmvn_accumulate(f.data)
        np.testing.assert_allclose(cmvns[k].properties['cmvn']['stats'], st, rtol=1e-12)
        want = orc.cmvn_apply(f.data, cmvns[k].properties['cmvn']['stats'])
        assert np.array_equal(cmvns[k].data, want)
        if not by_collection:
            assert cmvns[k].data[:, 1:].mean(axis=0) == pytest.approx(0, abs=1e-5)
            assert cmvns[k].data[:, 1:].var(axis=0) == pytest.approx(1, abs=1e-5)
    weights = {k: None for k in coll.keys()}
    assert apply_cmvn(coll, by_collection=by_collection, weights=weights) == cmvns
    skipped = apply_cmvn(coll, skip_dims=[0, 1], by_collection=False)
    for k, f in skipped.items():
        assert np.array_equal(f.data[:, :2], coll[k].data[:, :2])
        assert f.data[:, 2:].mean(axis=0) == pytest.approx(0, abs=1e-5)


@pytest.mark.parametrize('norm_vars, center', [(s, v) for s in (True, False) for v in (True, False)])
@pytest.mark.parametrize('window', [(40, 40), (600, 100), (30, 10), (7, 50)])
def test_sliding_cmvn(gpu, audio, norm_vars, center, window):
    """bit-exact against the oracle (same double-precision incremental sums) + the reference's own
    frame-70 check (test/postprocessor/test_cmvn.py:223-263)"""
    mfcc = MfccProcessor(dither=0).process(audio)
    backup = mfcc.data.copy()
    proc = SlidingWindowCmvnPostProcessor(
        normalize_variance=norm_vars, center=center, cmn_window=window[0], min_window=window[1])
    got = proc.process(mfcc)
    want = orc.sliding_cmn(mfcc.data, center=center, cmn_window=window[0], min_window=window[1],
                           normalize_variance=norm_vars)
    assert got.shape == mfcc.shape and got.dtype == mfcc.dtype
    np.testing.assert_allclose(got.data, want, rtol=1e-6, atol=1e-6)
    assert np.array_equal(got.times, mfcc.times) and np.array_equal(backup, mfcc.data)
    if window == (40, 40):
        frame = 70
        a, b = (frame - 20, frame + 20) if center else (frame - 40, frame + 1)
        ref = mfcc.data[frame] - mfcc.data[a:b].mean(axis=0)
        if norm_vars:
            ref = ref / mfcc.data[a:b].std(axis=0)
        assert np.all(np.isclose(got.data[frame], ref, atol=1e-6))


def test_sliding_cmvn_batch(gpu, synth_waves):
    proc = MfccProcessor(dither=0)
    feats = proc._process_batch([Audio(w, 16000) for w in synth_waves])
    feats.append(Features(feats[0].data[:1].copy(), feats[0].times[:1].copy()))
    post = SlidingWindowCmvnPostProcessor(cmn_window=50, min_window=20, normalize_variance=True)
    outs = post._process_batch(feats)
    for f, o in zip(feats, outs):
        want = orc.sliding_cmn(f.data, cmn_window=50, min_window=20, normalize_variance=True)
        np.testing.assert_allclose(o.data, want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('frame_length', [0.0165, 0.02, 0.03, 0.032])
@pytest.mark.parametrize('cls', [FilterbankProcessor, MfccProcessor])
def test_fast_kernel_window_lengths(gpu, audio, wave, cls, frame_length):
    """every even window length that pads to 512 samples runs on the register-resident kernel
    (generic per-element window test instead of the 25 ms special case)"""
    proc = cls(dither=0, frame_length=frame_length)
    got = proc.process(audio)
    want = _oracle(proc, wave)
    assert got.shape == want.shape
    assert_close(got.data, want, what=f'{cls.__name__} {frame_length}')
    plan = _backend.get_plan(proc._build_options())
    plan.run([np.asarray(wave, np.int16)])
    assert plan.kernel_name(1) == 'fbank512b_kernel'   # (flat batch, snip_edges, no dither)


@pytest.mark.parametrize('cls', [FilterbankProcessor, MfccProcessor, PlpProcessor,
                                 SpectrogramProcessor])
def test_fast_kernel_centred_frames(gpu, synth_waves, cls):
    """snip_edges = False on the register-resident kernel: the first / last frames of every
    utterance are reloaded with Kaldi's reflection, interior frames take the bulk path"""
    proc = cls(dither=0, snip_edges=False)
    waves = list(synth_waves)
    feats = proc._process_batch([Audio(w, 16000) for w in waves])
    plan = _backend.get_plan(proc._build_options())
    assert plan.kernel_name(1) == 'fbank512_kernel'
    for w, f in zip(waves, feats):
        want = _oracle(proc, w)
        assert f.shape == want.shape
        assert_close(f.data, want, rtol=1e-4, what=cls.__name__)
    # an utterance shorter than one window cannot take the clamped bulk loads: it runs on the generic
    # kernel (a second, masked launch), the other utterance stays where it always runs
    short = [waves[0], np.asarray(waves[1][:300])]
    feats = proc._process_batch([Audio(w, 16000) for w in short])
    assert (plan.kernel_name(1), plan.kernel_name(2)) == ('fbank512_kernel', 'mel_features_generic_kernel')
    for w, f in zip(short, feats):
        assert_close(f.data, _oracle(proc, w), rtol=1e-4, what=f'{proc.name} short')


@pytest.mark.parametrize('snip_edges', [True, False])
@pytest.mark.parametrize('cls', [FilterbankProcessor, MfccProcessor, PlpProcessor])
def test_fast_kernel_vtln(gpu, synth_waves, cls, snip_edges):
    """utterances with VTLN warps stay on the register-resident kernel: a workgroup works on one
    utterance and stages the mel tables of that utterance's warp factor"""
    proc = cls(dither=0, snip_edges=snip_edges)
    waves = list(synth_waves) + [np.zeros(100, np.int16)] if snip_edges else list(synth_waves)
    warps = [[1.0, 0.85, 1.2, 1.07][i % 4] for i in range(len(waves))]
    feats = proc._process_batch([Audio(w, 16000) for w in waves], vtln_warp=warps)
    plan = _backend.get_plan(proc._build_options())
    assert plan.kernel_name(1) == 'fbank512_kernel'
    for w, wf, f in zip(waves, warps, feats):
        want = _oracle(proc, w, wf)
        assert f.shape == want.shape
        if want.size:
            assert_close(f.data, want, rtol=1e-4, what=f'{cls.__name__} warp {wf}')
        assert f.properties[proc.name]['vtln_warp'] == wf


@pytest.mark.parametrize('snip_edges', [True, False])
@pytest.mark.parametrize('cls, sample_rate, opts', [
    (FilterbankProcessor, 8000, dict(num_bins=40)),           # telephone speech: 200 samples -> 256
    (FilterbankProcessor, 8000, dict(num_bins=23, use_energy=True, raw_energy=False)),
    (MfccProcessor, 8000, dict()),
    (PlpProcessor, 8000, dict()),
    (MfccProcessor, 16000, dict(frame_length=0.01, frame_shift=0.005)),  # 160 samples -> 256
    (FilterbankProcessor, 16000, dict(frame_length=0.008, frame_shift=0.004, num_bins=15)),  # -> 128
    (MfccProcessor, 8000, dict(frame_length=0.016)),          # 128 samples -> 128
    (FilterbankProcessor, 8000, dict(use_energy=True, htk_compat=True)),
    (MfccProcessor, 8000, dict(htk_compat=True, use_energy=False, raw_energy=False)),
    (MfccProcessor, 8000, dict(frame_length=0.03, frame_shift=0.011, remove_dc_offset=False)),  # 240 -> 256
    (PlpProcessor, 8000, dict(use_energy=False)),
])
def test_fast_kernel_short_frames(gpu, cls, sample_rate, opts, snip_edges):
    """frames that pad to 256 / 128 samples run on the register-resident kernel as the 512-point
    transform of the zero-extended frame (mel taps on every 2nd / 4th bin)"""
    n = int(0.7 * sample_rate)
    waves = [synth.utterances(11 + i, 1, n + 37 * i, sample_rate)[0] for i in range(3)]
    proc = cls(sample_rate=sample_rate, dither=0, snip_edges=snip_edges, **opts)
    warps = [1.0, 1.0, 1.0]
    feats = proc._process_batch([Audio(w, sample_rate) for w in waves], vtln_warp=warps)
    plan = _backend.get_plan(proc._build_options())
    # frames that pad to 256 samples: two frames per 16-lane row (X_a, X_b from one complex transform)
    padded = 1 << int(np.ceil(np.log2(opts.get('frame_length', 0.025) * sample_rate - 1e-9)))
    flat = 'fbank512b_kernel' if snip_edges else 'fbank512_kernel'
    assert plan.kernel_name(1) == ('fbank256x2_kernel' if padded == 256 else flat)
    for w, f in zip(waves, feats):
        want = _oracle(proc, w)
        assert f.shape == want.shape
        assert_close(f.data, want, rtol=1e-4, what=f'{cls.__name__} {sample_rate} {opts}')
    # with VTLN warps (per-utterance tables): the 512-point form of the zero-extended frames for the
    # warped utterances; the unwarped one keeps the kernel it runs on in any other batch
    warps = [0.9, 1.0, 1.15]
    feats = proc._process_batch([Audio(w, sample_rate) for w in waves], vtln_warp=warps)
    if padded == 256:
        assert (plan.kernel_name(1), plan.kernel_name(2)) == ('fbank256x2_kernel', 'fbank512_kernel')
    else:
        assert plan.kernel_name(1) == 'fbank512_kernel'
    for w, wf, f in zip(waves, warps, feats):
        assert_close(f.data, _oracle(proc, w, wf), rtol=1e-4, what=f'{proc.name} warp {wf} {opts}')


@pytest.mark.parametrize('snip_edges', [True, False])
@pytest.mark.parametrize('cls, sample_rate, opts', [
    (FilterbankProcessor, 8000, dict()),                      # fbank256x2_kernel: two frames per row
    (MfccProcessor, 8000, dict()),
    (PlpProcessor, 8000, dict()),
    (SpectrogramProcessor, 8000, dict()),
    (MfccProcessor, 16000, dict(frame_length=0.016, frame_shift=0.005)),
    (FilterbankProcessor, 16000, dict()),                     # fbank512_kernel
    (MfccProcessor, 44100, dict()),                           # fbank2048_kernel
    (FilterbankProcessor, 32000, dict()),                     # fbank1024x2_kernel: two frames per wave
    (MfccProcessor, 22050, dict()),                           # ... an odd window length
])
def test_features_do_not_depend_on_the_batch(gpu, cls, sample_rate, opts, snip_edges):
    """an utterance's features are the same bits whether it runs alone or inside any batch (the
    reference processes utterances one by one: shennong/processor/base.py:150-180).  The two-frames-
    per-row kernel pairs frames inside an utterance for exactly this reason; utterances with odd and
    even frame counts, and a one-frame utterance, move the pairing of everything behind them."""
    shift = int(round(opts.get('frame_shift', 0.01) * sample_rate))
    length = int(round(opts.get('frame_length', 0.025) * sample_rate))
    lengths = [length + shift * k + r for k, r in [(6, 3), (0, 0), (11, 7), (4, 1), (1, 0), (9, 5)]]
    # shorter than one window: no frame with snip_edges, reflected frames (generic kernel, alone or not)
    # without
    lengths.append(length - shift // 2 - 3)
    waves = [synth.utterances(31 + i, 1, n, sample_rate)[0] for i, n in enumerate(lengths)]
    proc = cls(sample_rate=sample_rate, dither=0, snip_edges=snip_edges, **opts)
    audios = [Audio(w, sample_rate) for w in waves]
    alone = [proc._process_batch([a])[0].data for a in audios]
    assert len({a.shape[0] % 2 for a in alone[:6]}) == 2      # odd and even frame counts both present
    assert alone[6].shape[0] == (0 if snip_edges else (lengths[6] + shift // 2) // shift)
    for order in ([0, 1, 2, 3, 4, 5], [5, 3, 1, 4, 2, 0], [2, 2, 1, 0], [6, 0, 3], [4, 6], [6, 6]):
        together = proc._process_batch([audios[i] for i in order])
        for i, f in zip(order, together):
            assert f.data.shape == alone[i].shape
            assert np.array_equal(f.data, alone[i]), f'{cls.__name__} {sample_rate}: utterance {i} in {order}'


@pytest.mark.parametrize('snip_edges', [True, False])
@pytest.mark.parametrize('cls, sample_rate, opts', [
    (FilterbankProcessor, 8000, dict()),                      # fbank256x2_kernel
    (FilterbankProcessor, 16000, dict()),                     # fbank512_kernel (dither keeps the round-2 form)
    (MfccProcessor, 44100, dict()),                           # fbank2048_kernel
    (FilterbankProcessor, 32000, dict()),                     # fbank1024x2_kernel
    (FilterbankProcessor, 16000, dict(frame_length=0.019, frame_shift=0.007, use_power=False)),  # generic
])
def test_default_dither_does_not_depend_on_the_batch(gpu, cls, sample_rate, opts, snip_edges):
    """dither = 1.0 is the reference's DEFAULT (processor/base.py:122): the noise stream of a frame is keyed
    by its index inside its utterance, the utterance's length and first samples and the call count of the
    plan - not by its row in the batch -, so the default configuration too returns the same bits for an
    utterance alone and in any batch (same call count: a fresh plan per call here), and different
    utterances / different calls draw different noise"""
    shift = int(round(opts.get('frame_shift', 0.01) * sample_rate))
    length = int(round(opts.get('frame_length', 0.025) * sample_rate))
    lengths = [length + shift * k + r for k, r in [(6, 3), (0, 0), (11, 7), (4, 1), (6, 3)]]
    waves = [synth.utterances(41 + i, 1, n, sample_rate)[0] for i, n in enumerate(lengths)]
    proc = cls(sample_rate=sample_rate, snip_edges=snip_edges, **opts)
    assert proc.dither == 1.0
    audios = [Audio(w, sample_rate) for w in waves]

    def fresh(batch):
        _backend.clear_plans()  # a new plan: its first call, the same noise stream key
        return [f.data for f in proc._process_batch(batch)]
    alone = [fresh([a])[0] for a in audios]
    for order in ([0, 1, 2, 3, 4], [4, 2, 0], [3, 3, 1]):
        for i, data in zip(order, fresh([audios[i] for i in order])):
            assert np.array_equal(data, alone[i]), f'{cls.__name__} {sample_rate}: utterance {i} in {order}'
    # the noise is there (dither 0 gives other bits), utterances of equal length draw different noise, and a
    # second call on the same plan draws a new stream
    quiet = cls(sample_rate=sample_rate, snip_edges=snip_edges, dither=0, **opts)._process_batch(audios)
    assert not np.array_equal(quiet[0].data, alone[0])
    assert not np.array_equal(alone[0] - quiet[0].data, alone[4] - quiet[4].data)
    _backend.clear_plans()
    first = proc._process_batch([audios[0]])[0].data
    again = proc._process_batch([audios[0]])[0].data
    assert np.array_equal(first, alone[0]) and not np.array_equal(again, first)
    _backend.clear_plans()


def test_delta_pitch_noise_does_not_depend_on_the_batch(gpu):
    """the same for the random term of KaldiPitchPostProcessor's delta-pitch column (default stddev 0.005)"""
    rng = np.random.default_rng(5)
    feats = []
    for n in (30, 1, 57, 30):
        data = np.stack([rng.uniform(-1, 1, n), rng.uniform(60, 300, n)], axis=1).astype(np.float32)
        feats.append(Features(data, np.arange(n) * 0.01, properties={'pitch': {}}, validate=False))
    post = KaldiPitchPostProcessor()
    assert post.delta_pitch_noise_stddev > 0
    plan_opts = post._build_options()

    def fresh(batch):
        _backend.clear_plans()
        return _backend.get_plan(plan_opts).run_post([f.data for f in batch])
    alone = [fresh([f])[0] for f in feats]
    for order in ([0, 1, 2, 3], [3, 2], [1, 0, 0]):
        for i, data in zip(order, fresh([feats[i] for i in order])):
            assert np.array_equal(data, alone[i]), (i, order)
    _backend.clear_plans()


@pytest.mark.parametrize('cls, sample_rate', [(MfccProcessor, 8000), (PlpProcessor, 8000),
                                              (FilterbankProcessor, 16000), (MfccProcessor, 44100),
                                              (MfccProcessor, 32000), (FilterbankProcessor, 22050)])
def test_features_do_not_depend_on_the_warps_of_the_batch(gpu, cls, sample_rate):
    """VTLN: an utterance runs on the same kernel, to the same bits, whatever the warp factors of the
    utterances around it (found by tools/fuzz_pipeline.py, seed 8 case 7: a streamed 8 kHz corpus whose
    batches held different speakers).  Reference: one utterance at a time, shennong/processor/base.py:150-180"""
    n = int(0.4 * sample_rate)
    waves = [synth.utterances(51 + i, 1, n + 211 * i, sample_rate)[0] for i in range(5)]
    warps = [1.0, 0.9, 1.0, 1.1, 1.0]
    proc = cls(sample_rate=sample_rate, dither=0)
    audios = [Audio(w, sample_rate) for w in waves]
    alone = [proc._process_batch([a], vtln_warp=[wf])[0].data for a, wf in zip(audios, warps)]
    for order in ([0, 1, 2, 3, 4], [0, 2, 4], [3, 0, 1], [4, 3]):
        together = proc._process_batch([audios[i] for i in order], vtln_warp=[warps[i] for i in order])
        for i, f in zip(order, together):
            assert np.array_equal(f.data, alone[i]), f'{cls.__name__} {sample_rate}: utterance {i} in {order}'


@pytest.mark.parametrize('cls, sample_rate', [(MfccProcessor, 8000), (FilterbankProcessor, 16000),
                                              (PlpProcessor, 8000), (FilterbankProcessor, 44100),
                                              (FilterbankProcessor, 32000), (PlpProcessor, 22050)])
def test_every_route_in_one_batch(gpu, cls, sample_rate):
    """snip_edges = False, VTLN warps and utterances shorter than a window together: warped / unwarped /
    sub-window utterances each take their own kernel inside one call (capi.hip: split_dual, run_short),
    also when one of the groups is empty; every utterance agrees with the oracle and with itself alone"""
    win = int(round(0.025 * sample_rate))
    n = int(0.3 * sample_rate)
    waves = [synth.utterances(71 + i, 1, m, sample_rate)[0]
             for i, m in enumerate((win - 9, n, n + 57, win - 40, n + 131))]
    warps = [0.9, 1.0, 1.1, 1.0, 1.0]
    proc = cls(sample_rate=sample_rate, dither=0, snip_edges=False)
    audios = [Audio(w, sample_rate) for w in waves]
    alone = [proc._process_batch([a], vtln_warp=[wf])[0].data for a, wf in zip(audios, warps)]
    for order in ([0, 1, 2, 3, 4], [0, 3], [0, 2], [3, 1], [0, 3, 0]):
        together = proc._process_batch([audios[i] for i in order], vtln_warp=[warps[i] for i in order])
        for i, f in zip(order, together):
            assert np.array_equal(f.data, alone[i]), f'{cls.__name__} {sample_rate}: utterance {i} in {order}'
    for w, wf, f in zip(waves, warps, alone):
        assert_close(f, _oracle(proc, w, wf), rtol=1e-4, what=f'{cls.__name__} {sample_rate} warp {wf}')


@pytest.mark.parametrize('snip_edges', [True, False])
@pytest.mark.parametrize('cls, sample_rate, opts', [
    (FilterbankProcessor, 44100, dict(num_bins=40)),          # 1102 samples -> 2048 (reference test rate)
    (FilterbankProcessor, 44100, dict(num_bins=23, use_energy=True, raw_energy=False)),
    (FilterbankProcessor, 44100, dict(use_energy=True, htk_compat=True, use_power=False)),
    (FilterbankProcessor, 44100, dict(remove_dc_offset=False, preemph_coeff=0.0, window_type='hamming')),
    (MfccProcessor, 44100, dict()),
    (MfccProcessor, 44100, dict(htk_compat=True, use_energy=False, num_ceps=20, num_bins=40)),
    (MfccProcessor, 48000, dict(use_energy=False)),           # 1200 samples -> 2048
    (PlpProcessor, 44100, dict()),
    (SpectrogramProcessor, 44100, dict()),
    (SpectrogramProcessor, 32000, dict(raw_energy=False)),    # 800 samples -> 1024: two frames per transform
    (MfccProcessor, 32000, dict()),
    (FilterbankProcessor, 32000, dict(num_bins=40)),
    (FilterbankProcessor, 32000, dict(num_bins=23, use_energy=True, raw_energy=False, htk_compat=True)),
    (FilterbankProcessor, 32000, dict(use_energy=True, use_power=False, remove_dc_offset=False)),
    (PlpProcessor, 32000, dict()),
    (FilterbankProcessor, 22050, dict(num_bins=40)),           # 551 samples (odd) -> 1024
    (MfccProcessor, 22050, dict(use_energy=False, window_type='hamming', preemph_coeff=0.0)),
    (SpectrogramProcessor, 22050, dict()),
    (PlpProcessor, 22050, dict(rasta=True)),
    (FilterbankProcessor, 16000, dict(frame_length=0.064, frame_shift=0.02, num_bins=40)),  # 1024 samples
    (FilterbankProcessor, 16000, dict(frame_length=0.1, frame_shift=0.03, num_bins=64)),    # 1600 -> 2048
])
def test_fast_kernel_long_frames(gpu, cls, sample_rate, opts, snip_edges):
    """frames that pad to 2048 samples (44.1 / 48 kHz; the reference tests MFCC at 44.1 kHz,
    test/processor/test_mfcc.py:129-137) run on the register-resident 2048-point kernel, frames that pad to
    1024 samples (22.05 / 32 kHz, odd window lengths included) two at a time on the same transform; VTLN warps
    included"""
    n = int(0.35 * sample_rate)
    waves = [synth.utterances(21 + i, 1, n + 1013 * i, sample_rate)[0] for i in range(3)]
    proc = cls(sample_rate=sample_rate, dither=0, snip_edges=snip_edges, **opts)
    plan = _backend.get_plan(proc._build_options())
    linear = opts.get('use_power', True) is False
    kw = dict(vtln_warp=[1.0, 1.0, 1.0]) if cls is not SpectrogramProcessor else {}
    feats = proc._process_batch([Audio(w, sample_rate) for w in waves], **kw)
    kernel = 'fbank1024x2_kernel' if proc.frame_length * sample_rate <= 1024 else 'fbank2048_kernel'
    assert plan.kernel_name(1) == kernel
    for w, f in zip(waves, feats):
        want = _oracle(proc, w)
        assert f.shape == want.shape
        assert_close(f.data, want, rtol=1e-4,
                     what=f'{cls.__name__} {sample_rate} {opts}')
    if cls is SpectrogramProcessor or linear:
        return
    warps = [0.9, 1.0, 1.15]
    feats = proc._process_batch([Audio(w, sample_rate) for w in waves], vtln_warp=warps)
    assert plan.kernel_name(1) == kernel
    for w, wf, f in zip(waves, warps, feats):
        assert_close(f.data, _oracle(proc, w, wf), rtol=1e-4, what=f'{proc.name} warp {wf} {opts}')


@pytest.mark.parametrize('sample_rate', [32000, 22050, 8000])
def test_paired_frames_of_unequal_energy(gpu, sample_rate):
    """two frames per transform share their roundings, the louder frame sets the error floor of both: a quiet
    frame beside an onset (digital silence beside a full-scale tone, a +-1 LSB murmur beside it, a 60 dB step)
    must still match the oracle - such pairs are transformed one frame at a time (kernels_fbank1024x2.hip)"""
    rng = np.random.default_rng(5)
    n = sample_rate
    t = np.arange(n) / sample_rate
    # (a tone over a noise floor 20 dB below it: the bins of a PURE tone far from its frequency are float
    # round-off in any transform, the oracle's included, and the cepstra of such a spectrum cancel to it)
    tone = (20000 * np.sin(2 * np.pi * 997.0 * t) + rng.integers(-3000, 3001, size=n)).astype(np.int16)
    waves = []
    for kind in range(4):
        w = tone.copy()
        quiet = slice(n // 4, n // 2 + 137 * kind)
        if kind == 0:
            w[quiet] = 0
        elif kind == 1:
            w[quiet] = rng.integers(-1, 2, size=w[quiet].shape)
        elif kind == 2:
            w[quiet] = rng.integers(-30, 31, size=w[quiet].shape)
        else:
            w[:n // 3] = 0
            w[n // 3:] = rng.integers(-20000, 20000, size=n - n // 3)
        waves.append(w)
    kernel = 'fbank256x2_kernel' if sample_rate == 8000 else 'fbank1024x2_kernel'
    for cls, opts in ((FilterbankProcessor, dict(num_bins=40 if sample_rate > 8000 else 23)), (MfccProcessor, dict()),
                      (SpectrogramProcessor, dict())):
        proc = cls(sample_rate=sample_rate, dither=0, **opts)
        feats = proc._process_batch([Audio(w, sample_rate) for w in waves])
        plan = _backend.get_plan(proc._build_options())
        assert plan.kernel_name(1) == kernel
        for k, (w, f) in enumerate(zip(waves, feats)):
            assert_close(f.data, _oracle(proc, w), rtol=1e-4, what=f'{proc.name} {sample_rate} Hz, signal {k}')


def test_long_frames_single_utterance_shorter_than_a_window(gpu):
    """snip_edges = False with an utterance shorter than one window: the generic kernel takes over"""
    wave = synth.utterances(3, 1, 900, 44100)[0]
    proc = FilterbankProcessor(sample_rate=44100, dither=0, snip_edges=False)
    got = proc.process(Audio(wave, 44100))
    assert_close(got.data, _oracle(proc, wave), rtol=1e-4, what=f'{proc.name} short utterance')
    plan = _backend.get_plan(proc._build_options())
    assert plan.kernel_name(1) == 'mel_features_generic_kernel'


def test_tables_too_large_for_lds_fall_back(gpu):
    """banks too wide for the LDS of the register-resident 512-point kernel run on the generic kernel
    instead of failing at launch (found by tests/tools/fuzz_parity.py: seed 12 case 69, seed 13 case 168)"""
    waves = [synth.utterances(90 + i, 1, n, 8000)[0] for i, n in enumerate((11120, 6622, 7353, 4236))]
    proc = FilterbankProcessor(sample_rate=8000, frame_length=0.03, num_bins=64, low_freq=0, dither=0,
                               window_type='hanning')
    warps = [0.85, 1.0, 1.1, 0.93]
    feats = proc._process_batch([Audio(w, 8000) for w in waves], vtln_warp=warps)
    for w, wf, f in zip(waves, warps, feats):
        assert_close(f.data, _oracle(proc, w, wf), rtol=1e-4, what=f'{proc.name} wide banks, warp {wf}')
    wave = synth.utterances(95, 1, 30000, 44100)[0]
    proc = MfccProcessor(sample_rate=44100, frame_length=0.008, frame_shift=0.0125, num_bins=59, num_ceps=3,
                         low_freq=100, high_freq=21750, htk_compat=True, use_energy=False, dither=0,
                         window_type='hanning')
    assert_close(proc.process(Audio(wave, 44100)).data, _oracle(proc, wave), rtol=1e-4, what=f'{proc.name} 59 bins at 44.1 kHz')


def test_short_frames_spectrogram_and_energy(gpu):
    """the spectrogram of a 256-sample frame needs its own 129 bins: the two-frames-per-transform kernel
    (round 6; the generic kernel until then, and still for 128-sample frames); the frame energy has no
    spectrum at all: fast kernel"""
    wave = synth.utterances(5, 1, 6000, 8000)[0]
    for opts in (dict(), dict(raw_energy=False), dict(snip_edges=False)):
        proc = SpectrogramProcessor(sample_rate=8000, dither=0, **opts)
        got = proc.process(Audio(wave, 8000))
        assert got.shape[1] == 129
        assert_close(got.data, _oracle(proc, wave), rtol=1e-4, family='spectrogram')
        plan = _backend.get_plan(proc._build_options())
        plan.run([wave])
        assert plan.kernel_name(1) == 'fbank256x2_kernel'
    proc = SpectrogramProcessor(sample_rate=8000, frame_length=0.016, frame_shift=0.008, dither=0)   # 128 samples
    got = proc.process(Audio(wave, 8000))
    assert got.shape[1] == 65
    assert_close(got.data, _oracle(proc, wave), rtol=1e-4, family='spectrogram')
    plan = _backend.get_plan(proc._build_options())
    plan.run([wave])
    assert plan.kernel_name(1) == 'mel_features_generic_kernel'
    eproc = EnergyProcessor(sample_rate=8000, dither=0)
    got = eproc.process(Audio(wave, 8000))
    np.testing.assert_allclose(got.data, _oracle(eproc, wave), rtol=1e-6)
    plan = _backend.get_plan(eproc._build_options())
    plan.run([wave])
    assert plan.kernel_name(1) == 'fbank512_kernel'


def test_more_than_2g_output_elements(gpu):
    """maximum sizes: a batch whose output has more than 2^31 elements (8.4 M spectrogram frames of
    257 bins, 8.6 GB) and whose waveform has more than 2^31 bytes - every row index is 64-bit.  Device
    resident; rows on both sides of the 2^31-element boundary and the last rows are compared with
    the same utterances processed alone."""
    import ctypes as C
    block = synth.utterances(77, 100, 48000)
    n_utts, nsamp, nfr = 28200, 48000, 298
    assert n_utts * nfr * 257 > 2**31 and n_utts * nsamp * 2 > 2**31
    proc = SpectrogramProcessor(dither=0)
    plan = _backend.get_plan(proc._build_options())
    soff = np.arange(n_utts + 1, dtype=np.int64) * nsamp
    foff = np.arange(n_utts + 1, dtype=np.int64) * nfr
    d_wave = _backend.DeviceBuffer(n_utts * nsamp * 2)
    flat = np.ascontiguousarray(block.reshape(-1))
    for k in range(n_utts // 100):  # utterance u holds block[u % 100]
        _backend.check(_backend.lib().snf_memcpy_h2d(
            C.c_void_p(d_wave.ptr + k * flat.nbytes), flat.ctypes.data_as(C.c_void_p), flat.nbytes))
    d_out = _backend.DeviceBuffer(n_utts * nfr * 257 * 4)
    plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
    assert plan.kernel_name(1) == 'fbank512_kernel'
    boundary = 2**31 // (nfr * 257)  # the utterance whose rows straddle element 2^31
    for u in (0, boundary - 1, boundary, boundary + 1, n_utts - 1):
        got = np.empty((nfr, 257), dtype=np.float32)
        _backend.check(_backend.lib().snf_memcpy_d2h(
            got.ctypes.data_as(C.c_void_p), C.c_void_p(d_out.ptr + u * nfr * 257 * 4), got.nbytes))
        alone = plan.run([block[u % 100]])[0]
        assert np.array_equal(got, alone), u
    d_out.free()
    # the same batch through a mel kind with a small row (delta on top): > 2^31 BYTES of samples
    mproc = MfccProcessor(dither=0)
    mplan = _backend.get_plan(mproc._build_options())
    d_mfcc = _backend.DeviceBuffer(n_utts * nfr * 13 * 4)
    mplan.run_device(d_wave.ptr, soff, foff, d_mfcc.ptr)
    dplan = _backend.get_plan(DeltaPostProcessor()._build_options())
    d_delta = _backend.DeviceBuffer(n_utts * nfr * 39 * 4)
    dplan.run_post_device(d_mfcc.ptr, 13, foff, d_delta.ptr)
    for u in (0, n_utts // 2, n_utts - 1):
        got = np.empty((nfr, 39), dtype=np.float32)
        _backend.check(_backend.lib().snf_memcpy_d2h(
            got.ctypes.data_as(C.c_void_p), C.c_void_p(d_delta.ptr + u * nfr * 39 * 4), got.nbytes))
        alone = dplan.run_post([mplan.run([block[u % 100]])[0]])[0]
        assert np.array_equal(got, alone), u
    for buf in (d_mfcc, d_delta):
        buf.free()
    # fbank-40 (fbank512b_kernel: the rows of a set leave through a buffer descriptor rebuilt per set): more
    # than 2^32 BYTES of output need 3.2 x the utterances - the wave buffer is reused as utterance u % 28200
    fproc = FilterbankProcessor(num_bins=40, dither=0)
    fplan = _backend.get_plan(fproc._build_options())
    d_out = _backend.DeviceBuffer(n_utts * nfr * 40 * 4)
    fplan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
    assert fplan.kernel_name(1) == 'fbank512b_kernel' and n_utts * nfr * 40 * 4 > 2**30
    for u in (0, n_utts // 2, n_utts - 1):
        got = np.empty((nfr, 40), dtype=np.float32)
        _backend.check(_backend.lib().snf_memcpy_d2h(
            got.ctypes.data_as(C.c_void_p), C.c_void_p(d_out.ptr + u * nfr * 160), got.nbytes))
        assert np.array_equal(got, fplan.run([block[u % 100]])[0]), u
    d_out.free()
    d_wave.free()


def test_more_than_4g_output_bytes_flat_kernel(gpu):
    """fbank-40 rows past byte 2^32 of the output (91 000 utterances, 4.3 GB of rows, 8.7 GB of samples): the
    flat kernel addresses every set through its own 64-bit base"""
    import ctypes as C
    block = synth.utterances(78, 100, 48000)
    n_utts, nsamp, nfr = 91000, 48000, 298
    assert n_utts * nfr * 160 > 2**32
    plan = _backend.get_plan(FilterbankProcessor(num_bins=40, dither=0)._build_options())
    soff = np.arange(n_utts + 1, dtype=np.int64) * nsamp
    foff = np.arange(n_utts + 1, dtype=np.int64) * nfr
    d_wave = _backend.DeviceBuffer(n_utts * nsamp * 2)
    flat = np.ascontiguousarray(block.reshape(-1))
    for k in range(n_utts // 100):
        _backend.check(_backend.lib().snf_memcpy_h2d(
            C.c_void_p(d_wave.ptr + k * flat.nbytes), flat.ctypes.data_as(C.c_void_p), flat.nbytes))
    d_out = _backend.DeviceBuffer(n_utts * nfr * 160)
    plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
    assert plan.kernel_name(1) == 'fbank512b_kernel'
    boundary = 2**32 // (nfr * 160)
    for u in (0, boundary - 1, boundary, boundary + 1, n_utts - 1):
        got = np.empty((nfr, 40), dtype=np.float32)
        _backend.check(_backend.lib().snf_memcpy_d2h(
            got.ctypes.data_as(C.c_void_p), C.c_void_p(d_out.ptr + u * nfr * 160), got.nbytes))
        assert np.array_equal(got, plan.run([block[u % 100]])[0]), u
    d_wave.free()
    d_out.free()


def test_long_utterance(gpu):
    """one 10-minute utterance (9.6 M samples, 59 998 frames): fbank / MFCC / pitch against the oracle
    (the pitch tracker walks the whole utterance with one wavefront)"""
    wave = synth.utterances(5, 1, 9600000)[0]
    for proc in (FilterbankProcessor(num_bins=40, dither=0), MfccProcessor(dither=0)):
        got = proc.process(Audio(wave, 16000))
        want = _oracle(proc, wave)
        assert got.shape == want.shape == (59998, want.shape[1])
        assert_close(got.data, want, what=proc.name)
    pproc = KaldiPitchProcessor()
    got = pproc.process(Audio(wave, 16000))
    _pitch_close(got.data, orc.pitch(pproc._options, wave))


# ---- reference test/processor/test_stability.py and test_parallel.py ------------------------------------
@pytest.mark.parametrize('same', [True, False])
@pytest.mark.parametrize('make', [
    lambda: EnergyProcessor(), lambda: FilterbankProcessor(), lambda: MfccProcessor(),
    lambda: PlpProcessor(), lambda: PlpProcessor(rasta=True), lambda: KaldiPitchProcessor(),
    lambda: SpectrogramProcessor()], ids=['energy', 'fbank', 'mfcc', 'plp', 'rasta-plp', 'pitch', 'spec'])
def test_stable(gpu, audio, make, same):
    """the features are exactly the same across computations (dither off), with one processor
    instance or two"""
    p1 = make()
    p2 = p1 if same else make()
    for p in (p1, p2):
        if hasattr(p, 'dither'):
            p.dither = 0
    assert p1.process(audio) == p2.process(audio)


def test_process_all_like_reference(gpu, wav_file, capsys):
    import multiprocessing
    from shennong_amd import Utterances
    utterances = Utterances([('u1', wav_file, 0, 0.2), ('u2', wav_file, 0, 0.2), ('u3', wav_file, 0, 0.2)])
    features = MfccProcessor().process_all(utterances)
    values = list(features.values())
    assert utterances.by_name().keys() == features.keys() and len(values) == 3
    assert all(values[0].is_close(v, atol=10) for v in values[1:])  # (dither is on: close, not equal)
    features = MfccProcessor().process_all(utterances, vtln_warp={f'u{n + 1}': 1.0 for n in range(3)})
    assert utterances.by_name().keys() == features.keys()
    with pytest.raises(TypeError):
        MfccProcessor().process_all(utterances, bad_name={f'u{n + 1}': 1.0 for n in range(3)})
    with pytest.raises(ValueError, match='is not a dict'):
        MfccProcessor().process_all(utterances, vtln_warp=1.0)
    with pytest.raises(ValueError, match='have different names'):
        MfccProcessor().process_all(utterances, vtln_warp={f'{n}': 1.0 for n in range(2)})
    proc = MfccProcessor()
    proc.set_logger('debug')
    with pytest.raises(ValueError, match='must be strictly positive'):
        proc.process_all(utterances, njobs=0)
    for njobs in (1, 2, 1000):
        capsys.readouterr()
        features = proc.process_all(utterances, njobs=njobs)
        if njobs > multiprocessing.cpu_count():
            assert 'CPU cores but reducing to' in capsys.readouterr().err
        assert utterances.by_name().keys() == features.keys()


def test_large_batches_and_pinned_utterances(gpu, wav_file):
    """`process_all` on a batch large enough for the pipelined path (pieces through plan clones on the copy
    threads, round 5) and on ``Utterances.pin()`` (upload straight from one page-locked block): the same bits
    as one `process` call per utterance - ragged lengths, odd lengths (pieces start on 16-byte boundaries only
    where an utterance does), an utterance without a frame, VTLN warps - and the reference's error for a
    mismatched sample rate"""
    from shennong_amd import Utterances
    rng = np.random.default_rng(11)
    lengths = [int(x) for x in rng.integers(16000, 64000, 900)]
    lengths[5], lengths[6], lengths[400] = 16001, 399, 31999       # odd, shorter than a window, odd
    waves = [synth.utterances(i, 1, n)[0] for i, n in enumerate(lengths)]
    assert sum(lengths) * 2 > _backend._LARGE_BATCH_BYTES
    utts = Utterances([(f'u{i:04d}', Audio(w, 16000, validate=False)) for i, w in enumerate(waves)])
    warps = {u.name: (1.0, 0.9, 1.15)[i % 3] for i, u in enumerate(utts)}
    pinned = utts.pin()
    assert [u.name for u in pinned] == [u.name for u in utts]
    for proc in (FilterbankProcessor(num_bins=40, dither=0), MfccProcessor(dither=0)):
        for kwargs in ({}, {'vtln_warp': warps}):
            plain = proc.process_all(utts, **kwargs)
            fast = proc.process_all(pinned, **kwargs)
            assert list(plain.keys()) == list(fast.keys()) == [u.name for u in utts]
            for i in (0, 5, 6, 7, 399, 400, 401, 899):
                name = f'u{i:04d}'
                one = proc.process(Audio(waves[i], 16000), **({'vtln_warp': warps[name]} if kwargs else {}))
                assert np.array_equal(plain[name].data, one.data), (proc.name, name)
                assert plain[name] == one
            for name in plain:
                assert np.array_equal(plain[name].data, fast[name].data), (proc.name, name)
                assert plain[name].properties == fast[name].properties
    with pytest.raises(ValueError, match='mismatch in sample rates'):
        FilterbankProcessor(sample_rate=8000).process_all(pinned)
    # files and segments are loaded once; the audio of a pinned utterance is what load_audio gave
    index = [('a', wav_file, 's1', 0.1, 0.6), ('b', wav_file, 's2', 0.0, 1.4), ('c', wav_file, 's1', 0.3, 0.9)]
    segs = Utterances(index).pin()
    assert segs.has_speakers() and [u.speaker for u in segs] == [u.speaker for u in Utterances(index)]
    proc = MfccProcessor(dither=0)
    got = proc.process_all(segs)
    want = proc.process_all(Utterances(index))
    assert all(got[k] == want[k] for k in 'abc')


def test_pitch_flat_search(gpu, monkeypatch):
    """the lane-per-candidate Viterbi search (csrc/kernels_pitch.hip 4c, the default for large batches since round
    5) returns the bits of the lane-per-state search (SNF_PITCH_FLAT=0) and of the oracle: ragged utterances, some
    short enough for RecomputeBacktraces, digital silence (every cost ties) and noise"""
    monkeypatch.setenv('SNF_PITCH_TEAM', '1')      # one wave per utterance whatever the batch size
    rng = np.random.default_rng(21)
    waves = [synth.utterances(300 + i, 1, int(n))[0] for i, n in enumerate(rng.integers(12000, 90000, 60))]
    waves[3] = np.zeros(40000, dtype=np.int16)
    waves[4] = (rng.standard_normal(50000) * 200).astype(np.int16)
    waves[5] = np.concatenate([waves[6][:9000] // 50, waves[6][:30000]])   # a quiet start: the ballast changes
    audios = [Audio(w, 16000, validate=False) for w in waves]
    proc = KaldiPitchProcessor()
    monkeypatch.setenv('SNF_PITCH_FLAT', '0')
    shipped = [f.data.copy() for f in proc._process_batch(audios)]
    monkeypatch.setenv('SNF_PITCH_FLAT', '1')
    flat = [f.data.copy() for f in proc._process_batch(audios)]
    for i, (a, b) in enumerate(zip(shipped, flat)):
        assert np.array_equal(a, b), i
    for i in (0, 3, 4, 5, 17, 59):
        np.testing.assert_array_equal(flat[i], orc.pitch(proc._options, waves[i]), err_msg=str(i))


def test_large_batches_from_several_threads(gpu):
    """the large-batch path runs its pieces through clones of the plan on two shared streams: concurrent calls
    with the same options (they take turns on the clones) and with different options (side by side) return what
    a sequential call returns"""
    from concurrent.futures import ThreadPoolExecutor
    from shennong_amd import Utterances
    rng = np.random.default_rng(5)
    corpora = []
    for c in range(3):
        lengths = [int(x) for x in rng.integers(20000, 70000, 500)]
        waves = [synth.utterances(1000 * c + i, 1, n)[0] for i, n in enumerate(lengths)]
        assert sum(lengths) * 2 > _backend._LARGE_BATCH_BYTES
        corpora.append(Utterances([(f'c{c}u{i:03d}', Audio(w, 16000, validate=False)) for i, w in enumerate(waves)]))
    procs = [FilterbankProcessor(num_bins=40, dither=0), MfccProcessor(dither=0)]
    want = {(p, c): procs[p].process_all(corpora[c]) for p in range(2) for c in range(3)}
    jobs = [(p, c) for c in range(3) for p in range(2)] * 3

    def work(job):
        p, c = job
        return procs[p].process_all(corpora[c] if (p + c) % 2 else corpora[c].pin())
    with ThreadPoolExecutor(6) as pool:
        got = list(pool.map(work, jobs))
    for job, coll in zip(jobs, got):
        ref = want[job]
        assert list(coll.keys()) == list(ref.keys())
        for name in list(ref.keys())[::7]:
            assert np.array_equal(coll[name].data, ref[name].data), (job, name)


def test_threaded_callers(gpu, synth_waves):
    """the reference's callers are joblib THREADS (processor/base.py:104-107, pipeline.py:545-565):
    concurrent `process` / `process_all` calls on one shared plan and on different plans return what
    the sequential calls return (a host-pointer call owns the plan's staging scratch for its whole
    duration; staging buffers come from a locked pool)"""
    from concurrent.futures import ThreadPoolExecutor
    audios = [Audio(w, 16000) for w in synth_waves] * 4
    procs = [FilterbankProcessor(num_bins=40, dither=0), MfccProcessor(dither=0),
             FilterbankProcessor(num_bins=40, dither=0)]  # the first and the last share a plan
    want = [[p.process(a).data for a in audios] for p in procs]

    def work(job):
        k, i = job
        return procs[k].process(audios[i]).data

    jobs = [(k, i) for i in range(len(audios)) for k in range(len(procs))]
    with ThreadPoolExecutor(8) as pool:
        got = list(pool.map(work, jobs))
    for (k, i), g in zip(jobs, got):
        assert np.array_equal(g, want[k][i]), (k, i)
    # whole batches from several threads (pinned staging buffers are pooled)
    big = [Audio(w, 16000) for w in synth.utterances(3, 40, 48000)]
    ref = [f.data for f in procs[0]._process_batch(big)]
    with ThreadPoolExecutor(4) as pool:
        outs = list(pool.map(lambda _: [f.data for f in procs[0]._process_batch(big)], range(8)))
    for out in outs:
        assert all(np.array_equal(a, b) for a, b in zip(out, ref))


def test_energy_odd_window(gpu):
    """25 ms at 22.05 kHz without rounding to a power of two is a 551-sample window: fine for the
    frame energy (no FFT), a Kaldi RealFft error for every spectral kind (found by tests/tools/fuzz_parity.py)"""
    wave = synth.utterances(9, 1, 22050, 22050)[0]
    for raw in (True, False):
        proc = EnergyProcessor(sample_rate=22050, round_to_power_of_two=False, raw_energy=raw, dither=0)
        got = proc.process(Audio(wave, 22050))
        np.testing.assert_allclose(got.data, _oracle(proc, wave), rtol=1e-6)
    with pytest.raises(RuntimeError):
        MfccProcessor(sample_rate=22050, round_to_power_of_two=False, dither=0).process(Audio(wave, 22050))


def test_vtln_option_errors_need_a_frame(gpu, wave):
    """Kaldi builds the mel banks of a warp factor when the first frame asks for them: vtln_low <=
    low_freq is an error for a warped utterance WITH frames, and goes unnoticed for one without
    (found by tests/tools/fuzz_parity.py)"""
    proc = FilterbankProcessor(num_bins=30, low_freq=100, dither=0)  # vtln_low = 100: bad once warped
    tiny = np.zeros(2, np.int16)
    feats = proc._process_batch([Audio(wave, 16000), Audio(tiny, 16000)], vtln_warp=[1.0, 0.85])
    assert feats[0].shape == (140, 30) and feats[1].shape == (0, 0)
    assert_close(feats[0].data, _oracle(proc, wave), family='fbank')
    assert orc.compute(proc._build_options(), tiny, 0.85).size == 0
    with pytest.raises(RuntimeError, match='vtln-low'):
        proc._process_batch([Audio(wave, 16000), Audio(tiny, 16000)], vtln_warp=[0.85, 1.0])
    with pytest.raises(RuntimeError):
        orc.compute(proc._build_options(), wave, 0.85)


# ---- closed-form known answers (tests/known_answers.py), the same cases the oracle is pinned with ------
import known_answers  # noqa: E402


@pytest.mark.parametrize('case', known_answers.CASES, ids=[c[0] for c in known_answers.CASES])
def test_known_answer(gpu, case):
    _, make, wave, check = case
    check(make().process(Audio(wave, 16000)).data)


def test_known_answer_parseval(gpu, audio):
    fbank = FilterbankProcessor(num_bins=23, dither=0).process(audio)
    mfcc = MfccProcessor(num_ceps=23, cepstral_lifter=0, use_energy=False, dither=0).process(audio)
    known_answers.parseval_check(fbank.data, mfcc.data)


def test_known_answer_delta_ramp(gpu):
    from shennong_amd import Features

    def deltas(x, order, window):
        feats = Features(x, np.arange(x.shape[0], dtype=np.float64))
        return DeltaPostProcessor(order=order, window=window).process(feats).data
    known_answers.ramp_delta_check(deltas)


def test_buffers_from_a_fresh_thread(gpu):
    """hipSetDevice is per thread: allocations and copies made by a new thread bind it to the selected
    GPU first (set_device only set a Python global)"""
    import threading
    gpu.set_device(gpu.get_device())
    result = {}

    def work():
        data = np.arange(1 << 16, dtype=np.float32)
        buf = gpu.DeviceBuffer(data.nbytes)
        buf.upload(data)
        back = np.empty_like(data)
        buf.download(back)
        result['ok'] = bool(np.array_equal(back, data)) and buf.device == gpu.get_device()
        buf.free()
    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert result.get('ok')


# ---- MFCC + delta + delta-delta through one plan (BASELINE config 3) ---------------------------------
@pytest.mark.parametrize('opts', [dict(), dict(use_energy=False, htk_compat=True),
                                  dict(raw_energy=False), dict(num_ceps=10, num_bins=30)])
def test_mfcc_with_deltas_equals_the_chain(gpu, audio, opts):
    proc = MfccProcessor(dither=0, **opts)
    fused = proc.process_with_deltas(audio)
    chained = DeltaPostProcessor().process(proc.process(audio))
    assert fused.shape == chained.shape == (140, 3 * proc.num_ceps)
    np.testing.assert_array_equal(fused.data, chained.data)
    assert fused.properties == chained.properties and np.array_equal(fused.times, chained.times)
    want = orc.deltas(orc.compute(proc._build_options(), audio.data), 2, 2)
    assert_close(fused.data, want, family='mfcc')


def test_mfcc_with_deltas_batch(gpu):
    """ragged batch: utterances of one frame, of several workgroups (> 328 frames: the +-4 frame halo
    between workgroups is recomputed), one too short for a frame, and per-utterance VTLN warps"""
    waves = [synth.utterances(50 + i, 1, n)[0] for i, n in enumerate((400, 560, 16000, 90000, 100, 48000, 131072))]
    warps = [1.0, 0.9, 1.0, 1.1, 1.0, 1.0, 0.95]
    proc = MfccProcessor(dither=0)
    audios = [Audio(w, 16000) for w in waves]
    fused = proc._process_batch_with_deltas(audios, vtln_warp=warps)
    plain = proc._process_batch(audios, vtln_warp=warps)
    for f, m in zip(fused, plain):
        if m.shape[0] == 0:
            assert f.shape[0] == 0
            continue
        np.testing.assert_array_equal(f.data, DeltaPostProcessor().process(m).data)
    # without warps every utterance is scheduled per workgroup as well
    for f, m in zip(proc._process_batch_with_deltas(audios), proc._process_batch(audios)):
        if m.shape[0]:
            np.testing.assert_array_equal(f.data, DeltaPostProcessor().process(m).data)


def test_mfcc_with_deltas_other_rates(gpu):
    """two launches (the shipped form of append_deltas) cover every MFCC configuration, e.g. frames that pad
    to 2048 samples"""
    proc = MfccProcessor(dither=0, sample_rate=44100)
    audio44 = Audio(synth.utterances(7, 1, 44100)[0], 44100)
    np.testing.assert_array_equal(proc.process_with_deltas(audio44).data,
                                  DeltaPostProcessor().process(proc.process(audio44)).data)


def test_mfcc_with_deltas_fused_form(gpu, audio, monkeypatch):
    """SNF_FUSED_DELTA=1 selects the one-launch form of round 2 (slower than the two launches, kept for the
    A/B in bench.py): same rows, and its refusals"""
    from shennong_amd import _backend
    monkeypatch.setenv('SNF_FUSED_DELTA', '1')
    _backend.clear_plans()
    try:
        proc = MfccProcessor(dither=0)
        np.testing.assert_array_equal(proc.process_with_deltas(audio).data,
                                      DeltaPostProcessor().process(proc.process(audio)).data)
        proc44 = MfccProcessor(dither=0, sample_rate=44100)   # frames pad to 2048 samples
        with pytest.raises(ValueError, match='append_deltas needs frames that pad to 512'):
            proc44.process_with_deltas(Audio(np.zeros(44100, np.int16), 44100))
    finally:
        _backend.clear_plans()

@pytest.mark.parametrize('opts', [dict(), dict(use_energy=False), dict(htk_compat=True, use_energy=False),
                                  dict(num_bins=40, num_ceps=13), dict(frame_length=0.02, num_ceps=16)])
def test_mfcc_dct_mfma(gpu, audio, wave, monkeypatch, opts):
    """SNF_DCT_MFMA=1: the DCT-II + lifter of MFCC as a second v_mfma_f32_4x4x1 chain behind the mel chain
    (the north_star's "DCT-II cepstral lifter as MFMA tiles"; the vector-pipe form ships because it is
    faster, bench.py extra.mfcc13_dct_mfma) against the oracle on test.wav and on 1 000 synthetic
    utterances, through a private plan.  Reference: shennong/processor/mfcc.py:84-86"""
    monkeypatch.setenv('SNF_DCT_MFMA', '1')
    _backend.clear_plans()
    try:
        proc = MfccProcessor(dither=0, **opts)
        a = audio
        got = proc.process(a)
        assert_close(got.data, _oracle(proc, a.data), what='mfcc dct mfma %s' % opts, family='mfcc')
        # (the chain form lives on fbank512_kernel only: fbank512b_kernel refuses plans with dct_mfma set)
        plan = _backend.get_plan(proc._build_options())
        assert plan.kernel_name(1) == 'fbank512_kernel', plan.kernel_name(1)
        sr = 16000
        waves = synth.ragged_utterances(4242, 1000, min_s=0.2, max_s=0.6, sample_rate=sr)
        batch = proc._process_batch([Audio(w, sr, validate=False) for w in waves])
        for i in range(0, 1000, 37):
            assert_close(batch[i].data, _oracle(proc, waves[i]), what='mfcc dct mfma batch', family='mfcc')
    finally:
        _backend.clear_plans()
    monkeypatch.delenv('SNF_DCT_MFMA')
    shipped = MfccProcessor(dither=0, **opts).process(a)
    assert_close(got.data, shipped.data, what='mfcc dct mfma vs vector form', family='mfcc')
    _backend.clear_plans()


@pytest.mark.parametrize('cls', [FilterbankProcessor, MfccProcessor, PlpProcessor])
def test_dither_is_the_same_stream_on_both_512_point_kernels(gpu, monkeypatch, cls):
    """fbank512b_kernel with dither (round 4: keys from a per-call table) draws exactly the noise
    fbank512_kernel draws (keys computed in the kernel): same call id -> the same bits, so which of the two
    kernels an utterance lands on (flat batch / batch with VTLN warps) cannot be seen in its features.
    Reference default: dither = 1.0, shennong/processor/base.py:122"""
    waves = synth.ragged_utterances(4711, 37, min_s=0.1, max_s=0.7)
    soff = np.zeros(len(waves) + 1, dtype=np.int64)
    np.cumsum([w.shape[0] for w in waves], out=soff[1:])
    opts = cls(dither=1.0)._build_options()

    def run(call):
        plan = _backend.Plan(opts)
        foff = np.zeros(len(waves) + 1, dtype=np.int64)
        np.cumsum([plan.num_frames(w.shape[0]) for w in waves], out=foff[1:])
        d_wave = _backend.upload_rows(waves, np.int16)
        d_out = _backend.DeviceBuffer(int(foff[-1]) * plan.ndims * 4)
        plan.run_device(d_wave.ptr, soff, foff, d_out.ptr, noise_call=call)
        out = np.empty((int(foff[-1]), plan.ndims), dtype=np.float32)
        d_out.download(out)
        return out, plan.kernel_name(1)
    new, kernel_new = run(7)
    again, _ = run(7)
    other, _ = run(8)
    monkeypatch.setenv('SNF_FBANK512_OLD', '1')
    old, kernel_old = run(7)
    assert (kernel_new, kernel_old) == ('fbank512b_kernel', 'fbank512_kernel')
    assert np.array_equal(new, old) and np.array_equal(new, again)
    assert not np.array_equal(new, other) and np.isfinite(new).all()


@pytest.mark.parametrize('cls, sample_rate', [(FilterbankProcessor, 16000), (MfccProcessor, 8000),
                                              (FilterbankProcessor, 44100)])
def test_dither_differs_between_utterances_that_start_alike(gpu, cls, sample_rate):
    """the dither stream of an utterance used to be keyed by its length and its first two samples
    only, so equal-length utterances that begin with digital silence drew the same noise - bit-identical
    leading frames across a corpus of fixed-length segments.  The key now holds a hash of 64 samples spread
    over the waveform: same length, same first samples, different content -> different dither from frame 0;
    identical utterances still draw identical noise inside one call (features are a function of the utterance)"""
    n = sample_rate
    lead = sample_rate // 4                      # a quarter of a second of digital silence
    a = np.zeros(n, dtype=np.int16)
    b = np.zeros(n, dtype=np.int16)
    a[lead:] = synth.utterances(1, 1, n - lead, sample_rate)[0]
    b[lead:] = synth.utterances(2, 1, n - lead, sample_rate)[0]
    proc = cls(sample_rate=sample_rate, num_bins=40)    # dither = 1.0, the reference's default
    fa, fb, fa2 = proc._process_batch([Audio(a, sample_rate), Audio(b, sample_rate), Audio(a.copy(), sample_rate)])
    silent = int(0.2 * fa.nframes * lead / (n // 4))    # frames that lie wholly inside the silence
    assert silent >= 4
    assert not np.array_equal(fa.data[:silent], fb.data[:silent])
    # on silence the features are pure dither: around the common spectral shape of the noise (the column
    # means) the two utterances' draws must be uncorrelated, not copies
    da = fa.data[:silent] - fa.data[:silent].mean(axis=0)
    db = fb.data[:silent] - fb.data[:silent].mean(axis=0)
    assert abs(np.corrcoef(da.ravel(), db.ravel())[0, 1]) < 0.35
    assert np.array_equal(fa.data, fa2.data)


def test_calls_enqueued_on_a_callers_stream(gpu, synth_waves):
    """the *_device entry points are asynchronous on a caller's stream (include/shennong_amd.h): several calls
    enqueued without waiting for each, a pair of snf_event_* marks around every one (what bench.py times), give
    the bits of the synchronous call and positive device times"""
    import ctypes as C
    L = _backend.lib()
    waves = [w for w in synth_waves][:8]
    flat = np.ascontiguousarray(np.concatenate(waves))
    plan = _backend.get_plan(FilterbankProcessor(num_bins=40, dither=0)._build_options())
    frames = [plan.num_frames(len(w)) for w in waves]
    soff = np.concatenate([[0], np.cumsum([len(w) for w in waves])]).astype(np.int64)
    foff = np.concatenate([[0], np.cumsum(frames)]).astype(np.int64)
    d_wave = _backend.DeviceBuffer(flat.nbytes)
    d_wave.upload(flat)
    outs = [_backend.DeviceBuffer(int(foff[-1]) * 40 * 4) for _ in range(3)]
    plan.run_device(d_wave.ptr, soff, foff, outs[0].ptr)          # the plan's own stream: waited for
    want = np.empty((int(foff[-1]), 40), np.float32)
    outs[0].download(want)
    stream = C.c_void_p()
    _backend.check(L.snf_stream_create(C.byref(stream)))
    marks = []
    for dst in outs[1:]:
        a, b = C.c_void_p(), C.c_void_p()
        _backend.check(L.snf_event_create(C.byref(a)))
        _backend.check(L.snf_event_create(C.byref(b)))
        _backend.check(L.snf_event_record(a, stream))
        plan.run_device(d_wave.ptr, soff, foff, dst.ptr, stream=stream.value)
        _backend.check(L.snf_event_record(b, stream))
        marks.append((a, b))
    _backend.check(L.snf_stream_synchronize(stream))
    for (a, b), dst in zip(marks, outs[1:]):
        ms = C.c_float(-1.0)
        _backend.check(L.snf_event_elapsed_ms(a, b, C.byref(ms)))
        assert 0.0 < ms.value < 1000.0
        got = np.empty_like(want)
        dst.download(got)
        np.testing.assert_array_equal(got, want)
        _backend.check(L.snf_event_destroy(a))
        _backend.check(L.snf_event_destroy(b))
    assert L.snf_event_record(None, stream) != 0      # a null event is an error, not a crash
    _backend.check(L.snf_stream_destroy(stream))
    for buf in outs + [d_wave]:
        buf.free()


def test_plp_banks_are_built_when_a_frame_asks(gpu):
    """The reference's PLP builds the mel banks of a warp factor when the first frame asks for them
    (shennong/processor/plp.py:482-494, :559) - unlike Kaldi's Fbank / Mfcc computers, whose constructors build the
    unwarped banks.  8 ms frames at 8 kHz with 23 bins leave an unwarped bin empty: a warped utterance runs, an
    unwarped one is an error, an utterance without frames is empty and sees no error - here as in the oracle."""
    opts = dict(sample_rate=8000, frame_length=0.008, frame_shift=0.02, num_bins=23, num_ceps=7, dither=0,
                window_type='rectangular', snip_edges=False, low_freq=0, high_freq=-200, vtln_low=100, vtln_high=-500)
    proc = PlpProcessor(**opts)
    wave = synth.utterances(7, 1, 9670, 8000)[0]
    got = proc.process(Audio(wave, 8000), vtln_warp=0.85)
    want = orc.compute(proc._build_options(), wave, 0.85)
    assert_close(got.data, want, family='plp', what='warped utterance, unwarped banks impossible')
    with pytest.raises(RuntimeError, match='num_bins too large'):
        proc.process(Audio(wave, 8000))
    with pytest.raises(RuntimeError, match='num_bins too large'):
        orc.compute(proc._build_options(), wave, 1.0)
    # (a frameless utterance - shorter than half a shift with centred frames - asks for no banks at all)
    both = proc._process_batch([Audio(wave, 8000), Audio(wave[:5000], 8000), Audio(wave[:40], 8000)],
                               vtln_warp=[0.85, 0.85, 1.0])
    assert_close(both[1].data, orc.compute(proc._build_options(), wave[:5000], 0.85), family='plp')
    assert both[2].data.size == 0 and orc.compute(proc._build_options(), wave[:40], 1.0).size == 0
    with pytest.raises(RuntimeError, match='num_bins too large'):
        proc._process_batch([Audio(wave, 8000), Audio(wave[:5000], 8000)], vtln_warp=[0.85, 1.0])
    # fbank builds the unwarped banks when the plan is made, like Kaldi's FbankComputer
    with pytest.raises(RuntimeError, match='num_bins too large'):
        FilterbankProcessor(**{k: v for k, v in opts.items() if k != 'num_ceps'}).process(
            Audio(wave, 8000), vtln_warp=0.85)
    with pytest.raises(RuntimeError, match='num_bins too large'):
        fb = FilterbankProcessor(**{k: v for k, v in opts.items() if k != 'num_ceps'})
        orc.compute(fb._build_options(), wave, 0.85)


def test_adversarial_waveforms(gpu):
    """digital silence, constants, full-scale squares, lone impulses, clipped noise, +-1 LSB noise, steps and
    bursts at 8 / 16 / 22.05 / 32 / 44.1 kHz through every mel family (and the pitch tracker, bit for bit):
    tools/adversarial_parity.py finds nothing outside the suite's tolerances but pure tones at their float32 floor"""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'adversarial_parity.py')
    spec = importlib.util.spec_from_file_location('adversarial_parity', path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    argv, sys_argv = ['adversarial_parity.py', '3'], __import__('sys').argv
    __import__('sys').argv = argv
    try:
        assert module.main() == 0
    finally:
        __import__('sys').argv = sys_argv


@pytest.mark.parametrize('opts', [
    dict(num_bins=80), dict(num_bins=80, use_energy=True), dict(num_bins=80, use_energy=True, htk_compat=True),
    dict(num_bins=80, use_energy=True, raw_energy=False, snip_edges=False), dict(num_bins=66, low_freq=0),
    dict(num_bins=100, high_freq=-200, window_type='hamming', use_log_fbank=False),
    dict(num_bins=128, low_freq=700, vtln_low=800), dict(num_bins=112, low_freq=400, vtln_low=500, frame_length=0.03)])
def test_wide_filterbanks(gpu, synth_waves, opts):
    """filterbanks of 65 ... 128 bins (fbank-80 at 16 kHz) run the 64-bin kernel twice, over the two halves of the
    bank (capi.hip: snf_plan::wide) - until round 6 they fell to the generic kernel; the energy column goes with
    the half it is adjacent to; VTLN batches of such banks stay on the generic kernel; alone == in a batch"""
    waves = list(synth_waves)
    proc = FilterbankProcessor(dither=0, **opts)
    feats = proc._process_batch([Audio(w, 16000) for w in waves])
    plan = _backend.get_plan(proc._build_options())
    assert plan.kernel_name(1) in ('fbank512b_kernel', 'fbank512_kernel')
    for w, f in zip(waves, feats):
        want = _oracle(proc, w)
        assert f.shape == want.shape and f.shape[1] == proc.ndims
        # (linear mel energies: no logarithm in front of the relative bound, bins 60 dB under the loudest: 2e-4 measured; the fuzzers carry 5e-4 there)
        assert_close(f.data, want, rtol=1e-4 if proc.use_log_fbank else 5e-4, what=f'{proc.name} {opts}')
    alone = proc._process_batch([Audio(waves[2], 16000)])[0]
    assert np.array_equal(alone.data, feats[2].data)
    warps = [0.9 + 0.05 * i for i in range(len(waves))]
    try:
        [_oracle(proc, w, wf) for w, wf in zip(waves, warps)]
    except RuntimeError:   # (a warped bank of that many bins has an empty bin: a Kaldi-class error on both sides)
        with pytest.raises(RuntimeError):
            proc._process_batch([Audio(w, 16000) for w in waves], vtln_warp=warps)
        return
    feats = proc._process_batch([Audio(w, 16000) for w in waves], vtln_warp=warps)
    assert plan.kernel_name(1) == 'mel_features_generic_kernel'
    for w, wf, f in zip(waves, warps, feats):
        assert_close(f.data, _oracle(proc, w, wf), rtol=1e-4 if proc.use_log_fbank else 5e-4,
                     what=f'{proc.name} {opts} warp {wf}')


@pytest.mark.parametrize('opts', [
    dict(num_bins=40, num_ceps=40), dict(num_bins=40, num_ceps=40, use_energy=False),
    dict(num_bins=40, num_ceps=24, htk_compat=True), dict(num_bins=40, num_ceps=17, htk_compat=True, use_energy=False),
    dict(num_bins=80, num_ceps=13), dict(num_bins=80, num_ceps=40, raw_energy=False, cepstral_lifter=0.0),
    dict(num_bins=30, num_ceps=30, snip_edges=False, energy_floor=5.0), dict(num_bins=64, num_ceps=64, low_freq=60)])
def test_mfcc_through_the_filterbank_kernel(gpu, synth_waves, opts):
    """MFCC plans with more than 16 cepstra (Kaldi's "hires" MFCC: 40 bins, 40 cepstra) or more than 64 bins: the
    filterbank kernel writes [log energy |] log-mel rows, mfcc_dct_kernel forms the cepstra (DCT, lifter, c0 :=
    energy, htk order) - until round 6 the generic kernel took these plans; VTLN batches still do; alone == in a
    batch, an utterance shorter than a window included (snip_edges = False)"""
    waves = list(synth_waves) + [synth.utterances(77, 1, 300)[0]]
    proc = MfccProcessor(dither=0, **opts)
    feats = proc._process_batch([Audio(w, 16000) for w in waves])
    plan = _backend.get_plan(proc._build_options())
    names = {plan.kernel_name(k) for k in range(1, 6)} - {None}
    assert 'mfcc_dct_kernel' in names and names & {'fbank512b_kernel', 'fbank512_kernel'}, names
    for w, f in zip(waves, feats):
        want = _oracle(proc, w)
        assert f.shape == want.shape
        assert_close(f.data, want, rtol=1e-4, what=f'{proc.name} {opts}')
    alone = proc._process_batch([Audio(waves[1], 16000)])[0]
    assert np.array_equal(alone.data, feats[1].data)
    warps = [0.9 + 0.04 * i for i in range(len(waves))]
    feats = proc._process_batch([Audio(w, 16000) for w in waves], vtln_warp=warps)
    assert plan.kernel_name(1) == 'mel_features_generic_kernel'
    for w, wf, f in zip(waves, warps, feats):
        assert_close(f.data, _oracle(proc, w, wf), rtol=1e-4, what=f'{proc.name} {opts} warp {wf}')
