"""Host-side logic: the reference's non-compute tests re-expressed against shennong_amd
(parameter surfaces, error behaviour before any device call, Frames, window, Features, Audio)."""

import os

import numpy as np
import pytest

from conftest import GOLDEN
from shennong_amd import Audio, Features, Utterances, _abi, window
from shennong_amd.frames import Frames
from shennong_amd.processor import (
    FilterbankProcessor, MfccProcessor, PlpProcessor, SpectrogramProcessor,
    KaldiPitchProcessor, KaldiPitchPostProcessor)
from shennong_amd.postprocessor import DeltaPostProcessor


@pytest.fixture(scope='module')
def golden():
    return np.load(os.path.join(GOLDEN, 'reference_numpy.npz'))


# ---- parameter surfaces (reference test_filterbank.py:10-35, test_mfcc.py:12-45, test_plp.py:12-25,
#      test_pitch_kaldi.py:16-36,60-80, test_delta.py:9-23) -------------------------------------------
def test_param_counts_and_names():
    frame = ['blackman_coeff', 'dither', 'frame_length', 'frame_shift', 'preemph_coeff',
             'remove_dc_offset', 'round_to_power_of_two', 'sample_rate', 'snip_edges', 'window_type']
    mel = ['high_freq', 'low_freq', 'num_bins', 'vtln_high', 'vtln_low']
    assert sorted(FilterbankProcessor().get_params()) == sorted(frame + mel + [
        'use_energy', 'energy_floor', 'raw_energy', 'htk_compat', 'use_log_fbank', 'use_power'])
    assert sorted(MfccProcessor().get_params()) == sorted(frame + mel + [
        'num_ceps', 'use_energy', 'energy_floor', 'raw_energy', 'cepstral_lifter', 'htk_compat'])
    assert len(PlpProcessor().get_params()) == 25
    assert len(SpectrogramProcessor().get_params()) == 12
    assert len(KaldiPitchProcessor().get_params()) == 13
    assert len(KaldiPitchPostProcessor().get_params()) == 13
    assert DeltaPostProcessor().get_params() == {'order': 2, 'window': 2}


def test_filterbank_params_roundtrip():
    params = {'num_bins': 0, 'use_energy': True, 'energy_floor': 10.0, 'raw_energy': False,
              'htk_compat': True, 'use_log_fbank': False, 'use_power': False}
    p = FilterbankProcessor(**params)
    out = p.get_params()
    assert len(out) == 21
    for k, v in params.items():
        assert out[k] == v
    q = FilterbankProcessor()
    q.set_params(**out)
    assert q.get_params() == out
    assert FilterbankProcessor(use_energy=True, num_bins=10).ndims == 11


def test_mfcc_set_params():
    m = MfccProcessor()
    assert m.get_params()['sample_rate'] == 16000
    m.set_params(sample_rate=0)
    assert m.get_params()['sample_rate'] == 0
    m.set_params(window_type='hanning')
    assert m.get_params()['window_type'] == 'hanning'
    with pytest.raises(ValueError):
        m.set_params(window_type='foo')
    with pytest.raises(ValueError):
        m.set_params(not_a_param=1)
    assert isinstance(m.frame_shift, np.float32) and isinstance(m.low_freq, np.float32)
    assert m.ndims == m.num_ceps == 13 and m.name == 'mfcc'


def test_plp_num_ceps_validation():
    for bad in (23, 25):
        with pytest.raises(ValueError) as err:
            PlpProcessor(num_ceps=bad)
        assert 'We must have num_ceps <= lpc_order+1' in str(err.value)
    for bad in (-1, 0):
        with pytest.raises(ValueError) as err:
            PlpProcessor(num_ceps=bad)
        assert 'must be > 0' in str(err.value)
    p = PlpProcessor(num_ceps=5)
    assert p.ndims == 5 and isinstance(p.compress_factor, np.float32)


def test_pitch_params():
    opts = {k: 0 for k in [
        'sample_rate', 'frame_shift', 'frame_length', 'min_f0', 'max_f0', 'soft_min_f0',
        'penalty_factor', 'lowpass_cutoff', 'resample_freq', 'delta_pitch', 'nccf_ballast',
        'lowpass_filter_width', 'upsample_filter_width']}
    assert KaldiPitchProcessor(**opts).get_params() == opts
    assert KaldiPitchProcessor().set_params(**opts).get_params() == opts
    post = {
        'pitch_scale': 0, 'pov_scale': 0, 'pov_offset': 0, 'delta_pitch_scale': 0,
        'delta_pitch_noise_stddev': 0, 'normalization_left_context': 0,
        'normalization_right_context': 0, 'delta_window': 0, 'delay': 0,
        'add_pov_feature': bool(10), 'add_normalized_log_pitch': False,
        'add_delta_pitch': False, 'add_raw_log_pitch': False}
    assert KaldiPitchPostProcessor(**post).get_params() == post
    assert KaldiPitchPostProcessor().set_params(**post).get_params() == post
    assert KaldiPitchProcessor().ndims == 2 and KaldiPitchPostProcessor().ndims == 3


def test_delta_params():
    d = DeltaPostProcessor()
    d.order = 0
    with pytest.raises(ValueError):
        d.window = 0
    with pytest.raises(ValueError):
        d.window = 2000
    d.window = 1
    assert d.get_params() == {'order': 0, 'window': 1}
    with pytest.raises(ValueError) as err:
        DeltaPostProcessor().ndims
    assert 'output dimension for delta processor depends on input' in str(err.value)


# ---- errors raised before any device call (reference test_filterbank.py:69-77 & co) ------------------
@pytest.mark.parametrize('cls', [
    FilterbankProcessor, MfccProcessor, PlpProcessor, SpectrogramProcessor, KaldiPitchProcessor])
def test_bad_signals(audio, cls):
    with pytest.raises(ValueError) as err:
        cls(sample_rate=8000).process(audio)
    assert 'mismatch in sample rates' in str(err.value)
    stereo = Audio(np.random.random((1000, 2)), sample_rate=16000)
    with pytest.raises(ValueError) as err:
        cls(sample_rate=stereo.sample_rate).process(stereo)
    assert 'must have one' in str(err.value)


def test_pitch_post_errors(audio):
    times = KaldiPitchProcessor().times(10)
    props = {'pipeline': [{'name': 'pitch', 'columns': [0, 1]}], 'pitch': {}}
    for cols in (1, 3):
        bad = Features(np.random.random((10, cols)), times, properties=props)
        with pytest.raises(ValueError) as err:
            KaldiPitchPostProcessor().process(bad)
        assert f'data shape must be (_, 2), but it is (_, {cols})' in str(err.value)
    p = KaldiPitchPostProcessor(add_pov_feature=False, add_normalized_log_pitch=False,
                                add_delta_pitch=False, add_raw_log_pitch=False)
    with pytest.raises(ValueError) as err:
        p.process(Features(np.random.random((10, 2)), times, properties=props))
    assert 'must be True' in str(err.value)


def test_process_all_argument_checks(wav_file):
    """reference test_parallel.py:34-70: validated before any audio is processed"""
    utts = Utterances([('u1', wav_file, 0, 0.2), ('u2', wav_file, 0, 0.2), ('u3', wav_file, 0, 0.2)])
    proc = MfccProcessor()
    with pytest.raises(ValueError) as err:
        proc.process_all(utts, njobs=0)
    assert 'must be strictly positive' in str(err.value)
    with pytest.raises(ValueError) as err:
        proc.process_all(utts, vtln_warp=1.0)
    assert 'is not a dict' in str(err.value)
    with pytest.raises(ValueError) as err:
        proc.process_all(utts, vtln_warp={f'{n}': 1.0 for n in range(2)})
    assert 'have different names' in str(err.value)
    assert utts['u1'].load_audio().nsamples == 3200


# ---- times (row T of SURVEY.md §8a): bit-exact float64 multiples of the float32 shift ---------------
def test_times_bit_exact(golden):
    t = MfccProcessor().times(140)
    assert t.dtype == np.float64 and np.array_equal(t, golden['times_140'])
    assert t[1, 0] == float(np.float32(0.01)) != 0.01
    p = KaldiPitchProcessor().times(3)
    assert p[1, 0] == 0.01 and p[0, 1] == 0.025
    assert np.allclose(Frames().times(1600), np.asarray(
        [[0, 0.025], [0.01, 0.035], [0.02, 0.045], [0.03, 0.055], [0.04, 0.065],
         [0.05, 0.075], [0.06, 0.085], [0.07, 0.095]]))


def test_properties_layout():
    p = FilterbankProcessor(num_bins=40).get_properties(vtln_warp=1.0)
    assert p['pipeline'] == [{'name': 'filterbank', 'columns': [0, 39]}]
    assert p['filterbank']['vtln_warp'] == 1.0 and len(p['filterbank']) == 22
    f = Features(np.zeros((3, 13), np.float32), np.arange(3.), properties={
        'pipeline': [{'name': 'mfcc', 'columns': [0, 12]}], 'mfcc': {}})
    q = DeltaPostProcessor().get_properties(f)
    assert q['pipeline'][-1] == {'name': 'delta', 'columns': [0, 38]}
    assert q['delta'] == {'order': 2, 'window': 2}


# ---- Frames (reference test/test_frames.py) -----------------------------------------------------------
def test_frames_params():
    p = {'sample_rate': 1, 'frame_shift': 1, 'frame_length': 1, 'snip_edges': False}
    assert Frames(**p).get_params() == p
    assert Frames().set_params(**p).get_params() == p


@pytest.mark.parametrize('snip_edges', [True, False])
def test_frames_literals(snip_edges):
    f = Frames(sample_rate=1, snip_edges=snip_edges)
    f.frame_shift, f.frame_length = 1, 1
    assert f.nframes(10) == 10 and f.samples_per_frame == 1 and f.samples_per_shift == 1
    assert np.array_equal(f.boundaries(10), np.repeat(np.arange(10), 2).reshape(10, 2) + (0, 1))
    assert np.array_equal(f.make_frames(np.arange(10)), np.arange(10)[:, np.newaxis])
    f.frame_length = 2
    n = 9 if snip_edges else 10
    assert f.nframes(10) == n
    framed = np.arange(n).repeat(2).reshape(n, 2) + (0, 1)
    if not snip_edges:
        framed[-1, -1] = 8
    assert np.array_equal(f.make_frames(np.arange(10)), framed)
    f.frame_length = 3
    n = 8 if snip_edges else 10
    framed = np.arange(n).repeat(3).reshape(n, 3) + (0, 1, 2)
    if not snip_edges:
        framed[-2, -1] = 8
        framed[-1, -2:] = (8, 7)
    assert np.array_equal(f.make_frames(np.arange(10)), framed)
    f.frame_length, f.frame_shift = 5, 3
    n = 2 if snip_edges else 3
    assert f.nframes(9) == n
    assert np.array_equal(f.boundaries(n), np.repeat(np.arange(n) * 3, 2).reshape(n, 2) + (0, 5))
    framed = (np.arange(n) * 3).repeat(5).reshape(n, 5) + (0, 1, 2, 3, 4)
    if not snip_edges:
        framed[-1, -1] = 8
    assert np.array_equal(f.make_frames(np.arange(10)), framed)
    f.frame_length, f.frame_shift = 2, 2
    assert f.nframes(10) == 5
    f.frame_length, f.frame_shift = 1, 2
    assert np.array_equal(f.make_frames(np.arange(10)), (np.arange(5) * 2)[:, np.newaxis])


@pytest.mark.parametrize('ndim, snip_edges, writeable', [
    (n, bool(s), bool(w)) for n in (1, 2, 3) for s in (0, 1) for w in (0, 1)])
def test_make_frames(ndim, snip_edges, writeable):
    f = Frames(snip_edges=snip_edges)
    shape = (2400,) + (2,) * (ndim - 1)
    aref = np.random.random(shape)
    array = np.copy(aref)
    frames = f.make_frames(array, writeable=writeable)
    assert np.array_equal(array, aref)
    assert frames.shape == (f.nframes(aref.shape[0]), f.samples_per_frame) + aref.shape[1:]
    if writeable is False:
        with pytest.raises(ValueError):
            frames[0] = 0
    else:
        frames[0] = -1
        assert (frames[0] == -1).all()


# ---- window (reference test/test_window.py) --------------------------------------------------------------
@pytest.mark.parametrize('type, length', [(t, l) for t in window.types() for l in (1, 2, 3, 10, 100)])
def test_window(type, length):
    win = window.window(length, type=type)
    assert win.ndim == 1 and win.shape == (length,)
    assert not np.any(np.isnan(win)) and win.max() <= 1.0 and win.min() >= 0.0
    assert not np.all(win == 0.0)
    if type == 'rectangular':
        assert np.all(win == 1.0)
    elif length > 2:
        assert not np.all(win == 1.0)
    if type == 'povey' and length > 2:
        assert win[0] == win[-1] == 0.0


def test_window_known_answers_and_errors():
    assert np.array_equal(window.window(5, type='hamming'),
                          np.array([0.08, 0.54, 1., 0.54, 0.08], dtype=np.float32))
    assert window.window(5, type='povey').tolist() == [
        0.0, 0.5547847151756287, 1.0, 0.5547847151756287, 0.0]
    for length in (-2, 0):
        with pytest.raises(ValueError) as err:
            window.window(length)
        assert 'length must be strictly positive' in str(err.value)
    for type in ('spam', 'pove'):
        with pytest.raises(ValueError) as err:
            window.window(10, type=type)
        assert 'type must be in' in str(err.value)


# ---- Audio / Features against fixtures captured from the reference's numpy code ---------------------------
def test_audio_astype_golden(golden):
    for src, dst in [('f32', 'i16'), ('f64', 'i16'), ('i32', 'i16'), ('i16', 'f32')]:
        data = golden[f'astype_{src}']
        dtype = {'i16': np.int16, 'f32': np.float32}[dst]
        got = Audio(data, 16000, validate=False).astype(dtype).data
        assert got.dtype == dtype
        assert np.array_equal(got, golden[f'astype_{src}_to_{dst}']), (src, dst)
    a = Audio(golden['astype_i16'], 16000)
    assert a.astype(np.int16) is a
    assert np.array_equal(a.astype(np.int32).data, golden['astype_i16'].astype(np.int32) * 32768)
    with pytest.raises(ValueError):
        a.astype(np.int8)


def test_audio_basics(audio, audio_8k):
    assert audio.sample_rate == 16000 and audio.nsamples == 22713 and audio.nchannels == 1
    assert audio.dtype == np.int16 and audio.is_valid() and audio.duration == 22713 / 16000
    assert audio_8k.sample_rate == 8000
    chunk = audio.segment([(0.0, 0.2)])[0]
    assert chunk.nsamples == 3200
    with pytest.raises(ValueError):
        audio.segment([(0.3, 0.2)])
    with pytest.raises(ValueError):
        Audio(np.array([2.0, 0.0]), 16000)
    col = Audio(audio.data.reshape((audio.nsamples, 1)), audio.sample_rate)
    assert col.shape == (22713,) and col == audio


def test_features_concatenate_golden(golden):
    fa = Features(golden['concat_d1'], golden['concat_t1'], properties={
        'pipeline': [{'name': 'mfcc', 'columns': [0, 2]}], 'mfcc': {'a': 1}})
    fb = Features(golden['concat_d2'], golden['concat_t2'], properties={
        'pipeline': [{'name': 'pitch', 'columns': [0, 1]}], 'pitch': {'b': 2}})
    fc = fa.concatenate(fb, tolerance=2)
    assert np.array_equal(fc.data, golden['concat_data'])
    assert np.array_equal(fc.times, golden['concat_times'])
    assert repr(fc.properties['pipeline']) == str(golden['concat_pipeline'])
    with pytest.raises(ValueError) as err:
        fa.concatenate(fb, tolerance=1)
    assert str(err.value) == str(golden['concat_err_tol'])
    with pytest.raises(ValueError) as err:
        fa.concatenate(fb)
    assert str(err.value) == str(golden['concat_err_notol'])


def test_features_validate_and_equality():
    data = (np.random.default_rng(0).random((5, 3)) + 0.5).astype(np.float32)
    times = np.vstack((np.arange(5) * 0.01, np.arange(5) * 0.01 + 0.025)).T
    f = Features(data, times, properties={'a': np.arange(3)})
    assert f.is_valid() and f.shape == (5, 3) and f.ndims == 3 and f.nframes == 5
    assert f == Features(data.copy(), times.copy(), properties={'a': [0, 1, 2]})
    assert f.is_close(Features(data + 1e-7, times, properties={'a': np.arange(3)}))
    assert not f == Features(data + 1e-7, times, properties={'a': np.arange(3)})
    assert not f.is_close(Features(data, times + 1e-9, properties={'a': np.arange(3)}))
    with pytest.raises(ValueError) as err:
        Features(data, times[::-1])
    assert 'not sorted' in str(err.value)
    with pytest.raises(ValueError) as err:
        Features(data, times[:4])
    assert 'mismatch in number of frames' in str(err.value)
    bad = data.copy()
    bad[2, 1] = np.nan
    with pytest.raises(ValueError) as err:
        Features(bad, times)
    assert 'non-finite' in str(err.value)
    assert Features(np.zeros((0, 0), np.float32), np.zeros((0, 2))).is_valid()
    assert f.copy(subsample=2).shape == (3, 3)


# ---- energy / VAD / CMVN host logic (reference test_energy.py, test_vad.py, test_cmvn.py) ----------
def test_energy_params():
    from shennong_amd.processor import EnergyProcessor
    c = {'window_type': 'hanning', 'compression': 'sqrt', 'dither': 0}
    p1 = EnergyProcessor(**c)
    p2 = EnergyProcessor().set_params(**c)
    assert p1.get_params() == p2.get_params()
    assert p1.ndims == 1 and p1.name == 'energy'
    assert len(p1.get_params()) == 12
    with pytest.raises(ValueError) as err:
        p1.compression = 'bad'
    assert 'compression must be in ' in str(err.value)
    # raw energy = rectangular window, no pre-emphasis, only inside the options record
    o = p1._build_options()
    assert o.kind == _abi.KIND_ENERGY and o.raw_energy == 1 and o.compression == 2
    assert p1.window_type == 'hanning'


def test_vad_params():
    from shennong_amd.postprocessor import VadPostProcessor
    p = VadPostProcessor()
    with pytest.raises(ValueError) as err:
        p.energy_mean_scale = -1
    assert 'must be >= 0' in str(err.value)
    with pytest.raises(ValueError) as err:
        p.frames_context = -1
    assert 'must be >= 0' in str(err.value)
    with pytest.raises(ValueError) as err:
        p.proportion_threshold = 0
    assert 'must be in ]0, 1[' in str(err.value)
    p = VadPostProcessor(energy_threshold=0, energy_mean_scale=0, frames_context=0,
                         proportion_threshold=0.1)
    assert p.get_params() == pytest.approx({
        'energy_threshold': 0, 'energy_mean_scale': 0, 'frames_context': 0,
        'proportion_threshold': 0.1})
    assert p.ndims == 1 and p.name == 'vad'


@pytest.mark.parametrize('dim', [-2, 0, 1, 3, 2.54, 'a'])
def test_cmvn_dim(dim):
    from shennong_amd.postprocessor import CmvnPostProcessor
    if dim in (1, 3):
        assert CmvnPostProcessor(dim).dim == dim
    else:
        with pytest.raises(ValueError) as err:
            CmvnPostProcessor(dim)
        assert 'dimension must be a strictly positive integer' in str(err.value)


def test_cmvn_params_and_errors():
    from shennong_amd import Features, FeaturesCollection
    from shennong_amd.postprocessor import (
        CmvnPostProcessor, SlidingWindowCmvnPostProcessor, apply_cmvn)
    c = CmvnPostProcessor(dim=1, stats=None)
    assert c.get_params()['dim'] == 1
    assert c.get_params()['stats'].shape == (2, 2)
    assert c.get_params()['stats'].dtype == np.float64
    assert c.get_params()['stats'].sum() == 0.0
    with pytest.raises(ValueError) as err:
        c.set_params(dim=None)
    assert 'cannot set attribute dim for CmvnPostProcessor' in str(err.value)
    with pytest.raises(ValueError) as err:
        c.set_params(stats=None)
    assert 'cannot set attribute stats for CmvnPostProcessor' in str(err.value)
    with pytest.raises(ValueError) as err:
        CmvnPostProcessor(13, stats=1)
    assert 'shape (2, 14), but is shaped as ()' in str(err.value)
    with pytest.raises(ValueError) as err:
        CmvnPostProcessor(13, stats=np.random.random((2, 13)))
    assert 'shape (2, 14), but is shaped as (2, 13)' in str(err.value)
    stats = np.random.random((2, 14))
    assert stats == pytest.approx(CmvnPostProcessor(13, stats=stats.copy()).stats)

    feats = Features(np.random.random((10, 13)).astype(np.float32), np.arange(10.0))
    proc = CmvnPostProcessor(13)
    with pytest.raises(ValueError) as err:
        proc.process(feats)
    assert 'insufficient accumulation of stats' in str(err.value)
    with pytest.raises(ValueError) as err:
        proc.accumulate(feats, weights=np.asarray([[1, 2], [3, 4]]))
    assert 'weights must have a single dimension' in str(err.value)
    with pytest.raises(ValueError) as err:
        proc.accumulate(feats, weights=np.asarray([]))
    assert 'there is 0 weights but 10 feature frames' in str(err.value)

    coll = FeaturesCollection(a=feats, b=feats.copy())
    with pytest.raises(ValueError) as err:
        apply_cmvn(coll, weights={})
    assert 'keys differ for ' in str(err.value)
    for sd in ([-1], [13]):
        with pytest.raises(ValueError) as err:
            apply_cmvn(coll, skip_dims=sd)
        assert 'out of bounds dimensions' in str(err.value)
    coll['new'] = Features(np.random.random((2, 1)), np.asarray([0, 1]))
    with pytest.raises(ValueError) as err:
        apply_cmvn(coll)
    assert 'must have consistent dimensions' in str(err.value)

    s = SlidingWindowCmvnPostProcessor(normalize_variance=True, center=False)
    assert s.get_params() == {'center': False, 'cmn_window': 600, 'min_window': 100,
                              'max_warnings': 5, 'normalize_variance': True}
    with pytest.raises(ValueError) as err:
        s.ndims
    assert 'dimension for sliding window CMVN processor depends on input' in str(err.value)
    o = s._build_options()
    assert (o.sliding_cmvn.center, o.sliding_cmvn.normalize_variance) == (0, 1)


def test_staging_rows_without_device():
    """the staging helper of the host-pointer path: concatenation semantics, and plain memory when a
    page-locked buffer cannot be had (no device here) - only the KIND of host memory changes"""
    import numpy as np
    from shennong_amd import _backend
    mats = [np.arange(6, dtype=np.float64).reshape(3, 2), np.zeros((0, 2)), np.ones((2, 2), np.float32)]
    staged, token = _backend.stage_rows(mats, np.float32)
    assert staged.dtype == np.float32 and staged.shape == (5, 2)
    assert np.array_equal(staged, np.concatenate(mats).astype(np.float32))
    _backend.STAGING.release(token)
    waves = [np.arange(5, dtype=np.int16), np.arange(3, dtype=np.int16)]
    staged, token = _backend.stage_rows(waves, np.int16)
    assert staged.tolist() == [0, 1, 2, 3, 4, 0, 1, 2]
    _backend.STAGING.release(token)
    single, token = _backend.stage_rows([np.arange(4, dtype=np.int32)], np.int16)
    assert single.dtype == np.int16 and single.tolist() == [0, 1, 2, 3]
    # a large request without a device falls back to plain memory (token None)
    big, token = _backend.STAGING.array((1 << 19, 2), np.float32)
    assert big.shape == (1 << 19, 2)
    _backend.STAGING.release(token)


def test_option_descriptors():
    """shennong_amd/_options.py: parameters are views on the processor's one option record"""
    import numpy as np
    from shennong_amd import _abi
    from shennong_amd._options import Option
    from shennong_amd.processor import FilterbankProcessor, MfccProcessor
    from shennong_amd.postprocessor import VadPostProcessor
    proc = MfccProcessor(frame_shift=0.02, dither=0, snip_edges=False, num_ceps=7)
    record = proc._record
    assert record.kind == _abi.KIND_MFCC
    assert (record.frame.frame_shift_ms, record.frame.dither, record.frame.snip_edges, record.num_ceps) \
        == (20.0, 0.0, 0, 7)
    assert isinstance(proc.frame_shift, np.float32) and proc.frame_shift == np.float32(0.02)
    assert proc.snip_edges is False and proc.window_type == 'povey'
    proc.low_freq = 100
    assert record.mel.low_freq == 100.0 and isinstance(proc.low_freq, np.float32)
    # the record handed to the library is a copy: later edits do not reach it
    sent = proc._build_options()
    proc.num_ceps = 9
    assert (sent.num_ceps, proc._build_options().num_ceps) == (7, 9)
    # two processors never share a record; descriptors live on the class and carry the documentation
    other = MfccProcessor()
    assert other.num_ceps == 13 and other._record is not proc._record
    assert isinstance(MfccProcessor.num_ceps, Option) and 'cepstra' in MfccProcessor.num_ceps.__doc__
    assert FilterbankProcessor(use_energy=1).use_energy is True
    # value checks run before anything is stored
    vad = VadPostProcessor()
    with pytest.raises(ValueError, match=r'proportion_threshold must be in \]0, 1\['):
        vad.proportion_threshold = 1.5
    assert vad.proportion_threshold == np.float32(0.6)


def test_paused_gc_restores_the_collector():
    """batched calls pause the cyclic collector while they make their objects (shennong_amd.utils.paused_gc):
    it comes back as it was, also when the body raises and when calls nest or the collector was off"""
    import gc
    from shennong_amd.utils import paused_gc
    assert gc.isenabled()
    with paused_gc():
        assert not gc.isenabled()
        with paused_gc():
            assert not gc.isenabled()
        assert not gc.isenabled()
    assert gc.isenabled()
    with pytest.raises(KeyError):
        with paused_gc():
            raise KeyError('x')
    assert gc.isenabled()
    gc.disable()
    try:
        with paused_gc():
            pass
        assert not gc.isenabled()
    finally:
        gc.enable()
    # sections that overlap on several threads (the batches in flight of a streamed corpus): the
    # first one to END must not switch the collector back on under the others
    import threading
    inside, leave_first, first_left = threading.Barrier(3), threading.Event(), threading.Event()
    seen = []

    def section(first):
        with paused_gc():
            inside.wait()
            if not first:
                first_left.wait()
                seen.append(gc.isenabled())
            else:
                leave_first.wait()
        if first:
            first_left.set()
    threads = [threading.Thread(target=section, args=(k == 0,)) for k in range(2)]
    for t in threads:
        t.start()
    inside.wait()
    leave_first.set()
    for t in threads:
        t.join()
    assert seen == [False] and gc.isenabled()


def test_streamed_batch_default_without_a_device():
    """the streamed pipeline's default batch is derived from the free HBM; without a device (host-logic tests)
    it is the four hours measured best on a 288 GB GPU"""
    from shennong_amd import _backend, pipeline
    if _backend.device_count() < 1:
        assert pipeline.default_batch_duration(1) == 14400.0
        assert pipeline.default_batch_duration(8) == 14400.0
    else:
        assert 600.0 <= pipeline.default_batch_duration(8) <= pipeline.default_batch_duration(1) <= 14400.0


def test_device_pool_follows_the_workload(monkeypatch):
    """round 6: a pool that is full of one leg's block sizes makes room for the next leg's by releasing what was
    parked longest ago (it used to refuse: every buffer of the next leg then went through hipMalloc / hipFree)"""
    from shennong_amd import _backend
    freed = []

    class Lib:
        def snf_free(self, ptr):
            freed.append(ptr.value)
            return 0

        def snf_set_device(self, device):
            return 0
    monkeypatch.setattr(_backend, 'lib', lambda: Lib())
    pool = _backend._DevicePool()
    M = 1 << 20
    pool.limit = 1000 * M
    assert pool.give(0, (400 * M, 1)) and pool.give(0, (400 * M, 2)) and pool._bytes == 800 * M and not freed
    assert pool.take(0, 150 * M) is None and pool.take(0, 300 * M) == (400 * M, 1)   # (at most twice the request)
    assert pool.give(0, (400 * M, 1)) and pool._bytes == 800 * M
    assert pool.give(0, (500 * M, 3)) and freed == [2] and pool._bytes == 900 * M   # 2 was parked longest ago
    assert pool.give(1, (900 * M, 4)) and sorted(freed) == [1, 2, 3] and pool._bytes == 900 * M
    assert not pool.give(0, (1001 * M, 5)) and pool._bytes == 900 * M               # larger than the whole pool
    assert pool.take(1, 900 * M) == (900 * M, 4) and pool._bytes == 0
    pool.give(0, (10 * M, 6))
    pool.clear()
    assert freed[-1] == 6 and pool._bytes == 0
