"""The observable behaviour the reference's own processor tests pin (shapes, error classes,
column relations), run against the HIP path: reference test/processor/test_filterbank.py,
test_mfcc.py, test_plp.py, test_spectrogram.py, test_pitch_kaldi.py and
test/postprocessor/test_delta.py.  (Parameter round-trips are host-only: tests/test_host_api.py.)"""

import numpy as np
import pytest

from shennong_amd import Audio, Features
from shennong_amd.processor import (
    FilterbankProcessor, KaldiPitchPostProcessor, KaldiPitchProcessor, MfccProcessor, PlpProcessor,
    SpectrogramProcessor)
from shennong_amd.postprocessor import DeltaPostProcessor

pytestmark = pytest.mark.gpu

STEREO = Audio(np.random.default_rng(0).random((1000, 2)), sample_rate=16000)


@pytest.mark.parametrize('cls, ndims', [(FilterbankProcessor, 23), (MfccProcessor, 13),
                                        (PlpProcessor, 13), (KaldiPitchProcessor, 2)])
def test_output_shapes_and_bad_signals(gpu, audio, cls, ndims):
    assert cls(frame_shift=0.01).process(audio).shape == (140, ndims)
    assert cls(frame_shift=0.02).process(audio).shape == (70, ndims)
    assert cls(frame_shift=0.02, frame_length=0.05).process(audio).shape == (69, ndims)
    with pytest.raises(ValueError):  # sample rate mismatch
        cls(sample_rate=8000).process(audio)
    with pytest.raises(ValueError):  # only mono signals are accepted
        cls(sample_rate=STEREO.sample_rate).process(STEREO)


@pytest.mark.parametrize('use_energy', [True, False])
@pytest.mark.parametrize('num_bins', [0, 1, 10, 23, 30])
def test_filterbank_num_bins(gpu, audio, use_energy, num_bins):
    proc = FilterbankProcessor(use_energy=use_energy, num_bins=num_bins)
    assert proc.ndims == num_bins + use_energy
    if num_bins >= 3:
        assert proc.process(audio).shape == (140, num_bins + use_energy)
    else:
        with pytest.raises(RuntimeError):
            proc.process(audio)


def test_filterbank_energy_column(gpu, audio):
    p1 = FilterbankProcessor(use_energy=False).process(audio)
    p2 = FilterbankProcessor(use_energy=True).process(audio)
    assert p1.shape[1] == p2.shape[1] - 1
    assert p1.data == pytest.approx(p2.data[:, 1:], rel=1e-1)  # (dither is on in both)


def test_mfcc_dither_off_three_ways(gpu, audio):
    p1 = MfccProcessor()
    p1.dither = 0
    p3 = MfccProcessor()
    p3.set_params(**{'dither': 0})
    assert p1.process(audio) == MfccProcessor(dither=0).process(audio) == p3.process(audio)


def test_mfcc_column_signal(gpu, audio):
    column = Audio(audio.data.reshape((audio.nsamples, 1)), audio.sample_rate)
    assert MfccProcessor().process(column).shape == (140, 13)


@pytest.mark.parametrize('num_ceps', [0, 1, 5, 13, 23, 25])
def test_mfcc_num_ceps(gpu, audio, num_ceps):
    proc = MfccProcessor(num_ceps=num_ceps)
    if 0 < proc.num_ceps <= proc.num_bins:
        assert proc.process(audio).shape == (140, num_ceps)
        proc.use_energy = False
        assert proc.process(audio).shape == (140, num_ceps)
    else:
        with pytest.raises(RuntimeError):
            proc.process(audio)


@pytest.mark.parametrize('num_bins', [0, 1, 5, 23])
def test_mfcc_num_bins(gpu, audio, num_bins):
    proc = MfccProcessor(num_bins=num_bins)
    proc.num_ceps = min(proc.num_ceps, num_bins)
    if proc.num_bins >= 3:
        assert proc.process(audio).shape == (140, proc.num_ceps)
        proc.use_energy = False
        assert proc.process(audio).shape == (140, proc.num_ceps)
    else:
        with pytest.raises(RuntimeError):
            proc.process(audio)


@pytest.mark.parametrize('cls, factor', [(MfccProcessor, 2 ** 0.5), (PlpProcessor, 1.0)])
def test_htk_compat(gpu, audio, cls, factor):
    p1 = cls(use_energy=True, htk_compat=False, dither=0).process(audio)
    p2 = cls(use_energy=True, htk_compat=True, dither=0).process(audio)
    assert p1.data[:, 0] == pytest.approx(p2.data[:, -1])
    p1 = cls(use_energy=False, htk_compat=False, dither=0).process(audio)
    p2 = cls(use_energy=False, htk_compat=True, dither=0).process(audio)
    assert p1.data[:, 0] * factor == pytest.approx(p2.data[:, -1])


@pytest.mark.parametrize('sample_rate', [8000, 44100])
def test_mfcc_resampled_audio(gpu, audio, sample_rate):
    import scipy.signal
    n = int(audio.nsamples * sample_rate / audio.sample_rate)
    resampled = Audio(scipy.signal.resample(audio.data, n).astype(audio.dtype), sample_rate,
                      validate=False)
    assert MfccProcessor(sample_rate=sample_rate).process(resampled).shape == (140, 13)
    with pytest.raises(ValueError, match='mismatch in sample rate'):
        MfccProcessor().process(resampled)


@pytest.mark.parametrize('dtype', [np.int16, np.int32, np.float32, np.float64])
def test_mfcc_any_audio_dtype(gpu, audio, dtype):
    """whatever the sample type, the processor works on the int16 signal (reference
    test_mfcc.py:144-173 compares with the wave Kaldi itself reads: the int16 values as float32)"""
    as_kaldi = Audio(audio.data.astype(np.float32) / 2 ** 15, audio.sample_rate, validate=True)
    assert as_kaldi.dtype == np.float32 and as_kaldi.is_valid()
    converted = audio.astype(dtype)
    assert converted.duration == as_kaldi.duration and converted.dtype == dtype and converted.is_valid()
    mfcc = MfccProcessor(dither=0).process(converted)
    mfcc_kaldi = MfccProcessor(dither=0).process(as_kaldi)
    assert mfcc.shape == mfcc_kaldi.shape and mfcc.dtype == mfcc_kaldi.dtype
    assert np.array_equal(mfcc.times, mfcc_kaldi.times)
    assert mfcc.properties == mfcc_kaldi.properties
    assert mfcc.data == pytest.approx(mfcc_kaldi.data)


@pytest.mark.parametrize('num_ceps', [-1, 0, 1, 5, 13, 23, 25])
def test_plp_num_ceps(gpu, audio, num_ceps):
    if num_ceps >= 23:
        with pytest.raises(ValueError, match=r'We must have num_ceps <= lpc_order\+1'):
            PlpProcessor(num_ceps=num_ceps)
    elif num_ceps > 0:
        proc = PlpProcessor(num_ceps=num_ceps)
        assert proc.num_ceps == num_ceps == proc.ndims
        assert proc.process(audio).shape == (140, num_ceps)
        proc.use_energy = False
        assert proc.process(audio).shape == (140, num_ceps)
    else:
        with pytest.raises(ValueError, match='must be > 0'):
            PlpProcessor(num_ceps=num_ceps)


def test_plp_outputs(gpu, audio):
    assert PlpProcessor(cepstral_lifter=0, cepstral_scale=0.9).process(audio).shape == (140, 13)
    assert PlpProcessor(snip_edges=False).process(audio).shape == (142, 13)
    assert PlpProcessor(snip_edges=False, rasta=True).process(audio).shape == (142, 13)
    feat = PlpProcessor(use_energy=True, raw_energy=False, energy_floor=np.exp(50)).process(audio)
    assert feat.shape == (140, 13)
    assert np.all(feat.data[:, 0] == 50)


def test_spectrogram(gpu, audio):
    noise = Audio(np.random.default_rng(1).random((10, 2)), 50)
    with pytest.raises(ValueError):
        SpectrogramProcessor(sample_rate=noise.sample_rate).process(noise)
    with pytest.raises(ValueError, match='mismatch in sample rates'):
        SpectrogramProcessor(sample_rate=noise.sample_rate + 1).process(audio)
    proc = SpectrogramProcessor(sample_rate=audio.sample_rate)
    feats = proc.process(audio)
    assert feats.shape == (140, 257) and feats.shape[1] == proc.ndims


def test_pitch_post(gpu, audio):
    raw = KaldiPitchProcessor().process(audio)
    post = KaldiPitchPostProcessor()
    params = post.get_params()
    data = post.process(raw)
    assert data.shape[1] == 3 and raw.shape[0] == data.shape[0]
    assert np.array_equal(raw.times, data.times)
    assert params == post.get_params()
    for cols in (1, 3):
        bad = Features(np.random.default_rng(2).random((raw.nframes, cols)), raw.times)
        with pytest.raises(ValueError) as err:
            post.process(bad)
        assert f'data shape must be (_, 2), but it is (_, {cols})' in str(err.value)
    for options in ((True, True, True, True), (True, True, True, False), (False, False, True, True),
                    (False, False, False, False)):
        proc = KaldiPitchPostProcessor(
            add_pov_feature=options[0], add_normalized_log_pitch=options[1],
            add_delta_pitch=options[2], add_raw_log_pitch=options[3])
        if sum(options):
            out = proc.process(raw)
            assert proc.ndims == sum(options) and out.shape == (raw.shape[0], sum(options))
            assert np.array_equal(raw.times, out.times) and out.times.shape[1] == 2
        else:  # all False is not supported by Kaldi
            with pytest.raises(ValueError, match='must be True'):
                proc.process(raw)


@pytest.mark.parametrize('order', [0, 1, 2, 5])
@pytest.mark.parametrize('window', [1, 2, 5])
def test_delta_output(gpu, audio, order, window):
    mfcc = MfccProcessor(dither=0).process(audio)
    delta = DeltaPostProcessor(order=order, window=window).process(mfcc)
    assert delta.shape == (mfcc.shape[0], mfcc.shape[1] * (order + 1))
    assert np.array_equal(delta.times, mfcc.times)
    assert delta.data[:, :mfcc.shape[1]] == pytest.approx(mfcc.data)
    with pytest.raises(ValueError, match='output dimension for delta processor depends on input'):
        DeltaPostProcessor().ndims
