"""shennong_amd.Audio against the reference's own expectations (reference test/test_audio.py, WAV
paths; the reference decodes flac / mp3 through pydub + ffmpeg and scans files with sox, neither of
which exists offline: those formats must fail with a ValueError that says so).  Host-only."""

import os

import numpy as np
import pytest

from conftest import GOLDEN
from shennong_amd import Audio

DTYPES = [np.int16, np.int32, np.float32, np.float64, float]


def test_scan(wav_file, audio):
    meta = Audio.scan(wav_file)
    assert meta.sample_rate == audio.sample_rate == 16000
    assert meta.nchannels == audio.nchannels == 1
    assert meta.nsamples == audio.nsamples == 22713
    assert meta.duration == pytest.approx(audio.duration)
    assert meta.duration == pytest.approx(1.419, rel=1e-3)
    meta = Audio.scan(os.path.join(GOLDEN, 'test.8k.wav'))
    assert meta.sample_rate == 8000 and meta.nchannels == 1


def test_scan_bad():
    with pytest.raises(ValueError) as err:
        Audio.scan(__file__)
    assert 'cannot scan audio file' in str(err.value)
    with pytest.raises(ValueError) as err:
        Audio.scan('/path/to/some/lost/place')
    assert 'file not found' in str(err.value)


def test_load(wav_file, audio):
    audio2 = Audio.load(wav_file)
    assert audio2.sample_rate == audio.sample_rate == 16000
    assert audio2.nchannels == audio.nchannels == 1
    assert audio2.duration == audio.duration == pytest.approx(1.419, rel=1e-3)
    assert audio2.data.shape == audio.data.shape == (22713,)
    assert audio2.nsamples == audio.nsamples == 22713
    assert audio2.dtype == audio.dtype == np.int16
    assert audio2.precision == audio.precision == 16
    assert np.all(audio2.data == audio.data)
    assert Audio.load(wav_file) is audio2  # the last decoded files are kept (audio.py:240-243)


def test_load_notaudio():
    with pytest.raises(ValueError) as err:
        Audio.load(__file__)
    assert 'Decoding failed' in str(err.value)


def test_load_badfile():
    with pytest.raises(ValueError) as err:
        Audio.load('/spam/spam/with/eggs')
    assert 'file not found' in str(err.value)


def test_save(tmpdir, audio):
    p = str(tmpdir.join('test.wav'))
    audio.save(p)
    with pytest.raises(ValueError) as err:  # cannot overwrite an existing file
        audio.save(p)
    assert 'file already exist' in str(err.value)
    with pytest.raises(ValueError) as err:  # cannot write without extension
        audio.save('toto')
    assert 'cannot write audio file without extension' in str(err.value)
    assert audio == Audio.load(p)
    for ext in ('.flac', '.mp3'):  # (pydub / ffmpeg formats in the reference)
        with pytest.raises(ValueError, match='only WAV files are supported'):
            audio.save(str(tmpdir.join('test' + ext)))


def test_save_float32(tmpdir):
    signal = np.zeros((1000,), dtype=np.float32)
    signal[10] = 1.0
    signal[20] = -1.0
    p = str(tmpdir.join('test2.wav'))
    audio = Audio(signal, 1000)
    assert audio.dtype == np.float32
    audio.save(p)
    meta = Audio.scan(p)
    assert meta.nchannels == 1
    assert meta.nsamples == 1000
    audio2 = Audio.load(p)
    assert audio2.dtype == np.float32
    assert audio2.nchannels == 1
    assert audio2.nsamples == 1000
    assert audio2.data.min() == -1.0
    assert audio2.data.max() == 1.0


def test_equal(audio):
    assert audio == audio
    assert audio == Audio(audio.data, audio.sample_rate)
    assert audio != Audio(audio.data, audio.sample_rate + 1)
    audio2 = Audio(audio.data * 2, audio.sample_rate)
    assert audio.duration == audio2.duration
    assert audio.sample_rate == audio2.sample_rate
    assert audio != audio2


@pytest.mark.parametrize('dtype', DTYPES)
def test_save_load(tmpdir, dtype):
    audio = Audio(np.random.default_rng(0).random((1000, 2)), 16000).astype(dtype)
    audio.save(tmpdir / 'test.wav')
    assert audio == Audio.load(tmpdir / 'test.wav')


def test_shape():
    # audio data shaped (n, 1) must be reshaped as (n,)
    rng = np.random.default_rng(1)
    for d in (rng.random((100,)), rng.random((100, 1))):
        assert Audio(d, 10).shape == (100,)


def test_channels_mono(audio):
    assert audio.nchannels == 1
    assert audio.shape == (audio.nsamples,)
    assert audio.channel(0) == audio
    with pytest.raises(ValueError):
        audio.channel(1)


def test_channels_stereo():
    data = np.random.default_rng(2).random((1000, 2))
    audio2 = Audio(data, sample_rate=16000)
    assert audio2.nchannels == 2
    assert audio2.shape == (1000, 2)
    for index in (0, 1):
        audio1 = audio2.channel(index)
        assert audio1.nchannels == 1
        assert audio1.shape == (1000,)
        assert all(np.equal(audio1.data, audio2.data[:, index]))
        assert not all(np.equal(audio1.data, audio2.data[:, 1 - index]))
        assert audio1.duration == audio2.duration
    with pytest.raises(ValueError):
        audio2.channel(2)


def test_isvalid(audio):
    assert audio.dtype is np.dtype(np.int16)
    assert audio.is_valid()
    # brutal cast from int16 to float32, still with values greater than 1
    audio2 = Audio(audio.data.astype(np.float32), audio.sample_rate, validate=False)
    assert audio2.dtype is np.dtype(np.float32)
    with pytest.warns(UserWarning):
        assert not audio2.is_valid()
    with pytest.raises(ValueError, match='invalid audio data'):
        with pytest.warns(UserWarning):
            Audio(audio.data.astype(np.float32), audio.sample_rate, validate=True)
    # smooth cast from int16 to float32
    audio3 = audio.astype(np.float32)
    assert audio3.dtype is np.dtype(np.float32)
    assert audio3.is_valid()
    data = audio3.data.copy()
    data[6] = 1.1
    with pytest.raises(ValueError, match='invalid audio data for type'):
        with pytest.warns(UserWarning):
            Audio(data, audio.sample_rate)
    with pytest.warns(UserWarning):
        assert not Audio(data, audio.sample_rate, validate=False).is_valid()
    audio5 = Audio(audio.data.astype(np.uint8), audio.sample_rate, validate=False)
    assert audio5.dtype is np.dtype(np.uint8)
    with pytest.warns(UserWarning):
        assert not audio5.is_valid()


@pytest.mark.parametrize('dtype', DTYPES)
def test_astype(audio, dtype):
    audio = Audio(audio.data[:10], audio.sample_rate)
    assert audio.dtype is np.dtype(np.int16)
    audio2 = audio.astype(dtype)
    assert audio2.dtype is np.dtype(dtype)
    assert audio.dtype is np.dtype(np.int16)
    assert audio2.is_valid()
    audio3 = audio2.astype(np.int16)
    assert audio3.data == pytest.approx(audio.data)
    assert audio3.dtype is np.dtype(np.int16)
    for dtype2 in set(DTYPES) - set([np.int16, dtype]):
        audio4 = audio2.astype(dtype2)
        assert audio4.is_valid()
        assert audio4.dtype is np.dtype(dtype2)
        assert audio4.astype(np.int16).data == pytest.approx(audio.data)


@pytest.mark.parametrize('dtype', [np.uint8, np.int64, np.longdouble, str, int])
def test_asbadtype(audio, dtype):
    with pytest.raises(ValueError):
        audio.astype(dtype)


@pytest.mark.parametrize('fs, backend', [
    (f, b) for f in [4000, 8000, 16000, 32000, 44100, 48000] for b in ('sox', 'scipy')])
def test_resample(audio, fs, backend):
    audio2 = audio.resample(fs, backend=backend)
    assert audio2.nchannels == audio.nchannels
    assert audio2.sample_rate == fs
    assert audio2.nsamples == pytest.approx(int(audio.nsamples * fs / audio.sample_rate), abs=1)
    assert audio2.data.mean() == pytest.approx(audio.data.mean(), abs=0.25)
    assert audio2.dtype == audio.dtype
    if fs >= audio.sample_rate:  # back to the original sample rate
        audio3 = audio2.resample(audio.sample_rate, backend=backend)
        assert audio3.nchannels == audio.nchannels
        assert audio3.sample_rate == audio.sample_rate
        assert audio3.dtype == audio.dtype


def test_resample_bad(audio):
    with pytest.raises(ValueError) as err:
        audio.resample(5, backend='a_bad_one')
    assert 'backend must be sox or scipy, it is' in str(err.value)
    with pytest.raises(ValueError) as err:
        audio.resample(0)
    assert 'resampling at 0 failed' in str(err.value)


def test_segment(audio):
    d = audio.duration
    assert audio.segment([(0., d)])[0] == audio
    assert audio.segment([(0., d + 10)])[0] == audio
    for parts in (2, 3):
        bounds = [(k * d / parts, (k + 1) * d / parts) for k in range(parts)]
        chunks = audio.segment(bounds)
        assert all(c.duration == pytest.approx(d / parts, rel=1e-3) for c in chunks)
        assert sum(c.nsamples for c in chunks) == audio.nsamples
        assert Audio(np.concatenate([c.data for c in chunks]), audio.sample_rate) == audio


def test_segment_bad(audio):
    with pytest.raises(ValueError, match='segments must be a list'):
        audio.segment(0)
    with pytest.raises(ValueError, match='must be pairs'):
        audio.segment([0, 1])
    with pytest.raises(ValueError, match='must be pairs'):
        audio.segment([(0, 1, 2)])
    with pytest.raises(ValueError, match='must be sorted'):
        audio.segment([(1, 0)])
