"""Native WAV ingest (include/shennong_amd.h snf_wav_scan / snf_wav_read_pcm16) against scipy's
reader - the one `Audio.load` uses, as the reference does (shennong/audio.py:243-286) - and the interval arithmetic of
`Utterance.load_audio` (utterances.py + Audio.segment).  Host only; the GPU tests run the same files through
Utterances.pin() and the pipeline."""
import ctypes as C
import os
import struct

import numpy as np
import pytest
import scipy.io.wavfile

from conftest import GOLDEN
from shennong_amd import Audio, Utterances, _backend
from shennong_amd.audio import load_int16_block, sample_range

WAV = os.path.join(GOLDEN, 'test.wav')
WAV_8K = os.path.join(GOLDEN, 'test.8k.wav')


def _scan(path):
    out = [C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32(), C.c_int32()]
    rc = _backend.lib().snf_wav_scan(os.fsencode(path), *[C.byref(x) for x in out])
    return rc, tuple(x.value for x in out)


def test_scan_matches_the_python_scan(tmp_path):
    for path in (WAV, WAV_8K):
        meta = Audio.scan(path)
        rc, (channels, rate, nsamples, bits, tag) = _scan(path)
        assert rc == 0 and (channels, rate, nsamples) == (meta.nchannels, meta.sample_rate, meta.nsamples)
        assert bits == 16 and tag == 1
    data = scipy.io.wavfile.read(WAV)[1]
    scipy.io.wavfile.write(tmp_path / 'f32.wav', 16000, (data / 2 ** 15).astype(np.float32))
    scipy.io.wavfile.write(tmp_path / 'stereo.wav', 16000, np.stack([data, data], axis=1))
    assert _scan(str(tmp_path / 'f32.wav')) == (0, (1, 16000, data.shape[0], 32, 3))
    assert _scan(str(tmp_path / 'stereo.wav')) == (0, (2, 16000, data.shape[0], 16, 1))
    (tmp_path / 'bad.wav').write_bytes(b'not a wave file at all')
    assert _scan(str(tmp_path / 'bad.wav'))[0] == -1 and _scan(str(tmp_path / 'missing.wav'))[0] == -1
    assert b'not a RIFF' in _backend.lib().snf_last_error() or b'not found' in _backend.lib().snf_last_error()


def _riff(chunks):
    body = b'WAVE' + b''.join(name + struct.pack('<I', len(data)) + data + (b'\0' if len(data) & 1 else b'')
                              for name, data in chunks)
    return b'RIFF' + struct.pack('<I', len(body)) + body


def test_read_matches_scipy(tmp_path):
    data = scipy.io.wavfile.read(WAV)[1]
    data8 = scipy.io.wavfile.read(WAV_8K)[1]
    # a LIST chunk of odd length in front of the data, the extensible form of the format chunk, a data chunk that
    # claims more than the file holds (streamed files): what recording tools write
    samples = np.arange(-500, 500, dtype='<i2')
    fmt = struct.pack('<HHIIHH', 1, 1, 16000, 32000, 2, 16)
    ext = (struct.pack('<HHIIHHHHI', 0xFFFE, 1, 16000, 32000, 2, 16, 22, 16, 4) + struct.pack('<H', 1)
           + b'\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71')   # KSDATAFORMAT_SUBTYPE_PCM
    (tmp_path / 'list.wav').write_bytes(_riff([(b'fmt ', fmt), (b'LIST', b'INFOabc'), (b'data', samples.tobytes())]))
    (tmp_path / 'ext.wav').write_bytes(_riff([(b'fmt ', ext), (b'data', samples.tobytes())]))
    streamed = _riff([(b'fmt ', fmt), (b'data', samples.tobytes())])
    where = streamed.index(b'data') + 4
    (tmp_path / 'streamed.wav').write_bytes(streamed[:where] + struct.pack('<I', 0xFFFFFFFF) + streamed[where + 4:])
    for name in ('list.wav', 'ext.wav', 'streamed.wav'):
        rate, want = scipy.io.wavfile.read(tmp_path / name)
        assert rate == 16000 and np.array_equal(want, samples)
    paths = [WAV, WAV_8K, WAV] + [str(tmp_path / n) for n in ('list.wav', 'ext.wav', 'streamed.wav')]
    first = [0, 10, data.shape[0] - 5, 0, 3, 990]
    count = [data.shape[0], 1000, 5, 1000, 17, 10]
    offsets = np.concatenate([[0], np.cumsum(count)])
    dst = np.full(offsets[-1] + 4, 77, dtype=np.int16)
    for threads in (1, 4):
        dst[:] = 77
        status = _backend.read_wav_pcm16(paths, first, count, dst, offsets[:-1], threads=threads)
        assert status.tolist() == [0] * 6
        want = [data, data8[10:1010], data[-5:], samples, samples[3:20], samples[990:]]
        assert np.array_equal(dst[:offsets[-1]], np.concatenate(want)) and (dst[offsets[-1]:] == 77).all()
    # what the native reader leaves to the general one, and what it reports
    scipy.io.wavfile.write(tmp_path / 'f32.wav', 16000, (data / 2 ** 15).astype(np.float32))
    scipy.io.wavfile.write(tmp_path / 'i32.wav', 16000, data.astype(np.int32) << 16)
    status = _backend.read_wav_pcm16([str(tmp_path / 'f32.wav'), str(tmp_path / 'i32.wav'), str(tmp_path / 'no.wav'), WAV],
                                     [0, 0, 0, 0], [4, 4, 4, data.shape[0] + 1], dst, [0, 4, 8, 12])
    assert status.tolist() == [1, 1, 2, 3]
    assert _backend.read_wav_pcm16([], [], [], dst, []).shape == (0,)


def test_sample_range_is_what_load_audio_cuts():
    audio = Audio.load(WAV)
    n, rate = audio.nsamples, audio.sample_rate
    for tstart, tstop in ((None, None), (0, 1), (0.5, 1.4), (0.1234, 0.98765), (1.0, 1.2), (0, 1.4195)):
        with_interval = () if tstart is None else (tstart, tstop)
        utt = Utterances([('u', WAV) + with_interval])['u']
        first, count = sample_range(n, rate, utt.tstart, utt.tstop)
        assert np.array_equal(utt.load_audio().data, audio.data[first:first + count]), (tstart, tstop)
    assert sample_range(100, 16000, 0.0, 1.0) == (0, 100) and sample_range(100, 16000, 1.0, 2.0) == (100, 0)


def test_load_int16_block_mixes_native_and_general_readers(tmp_path):
    data = scipy.io.wavfile.read(WAV)[1]
    scipy.io.wavfile.write(tmp_path / 'f32.wav', 16000, (data / 2 ** 15).astype(np.float32))
    scipy.io.wavfile.write(tmp_path / 'i32.wav', 16000, data.astype(np.int32) << 15)
    index = Utterances([('a', WAV, 0.2, 0.9), ('b', str(tmp_path / 'f32.wav'), 0, 1.4), ('c', str(tmp_path / 'i32.wav'), 0.1, 0.3),
                        ('d', WAV, 0, 1.0)])
    utts = list(index)
    metas = [Audio.scan(u.audio_file) for u in utts]
    lengths = [sample_range(m.nsamples, m.sample_rate, u.tstart, u.tstop)[1] for u, m in zip(utts, metas)]
    soff = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    block = np.zeros(soff[-1], dtype=np.int16)
    counts = load_int16_block(utts, metas, block, soff)
    assert counts.tolist() == lengths
    for k, utt in enumerate(utts):
        assert np.array_equal(block[soff[k]:soff[k + 1]], utt.load_audio().astype(np.int16).data), utt.name
    (tmp_path / 'bad.wav').write_bytes(_riff([(b'fmt ', struct.pack('<HHIIHH', 1, 1, 16000, 32000, 2, 16)),
                                              (b'data', bytes(40))]))
    short = Utterances([('x', str(tmp_path / 'bad.wav'))])
    meta = Audio.scan(str(tmp_path / 'bad.wav'))
    with pytest.raises(ValueError, match='fewer samples'):
        load_int16_block(list(short), [meta._replace(nsamples=meta.nsamples + 5)], np.zeros(64, np.int16),
                         np.array([0, meta.nsamples + 5]))
