"""Audio container: the input of every features processor

Counterpart of reference shennong/audio.py for WAV files.  An :class:`Audio` is a numpy array of
samples (one column per channel) and a sample rate; the sample type is one of int16, int32, float32
or float64 and each type has a full scale (2**15, 2**30 and 1.0): ``astype`` rescales between them
(audio.py:469-518), the processors ask for int16.  WAV files are read with scipy (audio.py:179-320;
the reference falls back to pydub / ffmpeg and to sox for flac, mp3, ...: neither exists offline,
those formats raise a ValueError that says so).  `save` writes WAV files and `resample` is the
reference's scipy backend (Fourier-domain resampling; its sox backend needs the external binary) -
neither is on the features path, both are what the reference's callers use (pipeline_manager.py:240).
"""

import collections
import functools
import os
import warnings

import numpy as np
import scipy.io.wavfile

# sample type -> (log2 of the full scale, smallest and largest valid sample)
_FORMATS = {
    np.dtype(np.int16): (15, -2**15, 2**15 - 1),
    np.dtype(np.int32): (30, -2**31, 2**31 - 1),
    np.dtype(np.float32): (0, -1, 1),
    np.dtype(np.float64): (0, -1, 1),
}
_Metadata = collections.namedtuple('_metadata', 'nchannels sample_rate nsamples duration')


def _read_wav(filename, what, **kwargs):
    filename = str(filename)
    if not os.path.isfile(filename):
        raise ValueError(f'{filename}: file not found')
    try:
        return scipy.io.wavfile.read(filename, **kwargs)
    except Exception as err:  # noqa (scipy raises ValueError and others on non-WAV content)
        raise ValueError(
            f'{filename}: {what} ({err}); only WAV files are supported here (the reference '
            f'decodes other formats with pydub/ffmpeg)') from None


def sample_range(nsamples, sample_rate, tstart, tstop):
    """(first sample, number of samples) of the (tstart, tstop) interval of a signal, in seconds - what
    ``Utterance.load_audio`` cuts (`Audio.segment`: int(t * rate) on both ends, a Python slice: clipped to the
    signal); the whole signal when neither bound is set"""
    if not (tstart or tstop):
        return 0, int(nsamples)
    start = min(max(int(tstart * sample_rate), 0), nsamples) if tstart is not None else 0
    stop = min(max(int(tstop * sample_rate), 0), nsamples) if tstop is not None else nsamples
    return start, max(stop - start, 0)


def load_int16_block(utterances, metadata, block, offsets, load=None):
    """The int16 samples of file-based `utterances` (path, tstart, tstop; `metadata`: their header scans) written
    to `block` at `offsets`: 16-bit mono PCM files by the native reader, side by side (snf_wav_read_pcm16), every
    other sample type through ``load(utterance) -> Audio`` and `Audio.astype` (reference audio.py:469-518)"""
    from shennong_amd import _backend
    first, count = [], []
    for utt, meta in zip(utterances, metadata):
        a, c = sample_range(meta.nsamples, meta.sample_rate, utt.tstart, utt.tstop)
        first.append(a)
        count.append(c)
    status = _backend.read_wav_pcm16([u.audio_file for u in utterances], first, count, block, offsets[:-1])
    for k in np.flatnonzero(status).tolist():
        utt = utterances[k]
        if status[k] != 1:
            raise ValueError(f'{utt.audio_file}: cannot read file' + (
                ', it holds fewer samples than its header says' if status[k] == 3 else ''))
        signal = (load or (lambda u: u.load_audio()))(utt)
        if signal.nchannels != 1:
            raise ValueError('signal must have one dimension, but it has {} ({})'.format(signal.nchannels, utt.name))
        data = signal.astype(np.int16).data
        if data.shape[0] != count[k]:   # pragma: nocover (the header scan and the decoder disagree)
            raise ValueError(f'{utt.audio_file}: {data.shape[0]} samples decoded, {count[k]} expected')
        block[offsets[k]:offsets[k + 1]] = data
    return np.asarray(count, dtype=np.int64)


def _scan_native(filename):
    """Header scan by the library (snf_wav_scan: 2 us per file against 15 for a memory-mapped scipy read) for the
    sample types it knows to describe exactly as scipy does - PCM of 16 or 32 bits, IEEE float of 32 or 64;
    None for everything else (and when the library has not been built): the caller's scipy path decides"""
    import ctypes as C
    try:
        from shennong_amd import _backend
        lib = _backend.lib()
    except (RuntimeError, OSError):
        return None
    channels, rate, nsamples, bits, tag = C.c_int32(), C.c_int32(), C.c_int64(), C.c_int32(), C.c_int32()
    if lib.snf_wav_scan(os.fsencode(str(filename)), C.byref(channels), C.byref(rate), C.byref(nsamples),
                        C.byref(bits), C.byref(tag)) != 0:
        return None
    if (tag.value, bits.value) not in ((1, 16), (1, 32), (3, 32), (3, 64)) or rate.value <= 0:
        return None
    return _Metadata(channels.value, rate.value, nsamples.value, nsamples.value / rate.value)


class Audio:
    """An audio signal with the given `data` and `sample_rate`"""
    def __init__(self, data, sample_rate, validate=True):
        # a single channel given as a column is a vector (the kernels take vectors)
        self._data = data[:, 0] if data.ndim == 2 and data.shape[1] == 1 else data
        self._sample_rate = int(sample_rate)
        if validate and not self.is_valid():
            raise ValueError(f'invalid audio data for type {self.dtype}')

    data = property(lambda self: self._data, doc='The samples, [nsamples] or [nsamples, nchannels]')
    sample_rate = property(lambda self: self._sample_rate, doc='Samples per second')
    shape = property(lambda self: self._data.shape)
    dtype = property(lambda self: self._data.dtype)
    nsamples = property(lambda self: self._data.shape[0])
    nchannels = property(lambda self: 1 if self._data.ndim == 1 else self._data.shape[1])
    duration = property(lambda self: self.nsamples / self.sample_rate, doc='Duration in seconds')
    precision = property(lambda self: self.dtype.itemsize * 8, doc='Bits per sample')

    def __eq__(self, other):
        return self.sample_rate == other.sample_rate and np.array_equal(self.data, other.data)

    # ---- files ------------------------------------------------------------------------------------
    # the pipeline reads a file once for its metadata and again for every segment of it: like the
    # reference (audio.py:240-243) keep the last two decoded files
    @classmethod
    @functools.lru_cache(maxsize=2)
    def load(cls, filename):
        """Creates an `Audio` instance from a WAV file (16/32 bits PCM or float)

        Raises ValueError if `filename` does not exist or is not a WAV file."""
        sample_rate, data = _read_wav(filename, 'cannot read file, Decoding failed')
        return cls(data, sample_rate, validate=False)

    @classmethod
    def scan(cls, filename):
        """Returns (nchannels, sample_rate, nsamples, duration) without decoding the samples
        (reference audio.py:179-240 asks sox; the WAV header is read through a memory map here).
        An in-memory :class:`Audio` is accepted too (benchmark / tests)."""
        if isinstance(filename, Audio):
            audio = filename
            return _Metadata(audio.nchannels, audio.sample_rate, audio.nsamples, audio.duration)
        native = _scan_native(filename)
        if native is not None:
            return native
        sample_rate, data = _read_wav(filename, 'cannot scan audio file', mmap=True)
        return _Metadata(1 if data.ndim == 1 else data.shape[1], sample_rate, data.shape[0],
                         data.shape[0] / sample_rate)

    def save(self, filename):
        """Writes the signal to the WAV file `filename` (reference audio.py:288-320; the other
        containers it writes through pydub / ffmpeg are refused here)

        Raises ValueError if the file exists, has no extension or is not ``.wav``."""
        filename = str(filename)
        if os.path.isfile(filename):
            raise ValueError(f'{filename}: file already exists')
        if '.' not in os.path.basename(filename):
            raise ValueError(f'{filename}: cannot write audio file without extension')
        if not filename.lower().endswith('.wav'):
            raise ValueError(f'{filename}: only WAV files can be written here (the reference encodes other '
                             f'formats with pydub/ffmpeg)')
        try:
            scipy.io.wavfile.write(filename, self.sample_rate, self.data)
        except (ValueError, OSError) as err:
            raise ValueError(f'{filename}: cannot write file, {err}') from None

    def resample(self, sample_rate, backend='scipy'):
        """The signal resampled to `sample_rate`, in its own sample type (reference audio.py:358-424).
        `backend` 'scipy' is scipy.signal.resample (Fourier method: exact for band-limited signals, slow
        for long ones); 'sox' - the reference's default - needs the sox binary and is refused here."""
        if backend not in ('sox', 'scipy'):
            raise ValueError(f'backend must be sox or scipy, it is {backend}')
        if backend == 'sox':
            raise ValueError('the sox backend needs the sox binary, which is not available here: '
                             'use backend="scipy"')
        if not isinstance(sample_rate, (int, np.integer)) or sample_rate <= 0:
            raise ValueError(f'resampling at {sample_rate} failed!')
        if sample_rate == self.sample_rate:
            return self
        import scipy.signal
        nsamples = int(self.nsamples * sample_rate / self.sample_rate)
        data = scipy.signal.resample(self.data, nsamples)
        if np.issubdtype(self.dtype, np.integer):
            lo, hi = _FORMATS[self.dtype][1:]
            data = np.clip(np.rint(data), lo, hi)  # (overshoot of the sinc interpolation near full scale)
        return Audio(data.astype(self.dtype), int(sample_rate), validate=False)

    # ---- views and conversions --------------------------------------------------------------------
    def channel(self, index):
        """Builds a mono signal from channel `index` of a multi-channel one"""
        if index >= self.nchannels:
            raise ValueError(
                f'not enough channels ({self.nchannels}) to extract '
                f'the index {index} (indices count starts at 0)')
        return self if self.nchannels == 1 else Audio(self.data[:, index], self.sample_rate)

    @staticmethod
    def _is_valid_dtype(dtype):
        try:
            return np.dtype(dtype) in _FORMATS
        except TypeError:
            return False

    def is_valid(self):
        """True if the sample type is supported and the samples lie within its full scale (warns
        and returns False otherwise)"""
        if self.dtype not in _FORMATS:
            warnings.warn(f'unsupported audio data type: {self.dtype}')
            return False
        if self.data.size == 0:
            return True
        _, lowest, highest = _FORMATS[self.dtype]
        dmin, dmax = self.data.min(), self.data.max()
        if dmin < lowest or dmax > highest:
            warnings.warn(
                f'invalid audio for type {self.dtype}: boundaries must be in '
                f'({lowest}, {highest}) but are ({dmin}, {dmax})')
            return False
        return True

    def astype(self, dtype):
        """Returns the signal converted to `dtype`, rescaled from the full scale of the current
        sample type to the full scale of the new one (reference audio.py:469-518)"""
        if not self._is_valid_dtype(dtype):
            raise ValueError(f'unsupported audio data type: {dtype}')
        dtype = np.dtype(dtype)
        if dtype == self.dtype:
            return self
        shift = _FORMATS[dtype][0] - _FORMATS[self.dtype][0]
        if shift >= 0 and self.dtype.kind == 'i' and dtype.kind == 'i':
            data = self.data.astype(dtype) << shift  # integer up-scaling is exact
        else:
            # float arithmetic with a power of two, then the C cast of the reference: truncation
            # toward zero, out-of-range values wrap (no clipping)
            data = self.data * 2.0 ** shift
        with np.errstate(invalid='ignore'):
            return Audio(data.astype(dtype), self.sample_rate, validate=False)

    def segment(self, segments):
        """Returns one :class:`Audio` per (tstart, tstop) pair of `segments`, in seconds"""
        if not isinstance(segments, list):
            raise ValueError('segments must be a list')
        bounds = []
        for pair in segments:
            if not hasattr(pair, '__len__') or len(pair) != 2:
                raise ValueError('segments elements must be pairs')
            if not pair[0] < pair[1]:
                raise ValueError('time indices in segments must be sorted')
            bounds.append((int(pair[0] * self.sample_rate), int(pair[1] * self.sample_rate)))
        return [Audio(self.data[start:stop], self.sample_rate, validate=False)
                for start, stop in bounds]
