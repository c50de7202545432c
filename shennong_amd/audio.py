"""Audio container: the input of every features processor

Mirror of reference shennong/audio.py for WAV files: ``data / sample_rate / nchannels / nsamples /
dtype``, ``astype`` (int16 <-> float scaling by 2**15, audio.py:469-518), ``segment``
(audio.py:520-561), ``channel`` (audio.py:328-357), ``load`` / ``save`` / ``scan`` through scipy
(audio.py:179-320; the reference falls back to pydub / ffmpeg and sox for flac, mp3, ...: neither
exists offline, those formats raise a ValueError that says so) and ``resample`` with the reference's
scipy backend (audio.py:358-425; the sox backend is the same call here).
"""

import collections
import functools
import os
import warnings

import numpy as np
import scipy.io.wavfile
import scipy.signal


class Audio:
    """An audio signal with the given `data` and `sample_rate`"""
    def __init__(self, data, sample_rate, validate=True):
        self._sample_rate = int(sample_rate)
        # force shape (n, 1) to be (n,)
        self._data = (
            data[:, 0] if data.ndim > 1 and data.shape[1] == 1 else data)
        if validate and not self.is_valid():
            raise ValueError(f'invalid audio data for type {self.dtype}')

    def __eq__(self, other):
        if self.sample_rate != other.sample_rate:
            return False
        return np.array_equal(self.data, other.data)

    @property
    def data(self):
        return self._data

    @property
    def sample_rate(self):
        return self._sample_rate

    @property
    def duration(self):
        return self.nsamples / self.sample_rate

    @property
    def nchannels(self):
        if self.data.ndim == 1:
            return 1
        return self.data.shape[1]

    @property
    def nsamples(self):
        return self.data.shape[0]

    @property
    def shape(self):
        return self.data.shape

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def precision(self):
        return self.dtype.itemsize * 8

    # the pipeline loads a file once for its metadata and again for every segment of it: like the
    # reference (audio.py:240-243) keep the last two decoded files
    @classmethod
    @functools.lru_cache(maxsize=2)
    def load(cls, filename):
        """Creates an `Audio` instance from a WAV file (16/32 bits PCM or float)

        Raises ValueError if `filename` does not exist or is not a WAV file."""
        filename = str(filename)
        if not os.path.isfile(filename):
            raise ValueError(f'{filename}: file not found')
        try:
            sample_rate, data = scipy.io.wavfile.read(filename)
        except Exception as err:  # noqa
            raise ValueError(
                f'{filename}: cannot read file, Decoding failed ({err}); only WAV files are '
                f'supported here (the reference decodes other formats with pydub/ffmpeg)') from None
        return cls(data, sample_rate, validate=False)

    def save(self, filename):
        """Saves the audio data to a WAV `filename`

        Raises ValueError if the file already exists, has no extension or is not ``.wav``."""
        filename = str(filename)
        if os.path.isfile(filename):
            raise ValueError(f'{filename}: file already exists')
        if '.' not in filename:
            raise ValueError(
                f'{filename}: cannot write audio file without extension')
        extension = filename.split('.')[-1]
        if extension.lower() != 'wav':
            raise ValueError(
                f'{filename}: cannot write file, only WAV files are supported here '
                f'(the reference encodes other formats with pydub/ffmpeg)')
        try:
            scipy.io.wavfile.write(filename, self.sample_rate, self.data)
        except ValueError as err:  # pragma: nocover
            raise ValueError(f'{filename}: cannot write file, {err}') from None

    def channel(self, index):
        """Builds a mono signal from channel `index` of a multi-channel one"""
        if index == 0 and self.nchannels == 1:
            return self
        if index >= self.nchannels:
            raise ValueError(
                f'not enough channels ({self.nchannels}) to extract '
                f'the index {index} (indices count starts at 0)')
        return Audio(self.data[:, index], self.sample_rate)

    def resample(self, sample_rate, backend='sox'):
        """Returns the audio signal resampled at the given `sample_rate`

        `backend` must be 'sox' or 'scipy' like in the reference; sox is not available here, so both
        run the reference's scipy backend (Fourier-domain `scipy.signal.resample`)."""
        if backend not in ('sox', 'scipy'):
            raise ValueError(f'backend must be sox or scipy, it is {backend}')
        if sample_rate == self.sample_rate:
            return self
        try:
            nsamples = int(self.nsamples * sample_rate / self.sample_rate)
            if nsamples <= 0:
                raise ValueError('no sample left')
            with warnings.catch_warnings():
                warnings.simplefilter('ignore', category=FutureWarning)
                data = scipy.signal.resample(self.data, nsamples)
        except (ValueError, TypeError, ZeroDivisionError):
            raise ValueError(f'resampling at {sample_rate} failed!') from None
        # resampling casts to float64: back to the original dtype
        return Audio(data.astype(self.dtype), sample_rate, validate=False)

    _metadata = collections.namedtuple(
        '_metadata', 'nchannels sample_rate nsamples duration')

    @classmethod
    def scan(cls, filename):
        """Returns (nchannels, sample_rate, nsamples, duration) without decoding the samples
        (reference audio.py:179-240 asks sox; wav headers are read with scipy here).  An
        in-memory :class:`Audio` is accepted too (benchmark / tests)."""
        if isinstance(filename, Audio):
            return cls._metadata(filename.nchannels, filename.sample_rate,
                                 filename.nsamples, filename.duration)
        filename = str(filename)
        if not os.path.isfile(filename):
            raise ValueError(f'{filename}: file not found')
        try:
            sample_rate, data = scipy.io.wavfile.read(filename, mmap=True)
        except Exception as err:  # noqa
            raise ValueError(f'{filename}: cannot scan audio file: {err}') from None
        nchannels = 1 if data.ndim == 1 else data.shape[1]
        return cls._metadata(nchannels, sample_rate, data.shape[0],
                             data.shape[0] / sample_rate)

    @staticmethod
    def _is_valid_dtype(dtype):
        return dtype in (np.dtype(t) for t in (
            np.int16, np.int32, np.float32, np.float64))

    def is_valid(self):
        """True if dtype is supported and samples are within the type's range"""
        if not self._is_valid_dtype(self.dtype):
            warnings.warn(f'unsupported audio data type: {self.dtype}')
            return False
        if self.dtype is np.dtype(np.int16):
            emin, emax = -2**15, 2**15 - 1
        elif self.dtype is np.dtype(np.int32):
            emin, emax = -2**31, 2**31 - 1
        else:
            emin, emax = -1, 1
        if self.data.size == 0:
            return True
        dmin, dmax = np.amin(self.data), np.amax(self.data)
        if dmin < emin or dmax > emax:
            warnings.warn(
                f'invalid audio for type {self.dtype}: boundaries must be in '
                f'({emin}, {emax}) but are ({dmin}, {dmax})')
            return False
        return True

    def astype(self, dtype):
        """Returns the signal converted to `dtype` (reference audio.py:469-518)"""
        if self.dtype is np.dtype(dtype):
            return self
        if not self._is_valid_dtype(dtype):
            raise ValueError(f'unsupported audio data type: {dtype}')
        if self.dtype is np.dtype(np.int16):
            if dtype is np.int32:
                # `data * 2**15` under the reference's numpy 1.x value-based casting
                data = self.data.astype(np.int32) * 2**15
            else:
                data = self.data / 2**15
        elif self.dtype is np.dtype(np.int32):
            if dtype is np.int16:
                data = self.data / 2**15
            else:
                data = self.data / 2**30
        else:
            if dtype is np.int16:
                data = self.data * 2**15
            elif dtype is np.int32:
                data = self.data * 2**30
            else:
                data = self.data
        with np.errstate(invalid='ignore'):
            # out-of-range floats wrap exactly like the reference's C cast
            return Audio(data.astype(dtype), self.sample_rate, validate=False)

    def segment(self, segments):
        """Returns audio chunks for a list of (tstart, tstop) pairs in seconds"""
        if not isinstance(segments, list):
            raise ValueError('segments must be a list')
        for segment in segments:
            try:
                if not len(segment) == 2:
                    raise ValueError('segments elements must be pairs')
            except TypeError:
                raise ValueError('segments elements must be pairs')
            if segment[0] >= segment[1]:
                raise ValueError('time indices in segments must be sorted')
        chunks = []
        for segment in segments:
            istart = int(segment[0] * self.sample_rate)
            istop = int(segment[1] * self.sample_rate)
            chunks.append(Audio(
                self.data[istart:istop], self.sample_rate, validate=False))
        return chunks
