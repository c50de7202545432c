"""Frames: extract overlapping frames from raw (sampled) signals

Mirror of reference shennong/frames.py:42-249.  The frame count is Kaldi's NumFrames (through the
C ABI, snf_num_frames, replacing kaldi.feat.window.num_frames at frames.py:137); note that
``make_frames`` starts every frame at ``frame * shift`` and mirrors only the tail when
``snip_edges=False`` (frames.py:213-215), unlike Kaldi's centred framing used by the processors.
"""

import numpy as np

from shennong_amd import _abi, _backend
from shennong_amd.base import BaseProcessor


class Frames(BaseProcessor):
    """Extract frames from raw signals"""
    def __init__(self, sample_rate=16000,
                 frame_shift=0.01, frame_length=0.025,
                 snip_edges=True):
        self._options = _abi.default_frame_options()
        self.sample_rate = sample_rate
        self.frame_shift = frame_shift
        self.frame_length = frame_length
        self.snip_edges = snip_edges

    @property
    def name(self):
        return 'frames'

    @property
    def sample_rate(self):
        """Waveform sample frequency in Hertz"""
        return self._options.samp_freq

    @sample_rate.setter
    def sample_rate(self, value):
        self._options.samp_freq = value

    @property
    def frame_shift(self):
        """Frame shift in seconds"""
        return self._options.frame_shift_ms / 1000.0

    @frame_shift.setter
    def frame_shift(self, value):
        self._options.frame_shift_ms = value * 1000.0

    @property
    def frame_length(self):
        """Frame length in seconds"""
        return self._options.frame_length_ms / 1000.0

    @frame_length.setter
    def frame_length(self, value):
        self._options.frame_length_ms = value * 1000.0

    @property
    def snip_edges(self):
        """If true, output only frames that completely fit in the file"""
        return bool(self._options.snip_edges)

    @snip_edges.setter
    def snip_edges(self, value):
        self._options.snip_edges = bool(value)

    @property
    def samples_per_frame(self):
        """The number of samples in one frame"""
        return int(self.frame_length * self.sample_rate)

    @property
    def samples_per_shift(self):
        """The number of samples between two shifts"""
        return int(self.frame_shift * self.sample_rate)

    def nframes(self, nsamples):
        """Returns the number of frames extracted from `nsamples`"""
        if self.samples_per_shift == 0:
            raise ValueError('cannot compute nframes: sample rate too low')
        return _backend.num_frames(self._options, nsamples)

    def first_sample_of_frame(self, frame):
        """Returns the index of the first sample of frame indexed `frame`"""
        return int(frame * self.samples_per_shift)

    def last_sample_of_frame(self, frame):
        """Returns the index+1 of the last sample of frame indexed `frame`"""
        return int(self.first_sample_of_frame(frame) + self.samples_per_frame)

    def times(self, nsamples):
        """Returns an array of (tstart, tstop) times of each frames of a signal"""
        nframes = self.nframes(nsamples)
        return np.vstack((
            np.arange(nframes) * self.frame_shift,
            np.arange(nframes) * self.frame_shift + self.frame_length)).T

    def boundaries(self, nframes):
        """Returns an array of (istart, istop) index boundaries of frames"""
        first = [self.first_sample_of_frame(i) for i in range(nframes)]
        return (np.asarray(first, dtype=np.int64).repeat(2).reshape(nframes, 2)
                + (0, self.samples_per_frame)).astype(int)

    def make_frames(self, array, writeable=False):
        """Returns an `array` divided in frames, shape [nframes, samples_per_frame, ...]"""
        nframes = self.nframes(array.shape[0])
        if not self.snip_edges:
            # mirror the data in the last frames
            n = self.last_sample_of_frame(nframes-1) - array.shape[0]
            array = np.concatenate((array, array[-n-1:-1][::-1]))
        if writeable is True:
            return self._make_frames_by_copy(array, nframes)
        return self._make_frames_by_view(array, nframes)

    def _make_frames_by_view(self, array, nframes):
        shape = (nframes, self.samples_per_frame) + array.shape[1:]
        strides = (array.strides[0] * self.samples_per_shift,
                   array.strides[0]) + array.strides[1:]
        return np.lib.stride_tricks.as_strided(
            array, shape=shape, strides=strides, writeable=False)

    def _make_frames_by_copy(self, array, nframes):
        boundaries = self.boundaries(nframes)
        nsamples = self.samples_per_frame
        framed = np.empty(
            (nframes, nsamples) + array.shape[1:], dtype=array.dtype)
        for i, (start, stop) in enumerate(boundaries):
            assert stop - start == nsamples
            framed[i] = array[start:stop]
        return framed
