"""Frames: extract overlapping frames from raw (sampled) signals

Counterpart of reference shennong/frames.py:42-249.  The frame count is Kaldi's NumFrames (through
the C ABI, snf_num_frames, replacing kaldi.feat.window.num_frames at frames.py:137).  Note the
reference's own convention, kept here: ``make_frames`` starts every frame at ``frame * shift`` and
mirrors only the tail when ``snip_edges=False`` (frames.py:213-215), unlike Kaldi's centred framing
used by the processors.
"""

import numpy as np

from shennong_amd import _abi, _backend
from shennong_amd._options import FLAG, SECONDS, Configurable, Option
from shennong_amd.base import BaseProcessor


class Frames(Configurable, BaseProcessor):
    """Extract frames from raw signals"""
    _kind = _abi.KIND_SPECTROGRAM  # (only the framing part of the option record is used)
    name = 'frames'

    def __init__(self, sample_rate=16000,
                 frame_shift=0.01, frame_length=0.025,
                 snip_edges=True):
        self._configure(locals())

    sample_rate = Option('frame.samp_freq', 'Waveform sample frequency in Hertz')
    frame_shift = Option('frame.frame_shift_ms', 'Frame shift in seconds', SECONDS)
    frame_length = Option('frame.frame_length_ms', 'Frame length in seconds', SECONDS)
    snip_edges = Option(
        'frame.snip_edges', 'If true, output only frames that completely fit in the file', FLAG)

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
        return _backend.num_frames(self._record.frame, nsamples)

    def first_sample_of_frame(self, frame):
        """Returns the index of the first sample of frame indexed `frame`"""
        return int(frame * self.samples_per_shift)

    def last_sample_of_frame(self, frame):
        """Returns the index+1 of the last sample of frame indexed `frame`"""
        return self.first_sample_of_frame(frame) + self.samples_per_frame

    def times(self, nsamples):
        """Returns an array of (tstart, tstop) times of each frames of a signal"""
        start = np.arange(self.nframes(nsamples)) * self.frame_shift
        return np.vstack((start, start + self.frame_length)).T

    def boundaries(self, nframes):
        """Returns an array [nframes, 2] of (istart, istop) sample indices of the frames"""
        first = np.arange(nframes, dtype=np.int64) * self.samples_per_shift
        return np.stack((first, first + self.samples_per_frame), axis=1).astype(int)

    def make_frames(self, array, writeable=False):
        """Returns `array` divided in frames, shape [nframes, samples_per_frame, ...]: a read-only
        strided view, or a copy when `writeable`"""
        nframes = self.nframes(array.shape[0])
        if not self.snip_edges:
            # the last frames reach beyond the data: mirror its tail
            missing = self.last_sample_of_frame(nframes - 1) - array.shape[0]
            array = np.concatenate((array, array[-missing - 1:-1][::-1]))
        step, width = self.samples_per_shift, self.samples_per_frame
        view = np.lib.stride_tricks.as_strided(
            array, shape=(nframes, width) + array.shape[1:],
            strides=(array.strides[0] * step, array.strides[0]) + array.strides[1:],
            writeable=False)
        return view.copy() if writeable is True else view
