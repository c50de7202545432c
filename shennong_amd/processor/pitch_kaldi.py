"""Kaldi pitch: Audio ---> KaldiPitchProcessor ---> Features ---> KaldiPitchPostProcessor

Same parameters, defaults, outputs and error messages as reference
shennong/processor/pitch_kaldi.py:73-540; the tracker (NCCF batched-lag correlation + Viterbi per
utterance, csrc/kernels_pitch.hip) and the post-processing run on the HIP backend (plan kinds PITCH
and PITCH_POST).
"""

import numpy as np

from shennong_amd import _abi, _backend
from shennong_amd._options import F32, FLAG, SECONDS, Configurable, Option
from shennong_amd.features import Features
from shennong_amd.postprocessor.base import FeaturesPostProcessor
from shennong_amd.processor.base import FeaturesProcessor, batch_features
from shennong_amd.utils import copy_properties


class KaldiPitchProcessor(Configurable, FeaturesProcessor):
    """Extracts the (NCCF, pitch) per frame from a speech signal"""
    _kind = _abi.KIND_PITCH
    name = 'pitch'
    ndims = 2

    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, min_f0=50, max_f0=400,
                 soft_min_f0=10, penalty_factor=0.1,
                 lowpass_cutoff=1000, resample_freq=4000,
                 delta_pitch=0.005, nccf_ballast=7000,
                 lowpass_filter_width=1, upsample_filter_width=5):
        super().__init__()
        self._configure(locals())

    sample_rate = Option('pitch.samp_freq', 'Waveform sample frequency in Hertz')
    frame_shift = Option('pitch.frame_shift_ms', 'Frame shift in seconds', SECONDS)
    frame_length = Option('pitch.frame_length_ms', 'Frame length in seconds', SECONDS)
    min_f0 = Option('pitch.min_f0', 'Minimum F0 to search for in Hertz')
    max_f0 = Option('pitch.max_f0', 'Maximum F0 to search for in Hertz')
    soft_min_f0 = Option(
        'pitch.soft_min_f0', 'Minimum F0 to search, applied in soft way, in Hertz (<= min_f0)')
    penalty_factor = Option('pitch.penalty_factor', 'Cost factor for F0 change', F32)
    lowpass_cutoff = Option('pitch.lowpass_cutoff', 'Cutoff frequency for low-pass filter, in Hertz')
    resample_freq = Option(
        'pitch.resample_freq',
        'Frequency that we down-sample the signal to, more than twice `lowpass_cutoff`')
    delta_pitch = Option(
        'pitch.delta_pitch', 'Smallest relative change in pitch that the algorithm measures', F32)
    nccf_ballast = Option(
        'pitch.nccf_ballast', 'Increasing this factor reduces NCCF for quiet frames')
    lowpass_filter_width = Option(
        'pitch.lowpass_filter_width', 'Integer that determines filter width of lowpass filter')
    upsample_filter_width = Option(
        'pitch.upsample_filter_width', 'Integer that determines filter width when upsampling NCCF')

    @property
    def _options(self):
        """The tracker's part of the option record (what the oracle's pitch entry point takes)"""
        return self._record.pitch

    def times(self, nframes):
        """(start, stop) of every output row in seconds"""
        start = np.arange(nframes) * self.frame_shift
        return np.vstack((start, start + self.frame_length)).T

    def _check(self, signal):
        if signal.nchannels != 1:
            raise ValueError(
                'audio signal must have one channel, but it has {}'
                .format(signal.nchannels))
        if self.sample_rate != signal.sample_rate:
            raise ValueError(
                'processor and signal mismatch in sample rates: '
                '{} != {}'.format(self.sample_rate, signal.sample_rate))

    def process(self, signal):
        """Extracts the (NCCF, pitch) from a given speech `signal`"""
        return self._process_batch([signal])[0]

    def _process_pinned(self, corpus):
        """`_process_batch` over a page-locked corpus (Utterances.pin(), or WAV files read into staging memory)"""
        if self.sample_rate != corpus.sample_rate:
            raise ValueError(
                'processor and signal mismatch in sample rates: '
                '{} != {}'.format(self.sample_rate, corpus.sample_rate))
        return _backend.get_plan(self._build_options()).run_pinned(
            corpus, None, check_finite=True,
            wrap=lambda datas: batch_features(datas, self.times, lambda _: self.get_properties()))

    def _process_batch(self, signals):
        for signal in signals:
            self._check(signal)
        waves = [s.astype(np.int16).data for s in signals]  # 16 bits integers, like every processor
        datas = _backend.get_plan(self._build_options()).run(waves, check_finite=True)
        return batch_features(datas, self.times, lambda _: self.get_properties())


class KaldiPitchPostProcessor(Configurable, FeaturesPostProcessor):
    """Processes the raw (NCCF, pitch) computed by the KaldiPitchProcessor

    Output columns, in that order when enabled: POV feature, mean-subtracted log-pitch,
    delta of log-pitch, raw log-pitch.
    """
    _kind = _abi.KIND_PITCH_POST
    name = 'pitch postprocessing'

    def __init__(self, pitch_scale=2.0, pov_scale=2.0, pov_offset=0.0,
                 delta_pitch_scale=10.0, delta_pitch_noise_stddev=0.005,
                 normalization_left_context=75,
                 normalization_right_context=75,
                 delta_window=2, delay=0,
                 add_pov_feature=True, add_normalized_log_pitch=True,
                 add_delta_pitch=True, add_raw_log_pitch=False):
        super().__init__()
        self._configure(locals())

    pitch_scale = Option(
        'pitch_post.pitch_scale', 'Scaling factor for the final normalized log-pitch value')
    pov_scale = Option(
        'pitch_post.pov_scale', 'Scaling factor for final probability of voicing feature')
    pov_offset = Option(
        'pitch_post.pov_offset', 'This can be used to add an offset to the POV feature')
    delta_pitch_scale = Option(
        'pitch_post.delta_pitch_scale', 'Term to scale the final delta log-pitch feature')
    delta_pitch_noise_stddev = Option(
        'pitch_post.delta_pitch_noise_stddev',
        'Standard deviation for noise we add to the delta log-pitch', F32)
    normalization_left_context = Option(
        'pitch_post.normalization_left_context',
        'Left-context (in frames) for moving window normalization')
    normalization_right_context = Option(
        'pitch_post.normalization_right_context',
        'Right-context (in frames) for moving window normalization')
    delta_window = Option(
        'pitch_post.delta_window', 'Number of frames on each side of central frame')
    delay = Option(
        'pitch_post.delay', 'Number of frames by which the pitch information is delayed')
    add_pov_feature = Option(
        'pitch_post.add_pov_feature', 'If true, the warped NCCF is added to output features', FLAG)
    add_normalized_log_pitch = Option(
        'pitch_post.add_normalized_log_pitch',
        'If true, the normalized log-pitch is added to output features', FLAG)
    add_delta_pitch = Option(
        'pitch_post.add_delta_pitch',
        'If true, time derivative of log-pitch is added to output features', FLAG)
    add_raw_log_pitch = Option(
        'pitch_post.add_raw_log_pitch', 'If true, log-pitch is added to output features', FLAG)

    _FLAGS = ('add_pov_feature', 'add_normalized_log_pitch', 'add_delta_pitch', 'add_raw_log_pitch')

    @property
    def _options(self):
        """The post-processing part of the option record"""
        return self._record.pitch_post

    @property
    def ndims(self):
        return sum(getattr(self, flag) for flag in self._FLAGS)

    def get_properties(self, features):
        """The raw pitch's properties with this processor's parameters next to the tracker's and
        the new column range (reference pitch_kaldi.py:491-495)"""
        properties = copy_properties(features.properties)
        properties['pitch'][self.name] = self.get_params()
        properties['pipeline'][0]['columns'] = [0, self.ndims - 1]
        return properties

    def _check(self, raw_pitch):
        if not any(getattr(self, flag) for flag in self._FLAGS):
            raise ValueError(
                'at least one of the following options must be True: ' + ', '.join(self._FLAGS))
        if raw_pitch.shape[1] != 2:
            raise ValueError(
                'data shape must be (_, 2), but it is (_, {})'
                .format(raw_pitch.shape[1]))

    def process(self, raw_pitch):
        """Post process a raw pitch data as specified by the options"""
        return self._process_batch([raw_pitch])[0]

    def _process_batch(self, raw_pitches):
        for raw in raw_pitches:
            self._check(raw)
        datas = _backend.get_plan(self._build_options()).run_post(
            [raw.data for raw in raw_pitches], check_finite=True)
        return [Features(d, raw.times, properties=self.get_properties(raw), validate=False)
                for d, raw in zip(datas, raw_pitches)]
