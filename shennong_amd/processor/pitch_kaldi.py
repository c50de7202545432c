"""Kaldi pitch: Audio ---> KaldiPitchProcessor ---> Features ---> KaldiPitchPostProcessor

Mirror of reference shennong/processor/pitch_kaldi.py:73-540 over the HIP backend
(NCCF batched-lag correlation + Viterbi per utterance; see csrc/kernels_pitch.hip).
"""

import copy

import numpy as np

from shennong_amd import _abi, _backend
from shennong_amd.features import Features
from shennong_amd.processor.base import FeaturesProcessor, batch_features
from shennong_amd.postprocessor.base import FeaturesPostProcessor


class KaldiPitchProcessor(FeaturesProcessor):
    """Extracts the (NCCF, pitch) per frame from a speech signal"""
    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, min_f0=50, max_f0=400,
                 soft_min_f0=10, penalty_factor=0.1,
                 lowpass_cutoff=1000, resample_freq=4000,
                 delta_pitch=0.005, nccf_ballast=7000,
                 lowpass_filter_width=1, upsample_filter_width=5):
        super().__init__()
        self._options = _abi.default_pitch_options()
        self.sample_rate = sample_rate
        self.frame_shift = frame_shift
        self.frame_length = frame_length
        self.min_f0 = min_f0
        self.max_f0 = max_f0
        self.soft_min_f0 = soft_min_f0
        self.penalty_factor = penalty_factor
        self.lowpass_cutoff = lowpass_cutoff
        self.resample_freq = resample_freq
        self.delta_pitch = delta_pitch
        self.nccf_ballast = nccf_ballast
        self.lowpass_filter_width = lowpass_filter_width
        self.upsample_filter_width = upsample_filter_width

    @property
    def name(self):
        return 'pitch'

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
    def min_f0(self):
        """Minimum F0 to search for in Hertz"""
        return self._options.min_f0

    @min_f0.setter
    def min_f0(self, value):
        self._options.min_f0 = value

    @property
    def max_f0(self):
        """Maximum F0 to search for in Hertz"""
        return self._options.max_f0

    @max_f0.setter
    def max_f0(self, value):
        self._options.max_f0 = value

    @property
    def soft_min_f0(self):
        """Minimum F0 to search, applied in soft way, in Hertz"""
        return self._options.soft_min_f0

    @soft_min_f0.setter
    def soft_min_f0(self, value):
        self._options.soft_min_f0 = value

    @property
    def penalty_factor(self):
        """Cost factor for F0 change"""
        return np.float32(self._options.penalty_factor)

    @penalty_factor.setter
    def penalty_factor(self, value):
        self._options.penalty_factor = value

    @property
    def lowpass_cutoff(self):
        """Cutoff frequency for low-pass filter, in Hertz"""
        return self._options.lowpass_cutoff

    @lowpass_cutoff.setter
    def lowpass_cutoff(self, value):
        self._options.lowpass_cutoff = value

    @property
    def resample_freq(self):
        """Frequency that we down-sample the signal to, in Hertz"""
        return self._options.resample_freq

    @resample_freq.setter
    def resample_freq(self, value):
        self._options.resample_freq = value

    @property
    def delta_pitch(self):
        """Smallest relative change in pitch that the algorithm measures"""
        return np.float32(self._options.delta_pitch)

    @delta_pitch.setter
    def delta_pitch(self, value):
        self._options.delta_pitch = value

    @property
    def nccf_ballast(self):
        """Increasing this factor reduces NCCF for quiet frames"""
        return self._options.nccf_ballast

    @nccf_ballast.setter
    def nccf_ballast(self, value):
        self._options.nccf_ballast = value

    @property
    def lowpass_filter_width(self):
        """Integer that determines filter width of lowpass filter"""
        return self._options.lowpass_filter_width

    @lowpass_filter_width.setter
    def lowpass_filter_width(self, value):
        self._options.lowpass_filter_width = value

    @property
    def upsample_filter_width(self):
        """Integer that determines filter width when upsampling NCCF"""
        return self._options.upsample_filter_width

    @upsample_filter_width.setter
    def upsample_filter_width(self, value):
        self._options.upsample_filter_width = value

    @property
    def ndims(self):
        return 2

    def times(self, nframes):
        """Returns the time label for the rows given by the `process` method"""
        return np.vstack((
            np.arange(nframes) * self.frame_shift,
            np.arange(nframes) * self.frame_shift + self.frame_length)).T

    def _build_options(self):
        opts = _abi.default_options(_abi.KIND_PITCH)
        opts.pitch = self._options
        return opts

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
        self._check(signal)
        wave = signal.astype(np.int16).data  # force 16 bits integers
        data = _backend.get_plan(self._build_options()).run([wave])[0]
        return Features(
            data, self.times(data.shape[0]), properties=self.get_properties())

    def _process_batch(self, signals):
        for signal in signals:
            self._check(signal)
        waves = [s.astype(np.int16).data for s in signals]
        datas = _backend.get_plan(self._build_options()).run(waves, check_finite=True)
        return batch_features(datas, self.times, lambda _: self.get_properties())


class KaldiPitchPostProcessor(FeaturesPostProcessor):
    """Processes the raw (NCCF, pitch) computed by the KaldiPitchProcessor

    Output columns, in that order when enabled: POV feature, mean-subtracted log-pitch,
    delta of log-pitch, raw log-pitch.
    """
    def __init__(self, pitch_scale=2.0, pov_scale=2.0, pov_offset=0.0,
                 delta_pitch_scale=10.0, delta_pitch_noise_stddev=0.005,
                 normalization_left_context=75,
                 normalization_right_context=75,
                 delta_window=2, delay=0,
                 add_pov_feature=True, add_normalized_log_pitch=True,
                 add_delta_pitch=True, add_raw_log_pitch=False):
        super().__init__()
        self._options = _abi.default_pitch_post_options()
        self.pitch_scale = pitch_scale
        self.pov_scale = pov_scale
        self.pov_offset = pov_offset
        self.delta_pitch_scale = delta_pitch_scale
        self.delta_pitch_noise_stddev = delta_pitch_noise_stddev
        self.normalization_left_context = normalization_left_context
        self.normalization_right_context = normalization_right_context
        self.delta_window = delta_window
        self.delay = delay
        self.add_pov_feature = add_pov_feature
        self.add_normalized_log_pitch = add_normalized_log_pitch
        self.add_delta_pitch = add_delta_pitch
        self.add_raw_log_pitch = add_raw_log_pitch

    @property
    def name(self):
        return 'pitch postprocessing'

    def _float(name, doc):  # noqa
        return property(
            lambda self: getattr(self._options, name),
            lambda self, value: setattr(self._options, name, value), doc=doc)

    def _bool(name, doc):  # noqa
        return property(
            lambda self: bool(getattr(self._options, name)),
            lambda self, value: setattr(self._options, name, bool(value)),
            doc=doc)

    pitch_scale = _float(
        'pitch_scale', 'Scaling factor for the final normalized log-pitch value')
    pov_scale = _float(
        'pov_scale', 'Scaling factor for final probability of voicing feature')
    pov_offset = _float(
        'pov_offset', 'This can be used to add an offset to the POV feature')
    delta_pitch_scale = _float(
        'delta_pitch_scale', 'Term to scale the final delta log-pitch feature')
    normalization_left_context = _float(
        'normalization_left_context',
        'Left-context (in frames) for moving window normalization')
    normalization_right_context = _float(
        'normalization_right_context',
        'Right-context (in frames) for moving window normalization')
    delta_window = _float(
        'delta_window', 'Number of frames on each side of central frame')
    delay = _float(
        'delay', 'Number of frames by which the pitch information is delayed')
    add_pov_feature = _bool(
        'add_pov_feature', 'If true, the warped NCCF is added to output features')
    add_normalized_log_pitch = _bool(
        'add_normalized_log_pitch',
        'If true, the normalized log-pitch is added to output features')
    add_delta_pitch = _bool(
        'add_delta_pitch',
        'If true, time derivative of log-pitch is added to output features')
    add_raw_log_pitch = _bool(
        'add_raw_log_pitch', 'If true, log-pitch is added to output features')

    @property
    def delta_pitch_noise_stddev(self):
        """Standard deviation for noise we add to the delta log-pitch"""
        return np.float32(self._options.delta_pitch_noise_stddev)

    @delta_pitch_noise_stddev.setter
    def delta_pitch_noise_stddev(self, value):
        self._options.delta_pitch_noise_stddev = value

    @property
    def ndims(self):
        return (
            self.add_pov_feature
            + self.add_normalized_log_pitch
            + self.add_delta_pitch
            + self.add_raw_log_pitch)

    def get_properties(self, features):
        properties = copy.deepcopy(features.properties)
        properties['pitch'][self.name] = self.get_params()
        properties['pipeline'][0]['columns'] = [0, self.ndims - 1]
        return properties

    def _build_options(self):
        opts = _abi.default_options(_abi.KIND_PITCH_POST)
        opts.pitch_post = self._options
        return opts

    def _check(self, raw_pitch):
        if not (self.add_pov_feature or self.add_normalized_log_pitch
                or self.add_delta_pitch or self.add_raw_log_pitch):
            raise ValueError(
                'at least one of the following options must be True: '
                'add_pov_feature, add_normalized_log_pitch, '
                'add_delta_pitch, add_raw_log_pitch')
        if raw_pitch.shape[1] != 2:
            raise ValueError(
                'data shape must be (_, 2), but it is (_, {})'
                .format(raw_pitch.shape[1]))

    def process(self, raw_pitch):
        """Post process a raw pitch data as specified by the options"""
        self._check(raw_pitch)
        data = _backend.get_plan(self._build_options()).run_post(
            [raw_pitch.data])[0]
        return Features(
            data, raw_pitch.times, properties=self.get_properties(raw_pitch))

    def _process_batch(self, raw_pitches):
        for raw in raw_pitches:
            self._check(raw)
        datas = _backend.get_plan(self._build_options()).run_post(
            [raw.data for raw in raw_pitches], check_finite=True)
        return [Features(d, raw.times, properties=self.get_properties(raw), validate=False)
                for d, raw in zip(datas, raw_pitches)]
