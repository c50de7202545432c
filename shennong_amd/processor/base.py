"""Speech features extraction models: Audio --> FeaturesProcessor --> Features

Same API surface as reference shennong/processor/base.py (FeaturesProcessor :21-107,
FramesProcessor :110-268, MelFeaturesProcessor :271-436): identical constructor signatures,
defaults, read/write option attributes, ``get_params/set_params``, ``ndims``, ``name``, ``times``,
``get_properties``, ``process`` and ``process_all``.  The arithmetic runs in hand-written gfx950 HIP
kernels behind the C ABI of ``include/shennong_amd.h`` instead of pykaldi; ``process_all`` is one
batched launch over all utterances instead of a joblib thread pool.
"""

import abc

import numpy as np

from shennong_amd import _abi, _backend
from shennong_amd.base import BaseProcessor
from shennong_amd.features import Features, FeaturesCollection
from shennong_amd.utils import copy_properties, get_njobs


def check_signal(processor, signal, what='signal', dims='one dimension'):
    """The mono / sample-rate checks shared by all processors
    (reference processor/base.py:411-419)"""
    if signal.nchannels != 1:
        raise ValueError(
            '{} must have {}, but it has {}'.format(
                what, dims, signal.nchannels))
    if processor.sample_rate != signal.sample_rate:
        raise ValueError(
            'processor and signal mismatch in sample rates: '
            '{} != {}'.format(processor.sample_rate, signal.sample_rate))


def batch_features(datas, times_of, properties_of, keys=None):
    """The Features of one batched launch, with the per-utterance host work reduced to what differs
    between utterances: the matrices were validated once for the whole batch (``check_finite`` of the
    plan's run call), generated times are sorted by construction, and times / properties are built
    once per distinct (frame count) / (key, e.g. the VTLN warp) and copied"""
    times_cache, props_cache, out = {}, {}, []
    for i, data in enumerate(datas):
        key = None if keys is None else keys[i]
        nframes = data.shape[0]
        if nframes not in times_cache:
            times_cache[nframes] = times_of(nframes)
        if key not in props_cache:
            props_cache[key] = properties_of(key)
        out.append(Features(data, times_cache[nframes].copy(),
                            properties=copy_properties(props_cache[key]), validate=False))
    return out


class FeaturesProcessor(BaseProcessor, metaclass=abc.ABCMeta):
    """Base class of all the features extraction models"""
    @abc.abstractproperty
    def name(self):  # pragma: nocover
        """Name of the processor"""

    @abc.abstractproperty
    def ndims(self):  # pragma: nocover
        """Dimension of the output features frames"""

    def get_properties(self, **kwargs):
        """Return the processors properties as a dictionary"""
        params = self.get_params()
        params.update(kwargs)
        return {
            'pipeline': [
                {'name': self.name, 'columns': [0, self.ndims-1]}],
            self.name: params}

    @abc.abstractmethod
    def process(self, signal):
        """Returns features processed from an input `signal`"""

    # -- batched path -----------------------------------------------------------
    def _process_batch(self, signals, **kwargs):
        """Default batch implementation: one `process` call per signal.  Audio processors override
        it with a single device launch over all signals."""
        return [self.process(s, **{k: v[i] for k, v in kwargs.items()})
                for i, s in enumerate(signals)]

    def process_all(self, utterances, njobs=None, **kwargs):
        """Returns features processed from several input `utterances`

        Same contract as reference processor/base.py:56-107: `njobs` is validated the same way
        (ValueError if <= 0) but only sizes the host-side audio loading; the features of all
        utterances are computed by ONE batched launch on the GPU.  Extra `kwargs` must be dicts
        keyed by utterance name and are forwarded to `process`.
        """
        njobs = get_njobs(njobs, log=self.log)
        for name, value in kwargs.items():
            if not isinstance(value, dict):
                raise ValueError(f'argument "{name}" is not a dict')
            if value.keys() != utterances.by_name().keys():
                raise ValueError(
                    f'utterances and "{name}" have different names')
        utts = list(utterances)
        signals = [u.load_audio() for u in utts]
        per_utt = {k: [v[u.name] for u in utts] for k, v in kwargs.items()}
        feats = self._process_batch(signals, **per_utt)
        return FeaturesCollection(
            (u.name, f) for u, f in zip(utts, feats))


class FramesProcessor(FeaturesProcessor, metaclass=abc.ABCMeta):
    """A base class for frame based features processors (Kaldi FrameExtractionOptions)"""
    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True):
        super().__init__()
        self._frame_options = _abi.default_frame_options()
        self._window_type = 'povey'
        self.sample_rate = sample_rate
        self.frame_shift = frame_shift
        self.frame_length = frame_length
        self.dither = dither
        self.preemph_coeff = preemph_coeff
        self.remove_dc_offset = remove_dc_offset
        self.window_type = window_type
        self.round_to_power_of_two = round_to_power_of_two
        self.blackman_coeff = blackman_coeff
        self.snip_edges = snip_edges

    # float options round-trip through a C float like the reference's Kaldi structs do, so the
    # getters return np.float32 (reference processor/base.py:150-243)
    @property
    def sample_rate(self):
        """Waveform sample frequency in Hertz"""
        return np.float32(self._frame_options.samp_freq)

    @sample_rate.setter
    def sample_rate(self, value):
        self._frame_options.samp_freq = value

    @property
    def frame_shift(self):
        """Frame shift in seconds"""
        return np.float32(self._frame_options.frame_shift_ms / 1000.0)

    @frame_shift.setter
    def frame_shift(self, value):
        self._frame_options.frame_shift_ms = value * 1000.0

    @property
    def frame_length(self):
        """Frame length in seconds"""
        return np.float32(self._frame_options.frame_length_ms / 1000.0)

    @frame_length.setter
    def frame_length(self, value):
        self._frame_options.frame_length_ms = value * 1000.0

    @property
    def dither(self):
        """Amount of dithering, 0.0 means no dither"""
        return np.float32(self._frame_options.dither)

    @dither.setter
    def dither(self, value):
        self._frame_options.dither = value

    @property
    def preemph_coeff(self):
        """Coefficient for use in signal preemphasis"""
        return np.float32(self._frame_options.preemph_coeff)

    @preemph_coeff.setter
    def preemph_coeff(self, value):
        self._frame_options.preemph_coeff = value

    @property
    def remove_dc_offset(self):
        """If True, subtract mean from waveform on each frame"""
        return bool(self._frame_options.remove_dc_offset)

    @remove_dc_offset.setter
    def remove_dc_offset(self, value):
        self._frame_options.remove_dc_offset = bool(value)

    @property
    def window_type(self):
        """'hamming', 'hanning', 'povey', 'rectangular' or 'blackman'"""
        return self._window_type

    @window_type.setter
    def window_type(self, value):
        windows = ['hamming', 'hanning', 'povey', 'rectangular', 'blackman']
        if value not in windows:
            raise ValueError(
                'window type must be in {}, it is {}'.format(windows, value))
        self._window_type = value
        self._frame_options.window_type = _abi.WINDOW_TYPES[value]

    @property
    def round_to_power_of_two(self):
        """If true, round window size to power of two by zero-padding the FFT input"""
        return bool(self._frame_options.round_to_power_of_two)

    @round_to_power_of_two.setter
    def round_to_power_of_two(self, value):
        self._frame_options.round_to_power_of_two = bool(value)

    @property
    def blackman_coeff(self):
        """Constant coefficient for generalized Blackman window"""
        return np.float32(self._frame_options.blackman_coeff)

    @blackman_coeff.setter
    def blackman_coeff(self, value):
        self._frame_options.blackman_coeff = value

    @property
    def snip_edges(self):
        """If true, output only frames that completely fit in the file"""
        return bool(self._frame_options.snip_edges)

    @snip_edges.setter
    def snip_edges(self, value):
        self._frame_options.snip_edges = bool(value)

    def times(self, nframes):
        """Returns the times label for the rows given by :func:`process`
        (float64 multiples of the float32 shift, reference processor/base.py:264-268)"""
        return np.vstack((
            np.arange(nframes) * self.frame_shift,
            np.arange(nframes) * self.frame_shift + self.frame_length)).T

    # -- shared device path --------------------------------------------------------
    def _options(self, kind):
        """A by-value copy of the option structs (reference processor/base.py:421-425)"""
        opts = _abi.default_options(kind)
        opts.frame = self._frame_options
        return opts

    def _run(self, opts, signals, vtln_warps=None):
        waves = [s.astype(np.int16).data for s in signals]  # force 16 bits integers
        return _backend.get_plan(opts).run(waves, vtln_warps, check_finite=True)


class MelFeaturesProcessor(FramesProcessor):
    """A base class for mel-based features processors (Kaldi MelBanksOptions)"""
    _kind = None

    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True, num_bins=23, low_freq=20,
                 high_freq=0, vtln_low=100, vtln_high=-500):
        super().__init__(
            sample_rate=sample_rate,
            frame_shift=frame_shift,
            frame_length=frame_length,
            dither=dither,
            preemph_coeff=preemph_coeff,
            remove_dc_offset=remove_dc_offset,
            window_type=window_type,
            round_to_power_of_two=round_to_power_of_two,
            blackman_coeff=blackman_coeff,
            snip_edges=snip_edges)
        self._mel_options = _abi.default_mel_options()
        self.num_bins = num_bins
        self.low_freq = low_freq
        self.high_freq = high_freq
        self.vtln_low = vtln_low
        self.vtln_high = vtln_high

    @property
    def num_bins(self):
        """Number of triangular mel-frequency bins (minimum 3)"""
        return self._mel_options.num_bins

    @num_bins.setter
    def num_bins(self, value):
        self._mel_options.num_bins = value

    @property
    def low_freq(self):
        """Low cutoff frequency for mel bins in Hertz"""
        return np.float32(self._mel_options.low_freq)

    @low_freq.setter
    def low_freq(self, value):
        self._mel_options.low_freq = value

    @property
    def high_freq(self):
        """High cutoff frequency for mel bins in Hertz (< 0: offset from Nyquist)"""
        return np.float32(self._mel_options.high_freq)

    @high_freq.setter
    def high_freq(self, value):
        self._mel_options.high_freq = value

    @property
    def vtln_low(self):
        """Low inflection point in piecewise linear VTLN warping function"""
        return np.float32(self._mel_options.vtln_low)

    @vtln_low.setter
    def vtln_low(self, value):
        self._mel_options.vtln_low = value

    @property
    def vtln_high(self):
        """High inflection point in piecewise linear VTLN warping function"""
        return np.float32(self._mel_options.vtln_high)

    @vtln_high.setter
    def vtln_high(self, value):
        self._mel_options.vtln_high = value

    def _options(self, kind):
        opts = super()._options(kind)
        opts.mel = self._mel_options
        return opts

    def _build_options(self):  # pragma: nocover
        raise NotImplementedError

    def process(self, signal, vtln_warp=1.0):
        """Compute features with the specified options

        Optional feature-level vocal tract length normalization when `vtln_warp` != 1.0.
        Raises ValueError if `signal` is not mono or sample rates mismatch, RuntimeError for
        Kaldi-class option errors (num_bins < 3, num_ceps > num_bins, bad frequencies...).
        """
        check_signal(self, signal)
        data = self._run(self._build_options(), [signal], [vtln_warp])[0]
        return Features(
            data, self.times(data.shape[0]),
            properties=self.get_properties(vtln_warp=vtln_warp))

    def _process_batch(self, signals, vtln_warp=None):
        for signal in signals:
            check_signal(self, signal)
        warps = [1.0] * len(signals) if vtln_warp is None else list(vtln_warp)
        datas = self._run(self._build_options(), signals, warps)
        return batch_features(
            datas, self.times, lambda w: self.get_properties(vtln_warp=w), warps)
