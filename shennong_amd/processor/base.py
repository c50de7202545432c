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
from shennong_amd._options import F32, FLAG, SECONDS_F32, Configurable, Option
from shennong_amd.audio import Audio
from shennong_amd.base import BaseProcessor
from shennong_amd.features import Features, FeaturesCollection
from shennong_amd.utils import copy_properties, get_njobs, paused_gc


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
        out.append(Features._of_batch(data, times_cache[nframes], props_cache[key]))
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

    def _wrap_batch(self, datas, **kwargs):
        """The Features of matrices this processor computed (the tail of `_process_batch`; the sharded driver
        builds the root's collection from gathered rows with it): `kwargs` hold one value per matrix"""
        keys = [tuple(sorted((k, v[i]) for k, v in kwargs.items())) for i in range(len(datas))]
        return batch_features(datas, self.times, lambda key: self.get_properties(**dict(key)), keys)

    def process_all(self, utterances, njobs=None, **kwargs):
        """Returns features processed from several input `utterances`

        Same contract as reference processor/base.py:56-107: `njobs` is validated the same way
        (ValueError if <= 0) but only sizes the host-side audio loading; the features of all
        utterances are computed by ONE batched launch on the GPU.  Extra `kwargs` must be dicts
        keyed by utterance name and are forwarded to `process`.

        The matrices of the returned collection are row-block VIEWS of one batch-sized array (cutting
        thousands of copies out of it would dominate the call): keeping a single `Features` alive
        keeps the whole batch in memory - `Features.copy()` detaches one.
        """
        njobs = get_njobs(njobs, log=self.log)
        with paused_gc():
            return self._process_all(utterances, **kwargs)

    def _process_all(self, utterances, **kwargs):
        for name, value in kwargs.items():
            if not isinstance(value, dict):
                raise ValueError(f'argument "{name}" is not a dict')
            if value.keys() != utterances.by_name().keys():
                raise ValueError(
                    f'utterances and "{name}" have different names')
        utts = list(utterances)
        corpus = getattr(utterances, '_pinned', None)
        if corpus is not None and hasattr(self, '_process_pinned'):
            # ``Utterances.pin()``: the audio sits in one page-locked block, checked and converted when it was built
            per_utt = {k: [v[u.name] for u in utts] for k, v in kwargs.items()}
            feats = self._process_pinned(corpus, **per_utt)
            return FeaturesCollection(zip([u.name for u in utts], feats))
        if utts and hasattr(self, '_process_pinned') and all(isinstance(u._audio, str) for u in utts):
            feats = self._process_files(utts, kwargs)
            if feats is not None:
                return FeaturesCollection(zip([u.name for u in utts], feats))
        # (an utterance that is a whole in-memory Audio needs no call: load_audio is for files and segments)
        signals = [u._audio if type(u._audio) is Audio and not (u._tstart or u._tstop) else u.load_audio()
                   for u in utts]
        per_utt = {k: [v[u.name] for u in utts] for k, v in kwargs.items()}
        feats = self._process_batch(signals, **per_utt)
        return FeaturesCollection(
            (u.name, f) for u, f in zip(utts, feats))


    def _process_files(self, utts, kwargs):
        """`process_all` of utterances that are WAV files (or intervals of them): the samples of all of them are
        read side by side straight into pooled page-locked memory (shennong_amd.audio.load_int16_block: 16-bit mono
        PCM natively, other sample types through the Python reader) and the batch is uploaded from there - no
        Audio object, conversion or check per utterance.  None when the files are not one mono sample rate (the
        general path then raises what the reference raises)."""
        from shennong_amd.audio import load_int16_block, sample_range
        scans = {}
        metas = [scans.get(u._audio) or scans.setdefault(u._audio, getattr(u, '_scan', None) or Audio.scan(u._audio))
                 for u in utts]
        if any(m.nchannels != 1 or m.sample_rate != metas[0].sample_rate for m in metas):
            return None
        lengths = [sample_range(m.nsamples, m.sample_rate, u.tstart, u.tstop)[1] for u, m in zip(utts, metas)]
        corpus = _backend.StagedCorpus(lengths, metas[0].sample_rate)
        try:
            load_int16_block(utts, metas, corpus.block, corpus.soff)
            per_utt = {k: [v[u.name] for u in utts] for k, v in kwargs.items()}
            return self._process_pinned(corpus, **per_utt)
        finally:
            corpus.release()


class FramesProcessor(Configurable, FeaturesProcessor, metaclass=abc.ABCMeta):
    """Base of the frame based processors: the ten framing parameters (Kaldi
    FrameExtractionOptions; reference processor/base.py:110-268) as views on the `frame` part of
    the option record, the `times` of the output rows and the batched device call"""
    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True):
        super().__init__()
        self._configure(locals())

    sample_rate = Option('frame.samp_freq', 'Waveform sample frequency in Hertz', F32)
    frame_shift = Option('frame.frame_shift_ms', 'Frame shift in seconds', SECONDS_F32)
    frame_length = Option('frame.frame_length_ms', 'Frame length in seconds', SECONDS_F32)
    dither = Option('frame.dither', 'Amount of dithering, 0.0 means no dither', F32)
    preemph_coeff = Option('frame.preemph_coeff', 'Coefficient for use in signal preemphasis', F32)
    remove_dc_offset = Option(
        'frame.remove_dc_offset', 'If True, subtract mean from waveform on each frame', FLAG)
    round_to_power_of_two = Option(
        'frame.round_to_power_of_two',
        'If true, round window size to power of two by zero-padding the FFT input', FLAG)
    blackman_coeff = Option(
        'frame.blackman_coeff', 'Constant coefficient for generalized Blackman window', F32)
    snip_edges = Option(
        'frame.snip_edges', 'If true, output only frames that completely fit in the file', FLAG)

    @property
    def window_type(self):
        """'hamming', 'hanning', 'povey', 'rectangular' or 'blackman'"""
        code = self._record.frame.window_type
        return next(name for name, value in _abi.WINDOW_TYPES.items() if value == code)

    @window_type.setter
    def window_type(self, value):
        windows = ['hamming', 'hanning', 'povey', 'rectangular', 'blackman']
        if value not in windows:
            raise ValueError(
                'window type must be in {}, it is {}'.format(windows, value))
        self._record.frame.window_type = _abi.WINDOW_TYPES[value]

    def times(self, nframes):
        """(start, stop) of every output row in seconds: float64 multiples of the float32 shift,
        which is what the reference produces (processor/base.py:264-268)"""
        start = np.arange(nframes) * self.frame_shift
        return np.vstack((start, start + self.frame_length)).T

    def _check_signals(self, signals):
        """`check_signal` for every signal of a batch (the option is read once, the common case - a mono
        signal at the processor's rate - costs two comparisons)"""
        rate = self.sample_rate
        for signal in signals:
            if signal._data.ndim != 1 or signal._sample_rate != rate:
                check_signal(self, signal)

    def _process_pinned(self, corpus):
        """`_process_batch` over a page-locked corpus (``Utterances.pin()``, or the WAV files of an index read
        into pooled staging memory: `_process_files`): uploaded from where it lies, no per-utterance conversion or
        check (one sample rate and mono by construction)"""
        if self.sample_rate != corpus.sample_rate:
            raise ValueError(
                'processor and signal mismatch in sample rates: '
                '{} != {}'.format(self.sample_rate, corpus.sample_rate))
        return _backend.get_plan(self._build_options()).run_pinned(
            corpus, None, check_finite=True, wrap=self._wrap_pinned)

    def _wrap_pinned(self, datas):
        return batch_features(datas, self.times, lambda _: self.get_properties())

    def _run(self, opts, signals, vtln_warps=None, wrap=None):
        """One batched launch over `signals` (forced to 16 bits integers like the reference does
        before Kaldi, processor/base.py:428); the batch is validated once.  `wrap(matrices)` builds the
        Features of the batch and is called while a large batch is still in flight (see Plan.run)"""
        i16 = np.dtype(np.int16)
        waves = [s._data if s._data.dtype == i16 else s.astype(np.int16).data for s in signals]
        return _backend.get_plan(opts).run(waves, vtln_warps, check_finite=True, wrap=wrap)


class MelFeaturesProcessor(FramesProcessor):
    """Base of the mel based processors: adds the five mel-bank parameters (Kaldi MelBanksOptions;
    reference processor/base.py:271-436) and `process(signal, vtln_warp)`"""
    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True, num_bins=23, low_freq=20,
                 high_freq=0, vtln_low=100, vtln_high=-500):
        FeaturesProcessor.__init__(self)
        self._configure(locals())

    num_bins = Option('mel.num_bins', 'Number of triangular mel-frequency bins (minimum 3)')
    low_freq = Option('mel.low_freq', 'Low cutoff frequency for mel bins in Hertz', F32)
    high_freq = Option(
        'mel.high_freq', 'High cutoff frequency for mel bins in Hertz (< 0: offset from Nyquist)', F32)
    vtln_low = Option(
        'mel.vtln_low', 'Low inflection point in piecewise linear VTLN warping function', F32)
    vtln_high = Option(
        'mel.vtln_high', 'High inflection point in piecewise linear VTLN warping function', F32)

    def process(self, signal, vtln_warp=1.0):
        """Compute features with the specified options

        Optional feature-level vocal tract length normalization when `vtln_warp` != 1.0.
        Raises ValueError if `signal` is not mono or sample rates mismatch, RuntimeError for
        Kaldi-class option errors (num_bins < 3, num_ceps > num_bins, bad frequencies...).
        """
        return self._process_batch([signal], vtln_warp=[vtln_warp])[0]

    def _process_batch(self, signals, vtln_warp=None):
        self._check_signals(signals)
        warps = [1.0] * len(signals) if vtln_warp is None else list(vtln_warp)
        return self._run(self._build_options(), signals, warps,
                         wrap=lambda datas: self._wrap_batch(datas, vtln_warp=warps))

    def _process_pinned(self, corpus, vtln_warp=None):
        if self.sample_rate != corpus.sample_rate:
            raise ValueError(
                'processor and signal mismatch in sample rates: '
                '{} != {}'.format(self.sample_rate, corpus.sample_rate))
        n = corpus.soff.shape[0] - 1
        warps = [1.0] * n if vtln_warp is None else list(vtln_warp)
        return _backend.get_plan(self._build_options()).run_pinned(
            corpus, warps, check_finite=True, wrap=lambda datas: self._wrap_batch(datas, vtln_warp=warps))

    def _wrap_batch(self, datas, vtln_warp=None):
        warps = [1.0] * len(datas) if vtln_warp is None else list(vtln_warp)
        return batch_features(
            datas, self.times, lambda w: self.get_properties(vtln_warp=w), warps)
