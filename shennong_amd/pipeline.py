"""High-level features extraction pipeline

Same entry points, configuration dictionary, checks and result layout as the reference's
``shennong/pipeline.py`` (``get_default_config``:97-210, ``extract_features``:213-280, the two-pass
ordering of ``_extract_features``:525-567 and ``_extract_pass_one/_two``:570-643, the processor
wiring of ``pipeline_manager.py``:247-313), but every stage runs as ONE batched launch over all the
utterances instead of a per-utterance thread pool:

    features (+ VTLN warps) -> [energy -> VAD -> CMVN statistics per speaker / utterance]
    -> [pitch -> pitch post-processing] -> CMVN apply -> delta -> pitch concatenation (tolerance 2)

Not provided by this backend (out of scope): VTLN *training* (`with_vtln`, the 'vtln'
configuration entry; precomputed `warps` are supported), CREPE pitch, bottleneck features.
"""

import os
import threading

import numpy as np
import yaml

from shennong_amd import _abi, _backend
from shennong_amd.features import Features, FeaturesCollection
from shennong_amd.logger import get_logger
from shennong_amd.postprocessor.cmvn import (
    CmvnPostProcessor, _fake_stats_for_dims)  # noqa: F401
from shennong_amd.utils import copy_properties, get_njobs


_PROCESSORS = {
    'energy': ('processor', 'EnergyProcessor'),
    'filterbank': ('processor', 'FilterbankProcessor'),
    'mfcc': ('processor', 'MfccProcessor'),
    'kaldi_pitch': ('processor', 'KaldiPitchProcessor'),
    'kaldi_pitch_post': ('processor', 'KaldiPitchPostProcessor'),
    'plp': ('processor', 'PlpProcessor'),
    'spectrogram': ('processor', 'SpectrogramProcessor'),
    'cmvn': ('postprocessor', 'CmvnPostProcessor'),
    'delta': ('postprocessor', 'DeltaPostProcessor'),
    'sliding_window_cmvn': ('postprocessor', 'SlidingWindowCmvnPostProcessor'),
    'vad': ('postprocessor', 'VadPostProcessor')}


def valid_features():
    """The main features the pipeline can extract (post-processing excluded)"""
    return ['spectrogram', 'filterbank', 'mfcc', 'plp']


def _processor_class(name):
    try:
        module, cls = _PROCESSORS[name]
    except KeyError:
        raise ValueError('invalid processor "{}"'.format(name)) from None
    import importlib
    return getattr(importlib.import_module(f'shennong_amd.{module}'), cls)


def _processor_params(name):
    return _processor_class(name)().get_params()


def get_default_config(features, to_yaml=False, yaml_commented=True,
                       with_pitch=False, with_cmvn=False, with_delta=False,
                       with_vtln=False):
    """Returns the default configuration for the specified pipeline

    Same dictionary layout as the reference (one entry per processor, parameters with their default
    values, `sample_rate` and `htk_compat` filtered out of the features entry, frame parameters
    filtered out of the pitch entry).

    Raises
    ------
    ValueError
        If `features` is not in :func:`valid_features`, or if `with_pitch` / `with_vtln` ask for
        something this backend does not provide ('crepe', VTLN training).
    """
    if features not in valid_features():
        raise ValueError('invalid features "{}", must be in {}'.format(
            features, ', '.join(valid_features())))
    if with_pitch not in (False, 'kaldi', 'crepe'):
        raise ValueError(
            f'with_pitch argument must be False, "kaldi" or "crepe" '
            f'but is "{with_pitch}"')
    if with_pitch == 'crepe':
        raise ValueError('crepe pitch is not available in this backend')
    if with_vtln not in (False, 'simple', 'full'):
        raise ValueError(
            f'with_vtln argument must be False, "simple" or "full" '
            f'but is "{with_vtln}"')
    if with_vtln:
        raise ValueError(
            'VTLN training is not available in this backend '
            '(precomputed warps can be given to extract_features)')

    config = {}
    config[features] = {
        k: v for k, v in _processor_params(features).items()
        if k not in ('sample_rate', 'htk_compat')}

    if with_pitch:
        config['pitch'] = {'processor': with_pitch}
        for key, value in _processor_params('kaldi_pitch').items():
            if key not in ('frame_length', 'frame_shift', 'sample_rate'):
                config['pitch'][key] = value
        config['pitch']['postprocessing'] = _processor_params('kaldi_pitch_post')

    if with_cmvn:
        config['cmvn'] = {'by_speaker': True, 'with_vad': True}
        config['cmvn']['vad'] = _processor_params('vad')

    if with_delta:
        config['delta'] = _processor_params('delta')

    if to_yaml:
        return _get_config_to_yaml(config, comments=yaml_commented)
    return config


def _get_config_to_yaml(config, comments=True):
    """Dict -> YAML string; with `comments` the parameters docstrings are interleaved as in the
    reference (pipeline.py:315-416)"""
    import re
    import textwrap

    class _Dumper(yaml.SafeDumper):
        pass
    _Dumper.add_representer(
        dict, lambda self, data: self.represent_dict(data.items()))
    _Dumper.add_representer(
        np.float32, lambda self, data: self.represent_float(float(data)))
    _Dumper.add_representer(
        np.float64, lambda self, data: self.represent_float(float(data)))
    text = yaml.dump(config, Dumper=_Dumper).strip()
    if not comments:
        return text + '\n'

    def docstring(processor, param, default):
        doc = getattr(_processor_class(processor), param).__doc__ or ''
        doc = re.sub(r'\n\n', '. ', doc)
        doc = re.sub(r'\n', ' ', doc)
        doc = re.sub(r'`', '', doc)
        doc = re.sub(':func:', '', doc)
        doc += '. Default is {}.'.format(default)
        doc = re.sub(r'\.+', '.', doc)
        doc = re.sub(r' +', ' ', doc)
        doc = re.sub(r'\. \.', '.', doc)
        return doc.strip()

    out, processors, prev_offset = [], [], 0
    for line in text.split('\n'):
        head = line.split(': ')[0]
        offset = len(head) - len(head.strip())
        for _ in range((prev_offset - offset) // 2):
            processors.pop()
        if line.endswith(':'):
            processor = line[:-1].strip()
            if processor == 'postprocessing':
                processor = f'{processors[-1]}_post'
            processors.append(processor)
            if processor == 'vad':
                out.append("  # The vad options are not used if 'with_vad' is false")
            out.append(line)
        else:
            param = line.split(': ')[0].strip()
            default = line.split(': ')[1].strip()
            processor = processors[-1]
            if processor == 'cmvn' and param == 'by_speaker':
                doc = ('If false, do normalization by utterance, '
                       'if true do normalization by speaker.')
            elif processor == 'cmvn' and param == 'with_vad':
                doc = ('If true do normalization only on frames where '
                       'voice activity has been detected, if false do not '
                       'consider voice activity for normalization.')
            elif processor == 'pitch' and param == 'processor':
                doc = 'Computing pitch using kaldi'
            elif 'pitch' in processor:
                doc = docstring('kaldi_' + processor, param, default)
            else:
                doc = docstring(processor, param, default)
            out += [' ' * offset + '# ' + w
                    for w in textwrap.wrap(doc, width=68 - offset)]
            out.append(line)
        prev_offset = offset
    return '\n'.join(out) + '\n'


def _init_config(config, log=get_logger('pipeline', 'warning')):
    """Loads (dict, YAML string or YAML file) and validates a configuration; same checks and
    messages as reference pipeline.py:419-493"""
    try:
        if os.path.isfile(config):
            log.debug('loading configuration from %s', config)
            config = open(config, 'r').read()
    except TypeError:
        pass
    if isinstance(config, str):
        try:
            config = yaml.load(config, Loader=yaml.FullLoader)
        except yaml.YAMLError as err:
            raise ValueError(f'error in configuration: {err}') from None

    known = valid_features() + ['cmvn', 'delta', 'pitch', 'vtln', 'bottleneck']
    unknown_keys = [k for k in config.keys() if k not in known]
    if unknown_keys:
        raise ValueError(
            'invalid keys in configuration: {}'.format(', '.join(unknown_keys)))
    if 'bottleneck' in config:
        raise ValueError('bottleneck features are not available in this backend')

    features = [k for k in config.keys() if k in valid_features()]
    if not features:
        raise ValueError(
            'the configuration does not define any features extraction '
            '(must have one and only one entry of {})'
            .format(', '.join(valid_features())))
    if len(features) > 1:
        raise ValueError(
            'more than one features extraction processors are defined, '
            '(must have one and only one entry of {}): {}'
            .format(', '.join(valid_features()), ', '.join(features)))

    if 'vtln' in config:
        if features[0] == 'spectrogram':
            raise ValueError(f'{features[0]} features do not support VTLN')
        raise ValueError(
            'VTLN training is not available in this backend '
            '(give precomputed warps to extract_features instead)')

    if 'cmvn' in config:
        if 'by_speaker' not in config['cmvn']:
            log.warning(
                'by_speaker option not specified for cmvn, '
                'assuming it is false and doing cmvn by utterance')
            config['cmvn']['by_speaker'] = False
        if 'with_vad' not in config['cmvn']:
            config['cmvn']['with_vad'] = True

    if 'pitch' in config:
        if config['pitch'].get('processor', 'kaldi') != 'kaldi':
            raise ValueError('only the kaldi pitch processor is available in this backend')
        if 'postprocessing' not in config['pitch']:
            config['pitch']['postprocessing'] = {}

    msg = []
    if 'pitch' in config:
        msg.append('kaldi pitch')
    if 'delta' in config:
        msg.append('delta')
    if 'cmvn' in config:
        msg.append('cmvn by {}{}'.format(
            'speaker' if config['cmvn']['by_speaker'] else 'utterance',
            ' with vad' if config['cmvn']['with_vad'] else ''))
    log.info(
        'pipeline configured for %s features extraction%s',
        features[0], ' with {}'.format(', '.join(msg)) if msg else '')
    return config


def _init_warps(warps, config, utterances, log):
    """Per-utterance float warps from warps given by utterance or by speaker
    (reference pipeline.py:496-522)"""
    features = [k for k in config.keys() if k in valid_features()][0]
    if features == 'spectrogram':
        raise ValueError(f'{features} features do not support VTLN')
    if 'vtln' in config:  # pragma: nocover (rejected by _init_config)
        raise ValueError(
            'warps are given but "vtln" processor already defined '
            'in the configuration')
    if warps.keys() == utterances.by_name().keys():
        log.info('VTLN warps are defined by utterance')
    elif not utterances.has_speakers() or \
            warps.keys() != utterances.by_speaker().keys():
        raise ValueError(
            'warps do not match utterances, either by speaker or by utterance')
    else:
        log.info('VTLN warps are defined by speaker')
        warps = {utt.name: warps[utt.speaker] for utt in utterances}
    return {name: float(warp) for name, warp in warps.items()}


def extract_features(configuration, utterances, warps=None, njobs=1,
                     log=get_logger('pipeline', 'warning')):
    """Speech features extraction pipeline

    Parameters
    ----------
    configuration : dict or str
        The pipeline configuration: a dictionary, a path to a YAML file or a YAML string
        (see :func:`get_default_config`).
    utterances : :class:`~shennong_amd.utterances.Utterances`
        The utterances to extract the features on.
    warps : dict, optional
        Precomputed VTLN warps (str: float) indexed by utterance name or by speaker.
    njobs : int, optional
        Validated like the reference; the work itself is batched on the GPU.

    Returns
    -------
    features : FeaturesCollection, one :class:`Features` per utterance, keyed by name

    Raises
    ------
    ValueError
        If the configuration, the utterances or the warps are invalid.
    """
    get_njobs(njobs, log=log)
    config = _init_config(configuration, log=log)
    log.info('detected format for utterances index is: %s',
             utterances.format(type=str))
    if warps:
        warps = _init_warps(warps, config, utterances, log)
    return _extract_features(config, utterances, warps, log)


def _batches(utterances, max_duration):
    """Consecutive runs of `utterances` (order kept) of at most `max_duration` seconds of audio each;
    an utterance longer than that is a batch of its own"""
    batch, total = [], 0.0
    for utt in utterances:
        duration = utt.duration
        if batch and total + duration > max_duration:
            yield batch
            batch, total = [], 0.0
        batch.append(utt)
        total += duration
    if batch:
        yield batch


_BATCH_POOL = None


def _batch_pool():
    """The threads of the batches in flight, made once: a thread keeps its copy stream (``_backend._copy_stream``
    is per thread) and its staging buffers from one corpus to the next instead of leaking them with a
    short-lived pool per call.  Eight threads: the most batches `extract_features_streamed` keeps in flight."""
    global _BATCH_POOL
    if _BATCH_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        with _backend._LOCK:   # (two first callers must not build two pools: the loser's never shuts down)
            if _BATCH_POOL is None:
                _BATCH_POOL = ThreadPoolExecutor(max_workers=8, thread_name_prefix='snf-batch')
    return _BATCH_POOL


def _in_flight(batches, work, depth):
    """``work(b, batch)`` for every batch, results in order, at most `depth` batches started and not yet
    handed over.  With `depth` > 1 the batches run on threads: the staging copy, the transfers and the
    launches of one (all outside the interpreter lock) overlap the per-utterance bookkeeping of another.
    Closing the generator early (an error in the consumer) cancels what has not started and WAITS for what
    is running: nothing touches the caller's buffers after it returns."""
    if depth <= 1:
        for b, batch in enumerate(batches):
            yield work(b, batch)
        return
    from collections import deque
    from concurrent.futures import wait
    pending = deque()
    pool = _batch_pool()
    try:
        for b, batch in enumerate(batches):
            pending.append(pool.submit(work, b, batch))
            if len(pending) >= depth:
                yield pending.popleft().result()
        while pending:
            yield pending.popleft().result()
    finally:
        for future in pending:
            future.cancel()
        wait(list(pending))


_BATCH_BYTES_PER_HOUR = 4 << 30   # HBM one hour of audio needs while its batch is in flight: 2.3 GB of pitch
                                  # scratch (19 GB per 10 000 x 3 s), audio, features, CMVN / delta copies


def default_batch_duration(depth=1):
    """Seconds of audio per streamed batch when the caller names none: four hours (measured best on a
    288 GB MI355X, see extract_features_streamed) unless `depth` such batches in flight would need more than
    half of the HBM that is free right now (_BATCH_BYTES_PER_HOUR each) - smaller devices get smaller batches
    instead of an out-of-memory error half way through a corpus; never below ten minutes."""
    if _backend.device_count() < 1:   # (host-logic tests with the device pipeline replaced by a stand-in)
        return 14400.0
    free, _ = _backend.mem_info()
    hours = (free // 2) / float(_BATCH_BYTES_PER_HOUR * max(int(depth), 1))
    return float(min(14400.0, max(600.0, 3600.0 * hours)))


def extract_features_streamed(configuration, utterances, sink, warps=None,
                              max_batch_duration=None, njobs=1, stats_reduce=None,
                              resident_bytes=16 << 30,
                              log=get_logger('pipeline', 'warning')):
    """:func:`extract_features` for a corpus that must not sit in memory at once (BASELINE config 5)

    The utterances are processed in consecutive batches of at most `max_batch_duration` seconds of
    audio (None: :func:`default_batch_duration` - four hours unless `njobs` batches of that size in flight
    would need more than half of the free HBM; one hour of 16 kHz audio is 115 MB of int16 up and, for 123
    columns, 177 MB of float32 down and needs ~4 GB of HBM while in flight; four hours were measured on an 8 h corpus of 3 s utterances: 1 h batches 54-57 hours of
    audio per second, 2 h 54-65, 4 h 59-64, the whole 8 h in one batch 70 - the pitch tracker's cost per utterance
    halves between 1 000 and 4 000 utterances per call; the results of a batch must fit a pooled page-locked block
    of at most 2 GiB (_backend._ResultBlock._FRESH): 4 h of 257 columns are 1.5 GB);
    each batch goes through the device-resident pipeline and its FeaturesCollection is
    handed to `sink` (a callable, e.g. ``KaldiStreamWriter.write``) and dropped.  The results are
    those of :func:`extract_features` on the whole corpus - bit for bit when no random term is
    configured (dither 0, delta-pitch noise 0; the noise of a frame depends on the call and on the
    frame's position in its batch, like the reference's global rand() stream): with CMVN by speaker the
    statistics need every utterance of a speaker before any can be normalised, so a first pass over
    the batches accumulates them (features + VAD only, summed in utterance order like the one-shot
    pipeline does) and the second pass recomputes the features instead of keeping them - on this
    hardware the features are cheaper to recompute than to store.  What the first pass does keep is the
    uploaded int16 audio, in HBM, up to `resident_bytes` (96 kB per 3 s utterance; 0 = keep nothing): the
    second pass of those batches starts from the device buffers - no file is read twice, nothing crosses
    the host link twice.  `njobs` (the reference's number of parallel jobs, pipeline.py:340-377) is the
    number of batches in flight: each runs on its own thread, `sink` is always called from the caller's
    thread and in corpus order.

    `stats_reduce(names, stats) -> stats` sums the speakers' statistics across processes when the
    corpus is sharded (see shennong_amd.distributed.extract_features_streamed_sharded).

    Returns the number of utterances written."""
    from shennong_amd.utterances import Utterances
    depth = min(get_njobs(njobs, log=log), 8)
    config = _init_config(configuration, log=log)
    if max_batch_duration is not None and not max_batch_duration > 0:
        raise ValueError('max_batch_duration must be strictly positive')
    if warps:
        warps = _init_warps(warps, config, utterances, log)
    by_speaker = 'cmvn' in config and config['cmvn']['by_speaker']
    if by_speaker and not utterances.has_speakers():
        raise ValueError(
            'cmvn normalization by speaker requested '
            'but no speaker information provided')
    utts = list(utterances)
    if max_batch_duration is None:
        max_batch_duration = default_batch_duration(depth)

    def sub(batch):
        return {u.name: warps[u.name] for u in batch} if warps else None

    hook = None
    resident = _ResidentWaves(resident_bytes) if by_speaker and resident_bytes > 0 else None
    running = None   # the generator of the batches in flight: closed (its threads joined) before the audio
                     # buffers are released, whatever ends the pass - the end of the corpus, an error in a
                     # batch, an exception from `sink`
    try:
        if by_speaker:
            total = {}

            def first_pass(b, batch):
                return _extract_features(config, Utterances(batch), sub(batch), log, stats_only=True,
                                         resident=resident, batch_id=b)

            running = _in_flight(_batches(utts, max_batch_duration), first_pass, depth)
            for speakers, per_utt in running:
                for speaker, stats in zip(speakers, per_utt):
                    if speaker in total:
                        total[speaker] += stats
                    else:
                        total[speaker] = stats.copy()
            running = None
            if stats_reduce is not None:
                names = list(total)
                reduced = stats_reduce(names, np.stack([total[k] for k in names]) if names
                                       else np.zeros((0, 2, 1), dtype=np.float64))
                total = dict(zip(names, reduced))

            def hook(names, _partial):
                return np.stack([total[k] for k in names])

        def second_pass(b, batch):
            return _extract_features(config, Utterances(batch), sub(batch), log, stats_hook=hook,
                                     resident=resident, batch_id=b)

        count = 0
        batches = running = _in_flight(_batches(utts, max_batch_duration), second_pass, depth)
        while True:
            # (nothing of batch k is referenced here while batch k + 1 is made: its page-locked result block
            # is back in the pool by then, see _backend.result_array)
            features = next(batches, None)
            if features is None:
                break
            sink(features)
            count += len(features)
            del features
        return count
    finally:
        if running is not None:
            running.close()   # (GeneratorExit inside _in_flight: pending batches cancelled, its pool joined)
        if resident is not None:
            resident.clear()


# The device-resident pipeline draws its random terms (dither, delta-pitch noise) from ONE named noise call
# (snf_set_noise_call): the features of an utterance are a function of the configuration and the utterance
# alone - the same in the statistics pass and the apply pass of the streamed pipeline, in any batch split,
# in every run (the reference, whose dither comes from Kaldi's global rand(), has none of these).
_NOISE_CALL = int(os.environ.get('SNF_NOISE_CALL', '1'))   # (another run of the same corpus with other noise: export another number)


class _Meta:
    """What the post-processors' `get_properties` need to know about features that live in HBM.
    The properties of a stage are the same for every utterance that went through the same processors
    with the same per-utterance arguments (warp factor, CMVN group): `key` names that history and
    `cache` holds one properties dictionary per history, copied once per utterance at the end."""
    __slots__ = ('_properties', '_source', 'ndims', 'nframes', 'times', 'key', '_derived')

    def __init__(self, properties, ndims, nframes, times, key=None, source=None):
        self._properties = properties
        self._source = source      # (cache, parent, make_properties) while the properties are not made yet
        self.ndims = ndims
        self.nframes = nframes
        self.times = times
        self.key = key
        self._derived = {}

    @property
    def properties(self):
        """The properties dictionary of this history, made when first read: a corpus run that writes the
        matrices and never looks at the properties (1 000 speakers x CMVN statistics x delta x pitch columns
        in every batch: two thirds of the host time of BASELINE config 5 before round 5) derives none"""
        if self._properties is None:
            cache, parent, make_properties = self._source
            found = cache.get(self.key)
            if found is None:
                found = cache[self.key] = make_properties(parent)
            self._properties, self._source = found, None
        return self._properties

    def derive(self, cache, tag, make_properties, ndims=None, nframes=None, times=None):
        """The _Meta after one more stage.  Utterances with the same history and frame count share one
        _Meta object, so the second utterance to take the same step finds the result here."""
        memo = (tag, ndims, nframes, id(times))
        found = self._derived.get(memo)
        if found is not None:
            return found
        key = (self.key, tag)
        found = self._derived[memo] = _Meta(
            cache.get(key), self.ndims if ndims is None else ndims,
            self.nframes if nframes is None else nframes,
            self.times if times is None else times, key, source=(cache, self, make_properties))
        return found


class _ResidentWaves:
    """Uploaded waveforms kept in HBM between the two passes of :func:`extract_features_streamed`
    (96 kB per 3 s utterance: 16 GiB hold 140 hours of 16 kHz audio), so that the second pass neither reads
    the audio files nor crosses the host link again.  Batches that do not fit the budget are re-uploaded."""
    def __init__(self, budget):
        self.budget = int(budget)
        self.held = 0
        self._items = {}
        self._lock = threading.Lock()  # (the batches in flight run on threads)

    def offer(self, key, d_wave, soff):
        with self._lock:
            if self.held + d_wave.nbytes > self.budget:
                return False
            self._items[key] = (d_wave, soff)
            self.held += d_wave.nbytes
            return True

    def take(self, key):
        with self._lock:
            item = self._items.pop(key, None)
            if item is not None:
                self.held -= item[0].nbytes
            return item

    def clear(self):
        with self._lock:   # (a batch thread that is still running may offer / take meanwhile)
            items = list(self._items.values())
            self._items.clear()
            self.held = 0
        for d_wave, _ in items:
            d_wave.free(synced=True)


def _extract_features(config, utterances, warps, log, tolerance=2, stats_hook=None,
                      stats_only=False, resident=None, batch_id=None, device_out=None):
    from shennong_amd.utils import paused_gc
    with paused_gc():   # (thousands of small objects per batch, none of them garbage: see utils.paused_gc)
        return _extract_features_body(config, utterances, warps, log, tolerance, stats_hook, stats_only,
                                      resident, batch_id, device_out)


def _extract_features_body(config, utterances, warps, log, tolerance=2, stats_hook=None,
                           stats_only=False, resident=None, batch_id=None, device_out=None):
    """The whole pipeline with the intermediate features resident in HBM: the waveforms go up once,
    every stage is one batched launch on device buffers (features, energy -> VAD, CMVN statistics and
    apply, delta, pitch and its post-processing, column concatenation), the final matrices come down
    once.  Stage order, arithmetic and properties are those of reference pipeline.py:525-643.

    `stats_only` (first pass of :func:`extract_features_streamed`): stop after the CMVN accumulation
    and return ``(group name of every utterance, per-utterance statistics [n, 2, dim + 1])``; the
    pitch stage, which the statistics do not depend on, is skipped.  `resident` (a _ResidentWaves) keeps
    the uploaded waveforms of batch `batch_id` in HBM after that pass and hands them to the next one.
    `device_out` (a list; :func:`shennong_amd.distributed.extract_features_sharded`): the final matrices
    STAY in HBM - one ``(DeviceBuffer [rows, ndims], names in row order, ndims)`` per sample rate is appended
    and the caller owns the buffers; the returned Features carry times and properties over data that were
    never downloaded (untouched host pages)."""
    features_name = [k for k in config.keys() if k in valid_features()][0]
    with_cmvn = 'cmvn' in config
    if with_cmvn and config['cmvn']['by_speaker'] and not utterances.has_speakers():
        raise ValueError(
            'cmvn normalization by speaker requested '
            'but no speaker information provided')

    from shennong_amd.audio import Audio
    utts = list(utterances)
    n = len(utts)
    metadata = {}
    for utt in utts:
        key = utt.audio_file if isinstance(utt.audio_file, str) else id(utt.audio_file)
        if key not in metadata:
            metadata[key] = Audio.scan(utt.audio_file)
    meta_of = [metadata[u.audio_file if isinstance(u.audio_file, str) else id(u.audio_file)]
               for u in utts]
    speakers = ('' if not utterances.has_speakers() else ' from {} speakers'.format(
        len(set(u.speaker for u in utts))))
    import datetime
    log.info('get %s utterances%s in %s audio files, total duration: %s',
             len(utts), speakers, len(metadata),
             datetime.timedelta(seconds=utterances.duration()))
    if not all(m.nchannels == 1 for m in meta_of):
        raise ValueError('all audio files are not mono')
    samplerates = sorted(set(m.sample_rate for m in meta_of))
    if len(samplerates) > 1:
        log.warning(
            'several sample rates found in audio files: %s, features '
            'extraction pipeline will work but this may not be a good '
            'idea to work on heterogeneous data',
            ', '.join(str(s) + 'Hz' for s in samplerates))

    from shennong_amd.processor.base import check_signal
    # (every launch below goes through an entry point that synchronises its stream before it returns, and the
    # one asynchronous copy is waited for before its buffer is released: the buffers are given back with
    # synced=True, without the device-wide wait DeviceBuffer.free() otherwise makes - which would stall this
    # thread behind the pitch tracker of the side thread)
    DB = _backend.DeviceBuffer
    frame_length = frame_shift = None
    cache = {}          # properties per processing history (see _Meta)
    groups_state = []   # per sample rate: buffers and tables of its utterances
    meta = [None] * n   # _Meta of the main features
    pmeta = [None] * n  # _Meta of the pitch features

    def offsets(counts):
        off = np.zeros(len(counts) + 1, dtype=np.int64)
        np.cumsum(counts, out=off[1:])
        return off

    # ---- pass one: features, (energy -> VAD), pitch ---------------------------------------------------
    for rate in samplerates:
        idx = [i for i in range(n) if meta_of[i].sample_rate == rate]
        proc = _processor_class(features_name)(**config[features_name])
        proc.sample_rate = rate
        if frame_length is None:
            frame_length, frame_shift = proc.frame_length, proc.frame_shift
        held = resident.take((batch_id, rate)) if resident is not None else None
        if held is not None:
            d_wave, soff = held
        else:
            waves, checked = [], set()
            for i in idx:
                audio = utts[i].load_audio()
                if (audio.nchannels, audio.sample_rate) not in checked:  # (one check per kind of signal)
                    check_signal(proc, audio)
                    checked.add((audio.nchannels, audio.sample_rate))
                waves.append(audio.astype(np.int16).data)
            soff = offsets([w.shape[0] for w in waves])
            d_wave = _backend.upload_rows(waves, np.int16)  # (page-locked staging: full link rate)
            del waves
        lengths = np.diff(soff)

        def frame_offsets(a_plan):
            frames_of = {x: a_plan.num_frames(x) for x in np.unique(lengths).tolist()}
            return offsets([frames_of[x] for x in lengths.tolist()])
        st = {'idx': idx, 'soff': soff, 'd_wave': d_wave}

        if 'pitch' in config and not stats_only:
            params = {k: v for k, v in config['pitch'].items()
                      if k not in ('processor', 'postprocessing')}
            params['sample_rate'] = rate
            params['frame_shift'] = frame_shift
            params['frame_length'] = frame_length
            pproc = _processor_class('kaldi_pitch')(**params)
            post = _processor_class('kaldi_pitch_post')(**config['pitch']['postprocessing'])
            pplan = _backend.get_plan(pproc._build_options())
            pfoff = frame_offsets(pplan)
            qplan = _backend.get_plan(post._build_options())
            pdim = qplan.post_ndims(2)

            def track(d_wave=d_wave, soff=soff, pfoff=pfoff, pplan=pplan, qplan=qplan, pdim=pdim):
                # the tracker needs nothing but the audio: it runs on a side thread (its own stream) while
                # this one takes the audio through the features, VAD, CMVN and delta, and is waited for where
                # the columns are joined (the audio buffer is released there, after its last reader)
                d_raw = d_pitch = None
                try:
                    d_raw = DB(max(int(pfoff[-1]) * 2 * 4, 16))
                    pplan.run_device(d_wave.ptr, soff, pfoff, d_raw.ptr)
                    d_pitch = DB(max(int(pfoff[-1]) * pdim * 4, 16))
                    qplan.run_post_device(d_raw.ptr, 2, pfoff, d_pitch.ptr, noise_call=_NOISE_CALL)
                except BaseException:
                    # a call that failed half way may have kernels enqueued that still write these blocks:
                    # the plain free() waits for the device before the pool can hand them to another thread
                    for block in (d_pitch, d_raw):
                        if block is not None:
                            block.free()
                    raise
                d_raw.free(synced=True)   # (both calls returned: their streams are synchronised)
                return d_pitch

            st.update(pfoff=pfoff, pdim=pdim, pitch_job=_backend.side_pool().submit(track))
            step = {}
            for i, t in zip(idx, np.diff(pfoff).tolist()):
                found = step.get(t)
                if found is None:
                    key = ('pitch', rate)
                    if key not in cache:
                        cache[key] = pproc.get_properties()
                    tkey = ('times', 'pitch', rate, t)
                    if tkey not in cache:
                        cache[tkey] = pproc.times(t)
                    found = step[t] = _Meta(cache[key], 2, t, cache[tkey], key).derive(
                        cache, 'post', post.get_properties, ndims=pdim)
                pmeta[i] = found

        opts = proc._build_options()
        plan = _backend.get_plan(opts)
        dim = plan.ndims
        foff = frame_offsets(plan)
        d_feat = DB(max(int(foff[-1]) * dim * 4, 16))
        vt = wlist = None
        if warps and features_name != 'spectrogram':
            wlist = [warps[utts[i].name] for i in idx]
            vt = np.asarray(wlist, dtype=np.float32)
        log.debug('extract %s on %d utterances at %d Hz', features_name, len(idx), rate)
        plan.run_device(d_wave.ptr, soff, foff, d_feat.ptr, vtln_warps=vt, noise_call=_NOISE_CALL)
        st.update(foff=foff, dim=dim, d_feat=d_feat)
        step = {}
        for k, (i, t) in enumerate(zip(idx, np.diff(foff).tolist())):
            warp = None if features_name == 'spectrogram' else wlist[k] if wlist is not None else 1.0
            found = step.get((warp, t))
            if found is None:
                key = (features_name, rate, warp)
                if key not in cache:
                    cache[key] = proc.get_properties(**({} if warp is None else {'vtln_warp': warp}))
                tkey = ('times', features_name, rate, t)
                if tkey not in cache:
                    cache[tkey] = proc.times(t)
                found = step[(warp, t)] = _Meta(cache[key], dim, t, cache[tkey], key)
            meta[i] = found

        if with_cmvn and config['cmvn']['with_vad']:
            energy = _processor_class('energy')()
            energy.frame_length = frame_length
            energy.frame_shift = frame_shift
            energy.sample_rate = rate
            eplan = _backend.get_plan(energy._build_options())
            efoff = frame_offsets(eplan)
            if not np.array_equal(efoff, foff):
                raise ValueError('energy and features differ in number of frames')
            d_energy = DB(max(int(foff[-1]) * 4, 16))
            eplan.run_device(d_wave.ptr, soff, foff, d_energy.ptr, noise_call=_NOISE_CALL)
            vad = _processor_class('vad')(**config['cmvn']['vad'])
            d_vad = DB(max(int(foff[-1]) * 4, 16))
            _backend.get_plan(vad._build_options()).run_post_device(
                d_energy.ptr, 1, foff, d_vad.ptr)
            d_energy.free(synced=True)
            st['d_vad'] = d_vad

        if 'pitch_job' in st:
            pass  # (the tracker still reads the audio: released where it is waited for)
        elif not (stats_only and resident is not None and resident.offer((batch_id, rate), d_wave, soff)):
            d_wave.free(synced=True)
        groups_state.append(st)

    # ---- CMVN: statistics of every utterance in one launch per sample rate, summed per speaker (or
    # kept per utterance) on the host in utterance order; one apply launch per sample rate -----------
    if with_cmvn:
        dims = set(st['dim'] for st in groups_state)
        if len(dims) != 1:  # pragma: nocover (one processor, one dimension)
            raise ValueError('features have inconsistent dimensions')
        dim = dims.pop()
        if config['cmvn']['by_speaker']:
            names = list(dict.fromkeys(u.speaker for u in utts))
            number = {name: g for g, name in enumerate(names)}
            group_of = np.asarray([number[u.speaker] for u in utts], dtype=np.int32)
        else:
            names = [u.name for u in utts]
            group_of = np.arange(n, dtype=np.int32)
        cplan = _backend.get_plan(_abi.default_options(_abi.KIND_CMVN))
        per_utt = np.zeros((n, 2, dim + 1), dtype=np.float64)
        for st in groups_state:
            local = np.zeros((len(st['idx']), 2, dim + 1), dtype=np.float64)
            cplan.cmvn_accumulate_device(
                st['d_feat'].ptr, dim, st['foff'], local,
                d_weights=st['d_vad'].ptr if 'd_vad' in st else None,
                groups=np.arange(len(st['idx']), dtype=np.int32))
            per_utt[st['idx']] = local
            if 'd_vad' in st:
                st['d_vad'].free(synced=True)
        if stats_only:
            for st in groups_state:
                st['d_feat'].free(synced=True)
            return [names[g] for g in group_of], per_utt
        stats = np.zeros((len(names), 2, dim + 1), dtype=np.float64)
        for i in range(n):
            stats[group_of[i]] += per_utt[i]
        if stats_hook is not None:
            # several processes share the utterances of a speaker: their partial statistics are
            # summed here (shennong_amd.distributed.extract_features_sharded)
            stats = stats_hook(names, stats)
        for g in range(len(names)):
            if stats[g, 0, -1] < 1.0:
                raise ValueError(
                    'insufficient accumulation of stats for CMVN, '
                    'must be >= 1.0 but is {}'.format(stats[g, 0, -1]))
        for st in groups_state:
            d_out = DB(max(int(st['foff'][-1]) * dim * 4, 16))
            cplan.cmvn_apply_device(
                st['d_feat'].ptr, dim, st['foff'], stats, d_out.ptr,
                groups=group_of[st['idx']], norm_vars=True)
            st['d_feat'].free(synced=True)
            st['d_feat'] = d_out
        step = {}  # (the utterances of a group that share a _Meta take this step once)
        for i, g in enumerate(group_of.tolist()):
            m = step.get((meta[i], g))
            if m is None:
                m = step[(meta[i], g)] = meta[i].derive(
                    cache, ('cmvn', g),
                    lambda m, g=g: CmvnPostProcessor(dim, stats=stats[g]).get_properties(m))
            meta[i] = m

    # ---- delta ----------------------------------------------------------------------------------------
    if 'delta' in config:
        delta = _processor_class('delta')(**config['delta'])
        dplan = _backend.get_plan(delta._build_options())
        for st in groups_state:
            odim = dplan.post_ndims(st['dim'])
            d_out = DB(max(int(st['foff'][-1]) * odim * 4, 16))
            dplan.run_post_device(st['d_feat'].ptr, st['dim'], st['foff'], d_out.ptr)
            st['d_feat'].free(synced=True)
            st['d_feat'], st['dim'] = d_out, odim
            step = {}
            for i in st['idx']:
                m = step.get(meta[i])
                if m is None:
                    m = step[meta[i]] = meta[i].derive(cache, 'delta', delta.get_properties, ndims=odim)
                meta[i] = m

    # ---- pitch columns (the number of frames can differ by a few because of the downsampling in the
    # pitch tracker: same tolerance as Kaldi's paste-feats), then the only device -> host copy -----------
    out = FeaturesCollection()
    results = [None] * n
    pending = []
    for st in groups_state:
        idx = st['idx']
        if 'pitch_job' in st:
            try:
                st['d_pitch'] = st.pop('pitch_job').result()
            except BaseException:
                st['d_wave'].free()   # (a tracker that failed half way may still have readers enqueued)
                raise
            st['d_wave'].free(synced=True)
            rows, step, trims = [], {}, {}
            for i in idx:
                hit = step.get((meta[i], pmeta[i]))
                if hit is None:  # (same frame counts, times and histories: trimmed and merged once)
                    # (the trimmed frame count and times depend on the two time axes only: one comparison per
                    # pair of axes, not one per speaker)
                    tkey = (meta[i].nframes, id(meta[i].times), pmeta[i].nframes, id(pmeta[i].times))
                    trim = trims.get(tkey)
                    if trim is None:
                        trim = trims[tkey] = Features._concatenate_meta(
                            meta[i].nframes, meta[i].ndims, meta[i].times, {},
                            pmeta[i].nframes, pmeta[i].times, {}, tolerance, log)[:2]
                    r, times = trim
                    hit = step[(meta[i], pmeta[i])] = (r, meta[i].derive(
                        cache, ('concat', pmeta[i].key),
                        lambda m, o=pmeta[i]: Features._concatenate_meta(
                            1, m.ndims, m.times[:1], m.properties, 1, m.times[:1], o.properties,
                            tolerance, log)[2],
                        ndims=meta[i].ndims + pmeta[i].ndims, nframes=r, times=times))
                rows.append(hit[0])
                meta[i] = hit[1]
            ooff = offsets(rows)
            odim = st['dim'] + st['pdim']
            d_out = DB(max(int(ooff[-1]) * odim * 4, 16))
            _backend.concat_columns_device(
                st['d_feat'].ptr, st['dim'], st['foff'], st['d_pitch'].ptr, st['pdim'],
                st['pfoff'], d_out.ptr, ooff)
            st['d_feat'].free(synced=True)
            st['d_pitch'].free(synced=True)
            st['d_feat'], st['dim'], st['foff'] = d_out, odim, ooff
        if device_out is not None:
            host = np.empty((int(st['foff'][-1]), st['dim']), dtype=np.float32)   # (never written, never read)
            if host.size:
                _backend.check_finite_device(st['d_feat'].ptr, host.size)
            device_out.append((st['d_feat'], [utts[i].name for i in idx], st['dim']))
        else:
            host = _backend.result_array((int(st['foff'][-1]), st['dim']), np.float32)
        if device_out is not None:
            pass
        elif host.size:
            # (Features.validate's data check, once for the batch and before it leaves HBM; the copy then
            # runs while the per-utterance objects below are made - they only need to know WHERE their rows
            # will be)
            _backend.check_finite_device(st['d_feat'].ptr, host.size)
            pending.append((st['d_feat'].download_async(host), st['d_feat']))
        else:
            st['d_feat'].free(synced=True)
        cuts = st['foff'].tolist()
        for k, i in enumerate(idx):
            results[i] = host[cuts[k]:cuts[k + 1]]  # views of the one downloaded array
    of_batch = Features._of_batch
    for i, utt in enumerate(utts):
        # what is this utterance's own; the processors' part of the properties and the times are shared by
        # every utterance with the same history / frame count and copied when first read (Features._of_batch)
        audio = {'file': (os.path.abspath(utt.audio_file) if isinstance(utt.audio_file, str) else None),
                 'sample_rate': meta_of[i].sample_rate}
        if utt.tstart is not None:
            audio['tstart'] = utt.tstart
            audio['tstop'] = utt.tstop
        audio['duration'] = utt.duration
        extra = {'audio': audio, 'speaker': utt.speaker} if utt.speaker else {'audio': audio}
        # (times are generated, hence sorted; the data were checked above: no per-utterance validate)
        out[utt.name] = of_batch(results[i], meta[i].times, meta[i], extra)
    for wait, d_feat in pending:
        wait()
        d_feat.free(synced=True)
    return out


def _extract_features_by_stage(config, utterances, warps, log, tolerance=2):
    """The same pipeline with every stage going through the host-pointer entry points of the
    processors (kept as the step-by-step cross-check of the device-resident path)"""
    features_name = [k for k in config.keys() if k in valid_features()][0]
    with_cmvn = 'cmvn' in config
    if with_cmvn and config['cmvn']['by_speaker'] and not utterances.has_speakers():
        raise ValueError(
            'cmvn normalization by speaker requested '
            'but no speaker information provided')

    utts = list(utterances)
    metadata = {}
    for utt in utts:
        key = id(utt.audio_file) if not isinstance(utt.audio_file, str) else utt.audio_file
        if key not in metadata:
            from shennong_amd.audio import Audio
            metadata[key] = Audio.scan(utt.audio_file)
    meta_of = [metadata[id(u.audio_file) if not isinstance(u.audio_file, str) else u.audio_file]
               for u in utts]
    speakers = ('' if not utterances.has_speakers() else ' from {} speakers'.format(
        len(set(u.speaker for u in utts))))
    import datetime
    log.info('get %s utterances%s in %s audio files, total duration: %s',
             len(utts), speakers, len(metadata),
             datetime.timedelta(seconds=utterances.duration()))
    if not all(m.nchannels == 1 for m in meta_of):
        raise ValueError('all audio files are not mono')
    samplerates = sorted(set(m.sample_rate for m in meta_of))
    if len(samplerates) > 1:
        log.warning(
            'several sample rates found in audio files: %s, features '
            'extraction pipeline will work but this may not be a good '
            'idea to work on heterogeneous data',
            ', '.join(str(s) + 'Hz' for s in samplerates))

    # ---- pass one: features, (energy -> VAD), pitch; one batched launch per stage and sample rate ----
    n = len(utts)
    audios = [u.load_audio() for u in utts]
    feats = [None] * n
    weights = [None] * n
    pitch = [None] * n
    frame_length = frame_shift = None
    for rate in samplerates:
        idx = [i for i in range(n) if meta_of[i].sample_rate == rate]
        group = [audios[i] for i in idx]
        proc = _processor_class(features_name)(**config[features_name])
        proc.sample_rate = rate
        if frame_length is None:
            frame_length, frame_shift = proc.frame_length, proc.frame_shift
        log.debug('extract %s on %d utterances at %d Hz', features_name, len(idx), rate)
        if warps:
            out = proc._process_batch(group, vtln_warp=[warps[utts[i].name] for i in idx])
        else:
            out = proc._process_batch(group)
        for i, f in zip(idx, out):
            feats[i] = f

        if with_cmvn and config['cmvn']['with_vad']:
            energy = _processor_class('energy')()
            energy.frame_length = frame_length
            energy.frame_shift = frame_shift
            energy.sample_rate = rate
            vad = _processor_class('vad')(**config['cmvn']['vad'])
            decisions = vad._process_batch(energy._process_batch(group))
            for i, v in zip(idx, decisions):
                weights[i] = v.data.reshape((v.shape[0], ))

        if 'pitch' in config:
            params = {k: v for k, v in config['pitch'].items()
                      if k not in ('processor', 'postprocessing')}
            params['sample_rate'] = rate
            params['frame_shift'] = frame_shift
            params['frame_length'] = frame_length
            raw = _processor_class('kaldi_pitch')(**params)._process_batch(group)
            post = _processor_class('kaldi_pitch_post')(
                **config['pitch']['postprocessing'])._process_batch(raw)
            for i, p in zip(idx, post):
                pitch[i] = p

    for i, utt in enumerate(utts):
        props = feats[i].properties
        if utt.speaker:
            props['speaker'] = utt.speaker
        props['audio'] = {
            'file': (os.path.abspath(utt.audio_file)
                     if isinstance(utt.audio_file, str) else None),
            'sample_rate': meta_of[i].sample_rate}
        if utt.tstart is not None:
            props['audio']['tstart'] = utt.tstart
            props['audio']['tstop'] = utt.tstop
        props['audio']['duration'] = utt.duration

    # ---- CMVN: statistics of every utterance in one launch, summed per speaker (or kept per
    # utterance) in utterance order; one apply launch ------------------------------------------------
    if with_cmvn:
        dims = set(f.ndims for f in feats)
        if len(dims) != 1:  # pragma: nocover (one processor, one dimension)
            raise ValueError('features have inconsistent dimensions')
        dim = dims.pop()
        if config['cmvn']['by_speaker']:
            names = list(dict.fromkeys(u.speaker for u in utts))
            groups = np.asarray([names.index(u.speaker) for u in utts], dtype=np.int32)
        else:
            names = [u.name for u in utts]
            groups = np.arange(n, dtype=np.int32)
        for i, f in enumerate(feats):
            if weights[i] is not None and weights[i].shape[0] != f.nframes:
                raise ValueError(
                    'there is {} weights but {} feature frames, must be equal'
                    .format(weights[i].shape[0], f.nframes))
        plan = _backend.get_plan(_abi.default_options(_abi.KIND_CMVN))
        mats = [np.asarray(f.data, dtype=np.float32) for f in feats]
        stats = np.zeros((len(names), 2, dim + 1), dtype=np.float64)
        use_weights = config['cmvn']['with_vad']
        plan.cmvn_accumulate(
            mats, stats, groups=groups,
            weights=[w.astype(np.float32) for w in weights] if use_weights else None)
        for g in range(len(names)):
            if stats[g, 0, -1] < 1.0:
                raise ValueError(
                    'insufficient accumulation of stats for CMVN, '
                    'must be >= 1.0 but is {}'.format(stats[g, 0, -1]))
        datas = plan.cmvn_apply(mats, stats, groups=groups, norm_vars=True)
        for i, f in enumerate(feats):
            cmvn = CmvnPostProcessor(dim, stats=stats[groups[i]])
            feats[i] = Features(datas[i], f.times, properties=cmvn.get_properties(f))

    # ---- delta: one launch; pitch concatenation on the host ------------------------------------------
    if 'delta' in config:
        feats = _processor_class('delta')(**config['delta'])._process_batch(feats)

    out = FeaturesCollection()
    for i, utt in enumerate(utts):
        f = feats[i]
        if pitch[i] is not None:
            # the number of frames can differ by a few because of the downsampling in the pitch
            # tracker (same tolerance as Kaldi's paste-feats)
            f = f.concatenate(pitch[i], tolerance=tolerance, log=log)
        out[utt.name] = f
    return out
