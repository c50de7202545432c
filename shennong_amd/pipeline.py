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
import time

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
    if _too_large_for_one_batch(utterances):
        # the corpus in one launch per stage would not fit the HBM that is free (the pitch tracker alone keeps
        # 2.3 GB of scratch per hour of audio): the same features - bit for bit, see extract_features_streamed -
        # from bounded batches, collected here (the reference, on the CPU, takes a corpus of any size)
        log.info('the corpus does not fit one device batch: extracting in batches')
        collected = FeaturesCollection()
        extract_features_streamed(config, utterances, collected.update, warps=warps, njobs=min(njobs, 2), log=log)
        return collected
    if warps:
        warps = _init_warps(warps, config, utterances, log)
    return _extract_features(config, _view_of(utterances), warps, log)


def _too_large_for_one_batch(utterances):
    """True when one batch over all of `utterances` would need more than half of the HBM that is free
    (_BATCH_BYTES_PER_HOUR per hour of audio while a batch is in flight)"""
    if _backend.device_count() < 1:   # (host-logic tests with the device pipeline replaced by a stand-in)
        return False
    need = utterances.duration() / 3600.0 * _BATCH_BYTES_PER_HOUR
    if need < (8 << 30):              # (no query for what certainly fits)
        return False
    free, _ = _backend.mem_info()
    return need > free // 2


def _view_of(utterances):
    """`utterances` itself, or - for a pinned index (``Utterances.pin()``) - the view that tells the pipeline
    where its audio lies in the page-locked block"""
    pinned = getattr(utterances, '_pinned', None)
    if pinned is None:
        return utterances
    return _BatchView(list(utterances), utterances.has_speakers(), pinned=pinned, first=0)


def extract_features_warp(configuration, utterances, warp, log=get_logger('pipeline', 'warning'), njobs=1):
    """Speech features extraction pipeline when all features are warped by the same factor

    What the reference's VTLN trainer calls between its iterations (reference pipeline.py:650-696): the main
    features of every utterance with ``vtln_warp=warp`` and, when the configuration has a 'delta' entry, their
    deltas - no pitch, no CMVN, and properties without the per-utterance 'audio' / 'speaker' entries, exactly
    what ``delta.process(features.process(audio, vtln_warp=warp))`` returns.  One batched launch per stage
    instead of a thread pool; `njobs` is validated like the reference's.

    Raises ValueError for an invalid configuration and for spectrogram features (no VTLN there)."""
    get_njobs(njobs, log=log)
    config = _init_config(configuration, log=log)
    features = [k for k in config.keys() if k in valid_features()][0]
    if features == 'spectrogram':
        raise ValueError(f'{features} features do not support VTLN')
    warps = {utt.name: float(warp) for utt in utterances}
    return _extract_features(config, _view_of(utterances), warps, log, stages=('delta',),
                             utterance_properties=False)


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


def _in_flight(batches, work, depth, deferred=False):
    """``work(b, batch)`` for every batch, results in order, at most `depth` batches started and not yet
    handed over.  With `depth` > 1 the batches run on threads: the staging copy, the transfers and the
    launches of one (all outside the interpreter lock) overlap the per-utterance bookkeeping of another.
    `deferred`: `work` returns ``(result, finish)`` with the copy of the result still on its way
    (:func:`_extract_features` with ``defer=True``); `finish()` is called here, as late as the order allows -
    with one batch at a time, after the NEXT batch's work, so that the copy of batch k crosses the link beside
    the launches and the bookkeeping of batch k + 1 instead of being waited for.
    Closing the generator early (an error in the consumer) cancels what has not started and WAITS for what
    is running and for the copies on their way: nothing touches the caller's buffers after it returns."""
    from collections import deque
    done = deque()       # deferred results whose copy has not been waited for

    def hand_over(item):
        if not deferred:
            return item
        result, finish = item
        finish()
        return result

    try:
        if depth <= 1:
            for b, batch in enumerate(batches):
                item = work(b, batch)
                if not deferred:
                    yield item
                    continue
                done.append(item)
                if len(done) > 1:
                    yield hand_over(done.popleft())
            while done:
                yield hand_over(done.popleft())
            return
        from concurrent.futures import wait
        pending = deque()
        pool = _batch_pool()
        try:
            for b, batch in enumerate(batches):
                pending.append(pool.submit(work, b, batch))
                if len(pending) >= depth:
                    yield hand_over(pending.popleft().result())
            while pending:
                yield hand_over(pending.popleft().result())
        finally:
            for future in pending:
                future.cancel()
            wait(list(pending))
            if deferred:
                for future in pending:
                    if not future.cancelled() and future.exception() is None:
                        done.append(future.result())
    finally:
        while done:   # (the run ended early: the copies must have landed before their blocks go back to the pools)
            try:
                done.popleft()[1]()
            except Exception:  # pragma: nocover
                pass


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
                              log=get_logger('pipeline', 'warning'), stats=None):
    """:func:`extract_features` for a corpus that must not sit in memory at once (BASELINE config 5)

    The utterances are processed in consecutive batches of at most `max_batch_duration` seconds of
    audio (None: :func:`default_batch_duration` - four hours unless `njobs` batches of that size in flight
    would need more than half of the free HBM; one hour of 16 kHz audio is 115 MB of int16 up and, for 123
    columns, 177 MB of float32 down and needs ~4 GB of HBM while in flight; four hours were measured on an 8 h corpus of 3 s utterances: 1 h batches 54-57 hours of
    audio per second, 2 h 54-65, 4 h 59-64, the whole 8 h in one batch 70 - the pitch tracker's cost per utterance
    halves between 1 000 and 4 000 utterances per call; the results of a batch must fit a pooled page-locked block
    - two of them are alive at a time, 4 GiB in all (_backend._ResultBlock._FRESH): 4 h of 257 columns are 1.5 GB);
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

    A pinned index (``utterances.pin()``: the audio loaded once into ONE page-locked int16 block) is taken
    straight from that block: a batch is a stretch of it, uploaded from where it lies - no per-utterance load,
    conversion, check or gather (what bounded this function before round 6: the host, not the link).
    `stats` (a :class:`RunStats`): link bytes, transfer waits and kernel milliseconds summed over the run.

    Returns the number of utterances written."""
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
    has_speakers, pinned = utterances.has_speakers(), getattr(utterances, '_pinned', None)
    all_names, all_speakers = [u.name for u in utts], [u.speaker for u in utts]

    all_batches = list(_batches(utts, max_batch_duration))   # (the same cuts in both passes: made once)

    def views():
        """the batches, in order; the audio of a pinned batch that is not in HBM from an earlier pass is sent
        ahead: started when the batch BEFORE it is handed out, so that it crosses the link beside that one's
        kernels"""
        def view_of(batch, first, b):
            last = first + len(batch)
            view = _BatchView(batch, has_speakers, pinned=pinned, first=first, names=all_names[first:last],
                              speakers=all_speakers[first:last])
            if pinned is not None and not (resident is not None and resident.holds((b, pinned.sample_rate))):
                view.prefetch = _Prefetch(view)
            return view
        batches = iter(all_batches)
        ahead = first = None
        try:
            batch, first, b = next(batches, None), 0, 0
            ahead = view_of(batch, first, b) if batch is not None else None
            while ahead is not None:
                current, ahead = ahead, None
                first, b = first + len(current), b + 1
                batch = next(batches, None)
                if batch is not None:
                    ahead = view_of(batch, first, b)
                yield current
        finally:
            if ahead is not None and ahead.prefetch is not None:   # (the run ended before this batch was handed out)
                ahead.prefetch.cancel()

    def sub(batch):
        return {u.name: warps[u.name] for u in batch} if warps else None

    hook = None
    resident = _ResidentWaves(resident_bytes) if by_speaker and resident_bytes > 0 else None
    # The cyclic collector stays paused for the whole run, not only inside the batches: re-enabled after a batch,
    # its first pass walks the ~100 000 objects that batch just made - all alive: 5-11 ms per batch, inside
    # whatever allocates next (the sink).  The batch's objects go when the sink has returned (reference counts);
    # every eighth batch the young generations are collected by hand, so that cyclic garbage of the sink's own
    # making does not pile up over a long corpus.
    from shennong_amd.utils import paused_gc
    gc_pause = paused_gc().__enter__()
    running = None   # the generator of the batches in flight: closed (its threads joined) before the audio
                     # buffers are released, whatever ends the pass - the end of the corpus, an error in a
                     # batch, an exception from `sink`
    try:
        if by_speaker:
            number, table = {}, [None]   # speaker -> row of table[0], float64 [speakers, 2, dim + 1]

            def first_pass(b, batch):
                return _extract_features(config, batch, sub(batch), log, stats_only=True,
                                         resident=resident, batch_id=b, stats=stats)

            running = _in_flight(views(), first_pass, depth)
            for speakers, per_utt in running:
                # (summed in utterance order like the one-shot pipeline: np.add.at adds row after row)
                for speaker in speakers:
                    if speaker not in number:
                        number[speaker] = len(number)
                per_utt = np.asarray(per_utt)
                if table[0] is None or len(number) > table[0].shape[0]:
                    grown = np.zeros((max(2 * len(number), 64),) + per_utt.shape[1:], dtype=np.float64)
                    if table[0] is not None:
                        grown[:table[0].shape[0]] = table[0]
                    table[0] = grown
                np.add.at(table[0], np.fromiter((number[s] for s in speakers), np.int64, len(speakers)), per_utt)
            running = None
            names = list(number)
            total_of = table[0][:len(names)] if names else np.zeros((0, 2, 1), dtype=np.float64)
            if stats_reduce is not None:
                total_of = np.asarray(stats_reduce(names, np.ascontiguousarray(total_of)))

            def hook(batch_names, _partial):
                return total_of[np.fromiter((number[k] for k in batch_names), np.int64, len(batch_names))]
            hook.replaces = True   # (the batch's own partial sums are not looked at: not made either)

        def second_pass(b, batch):
            return _extract_features(config, batch, sub(batch), log, stats_hook=hook,
                                     resident=resident, batch_id=b, stats=stats, defer=True)

        count = handed = 0
        batches = running = _in_flight(views(), second_pass, depth, deferred=True)
        while True:
            # (nothing of batch k is referenced here while batch k + 1 is made: its page-locked result block
            # is back in the pool by then, see _backend.result_array)
            features = next(batches, None)
            if features is None:
                break
            sink(features)
            count += len(features)
            del features
            handed += 1
            if handed % 8 == 0:
                gc_pause.collect_young()
        return count
    finally:
        try:
            if running is not None:
                running.close()   # (GeneratorExit inside _in_flight: pending batches cancelled, its pool joined)
            if resident is not None:
                resident.clear()
        finally:
            gc_pause.__exit__(None, None, None)


# The device-resident pipeline draws its random terms (dither, delta-pitch noise) from ONE named noise call
# (snf_set_noise_call): the features of an utterance are a function of the configuration and the utterance
# alone - the same in the statistics pass and the apply pass of the streamed pipeline, in any batch split,
# in every run (the reference, whose dither comes from Kaldi's global rand(), has none of these).
_NOISE_CALL = int(os.environ.get('SNF_NOISE_CALL', '1'))   # (another run of the same corpus with other noise: export another number)


class _Meta:
    """What the post-processors' `get_properties` need to know about features that live in HBM.
    The properties of a stage are the same for every utterance that went through the same processors
    with the same per-utterance arguments (warp factor, CMVN group): `key` names that history and
    `cache` holds one properties dictionary per history, copied once per utterance at the end.
    A _Meta points to its parent only (no table of children: parent and child would form a reference cycle,
    and the 3 000 of them a by-speaker batch makes would wait for the cyclic collector instead of going with
    the batch)."""
    __slots__ = ('_properties', '_source', 'ndims', 'nframes', 'times', 'key', '_json')

    def __init__(self, properties, ndims, nframes, times, key=None, source=None):
        self._properties = properties
        self._source = source      # (cache, parent, make_properties) while the properties are not made yet
        self.ndims = ndims
        self.nframes = nframes
        self.times = times
        self.key = key
        self._json = None

    def json(self, dumps):
        """`properties` as JSON text, encoded once (serializers: Features._json_properties)"""
        if self._json is None:
            self._json = dumps(self.properties)
        return self._json

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
        """The _Meta after one more stage (the stages call this once per distinct history and frame count,
        see _classes_of)"""
        key = (self.key, tag)
        return _Meta(cache.get(key), self.ndims if ndims is None else ndims,
                     self.nframes if nframes is None else nframes,
                     self.times if times is None else times, key, source=(cache, self, make_properties))


_TRACK_IN_FIRST_PASS = os.environ.get('SNF_TRACK_IN_FIRST_PASS', '1') != '0'   # (developer knob: A/B runs)


class _ResidentWaves:
    """Uploaded waveforms kept in HBM between the two passes of :func:`extract_features_streamed`
    (96 kB per 3 s utterance: 16 GiB hold 140 hours of 16 kHz audio), so that the second pass neither reads
    the audio files nor crosses the host link again.  Batches that do not fit the budget are re-uploaded."""
    def __init__(self, budget):
        self.budget = int(budget)
        self.held = 0
        self._items = {}
        self._tracks = {}   # pitch trackers started on resident audio in the first pass (see offer_track)
        self._lock = threading.Lock()  # (the batches in flight run on threads)

    def offer(self, key, d_wave, soff):
        with self._lock:
            if self.held + d_wave.nbytes > self.budget:
                return False
            self._items[key] = (d_wave, soff)
            self.held += d_wave.nbytes
            return True

    def holds(self, key):
        with self._lock:
            return key in self._items

    def take(self, key):
        with self._lock:
            item = self._items.pop(key, None)
            if item is not None:
                self.held -= item[0].nbytes
            return item

    def offer_track(self, key, job):
        """The pitch tracker of a resident batch, started in the FIRST pass - which is bound by the uploads and
        leaves the GPU idle - so that the second pass, bound by the downloads, finds the pitch columns made
        (3 floats per frame: 0.5 GB per 125 h, not counted against the budget).  `job.result()` is the device
        block; the audio it reads is the resident block of the same key."""
        with self._lock:
            self._tracks[key] = job

    def take_track(self, key):
        with self._lock:
            return self._tracks.pop(key, None)

    def clear(self):
        with self._lock:   # (a batch thread that is still running may offer / take meanwhile)
            items = list(self._items.values())
            tracks = list(self._tracks.values())
            self._items.clear()
            self._tracks.clear()
            self.held = 0
        for job in tracks:   # (before the audio goes: a tracker that still runs reads it)
            try:
                job.result().free()
            except BaseException:  # pragma: nocover  (the tracker failed: it gave its blocks back itself)
                pass
        for d_wave, _ in items:
            d_wave.free(synced=True)


class _BatchView:
    """A run of utterances of an index as the device pipeline needs it: iteration, `has_speakers`, `duration`
    - what ``Utterances(batch)`` gave, minus its validation (names, formats, duplicates: the index went through
    it when it was made; 4 us per utterance and batch) - plus, when the index was pinned (``Utterances.pin()``),
    WHERE the audio of the run lies in the page-locked block: utterances `first` ... `first + len` of `pinned`."""
    def __init__(self, utts, has_speakers, pinned=None, first=0, names=None, speakers=None):
        self._utts = utts
        self._has_speakers = bool(has_speakers)
        self.pinned, self.first = pinned, int(first)
        self.prefetch = None   # a _Prefetch when the audio of the run was sent ahead
        # (read once per corpus instead of once per utterance, batch and pass)
        self.names = names if names is not None else [u.name for u in utts]
        self.speakers = speakers if speakers is not None else [u.speaker for u in utts]

    def __iter__(self):
        return iter(self._utts)

    def __len__(self):
        return len(self._utts)

    def has_speakers(self):
        return self._has_speakers

    def duration(self):
        return sum(u.duration for u in self._utts)


class _Prefetch:
    """The audio of a pinned view on its way to HBM before its batch starts: the upload of batch k + 1 runs
    beside the kernels of batch k (the 'double-buffered chunks' of SURVEY.md 8d, config 5).  Started on the
    thread that walks the batches, taken (waited for) by the thread that runs the batch."""
    def __init__(self, view):
        pinned, a, n = view.pinned, view.first, len(view)
        s0, s1 = int(pinned.soff[a]), int(pinned.soff[a + n])
        self.soff = pinned.soff[a:a + n + 1] - s0
        self.nbytes = 2 * (s1 - s0)
        self.block = _backend.DeviceBuffer(max(self.nbytes, 16))
        try:
            self._wait = self.block.upload_async(pinned.block[s0:s1])
        except BaseException:
            self.block.free()
            raise

    def take(self):
        """(block, sample offsets); the caller owns the block"""
        block, self.block = self.block, None
        try:
            self._wait()
        except BaseException:
            block.free()
            raise
        return block, self.soff

    def cancel(self):
        if self.block is not None:
            block, self.block = self.block, None
            try:
                self._wait()
            finally:
                block.free()

    def __del__(self):   # (a view that was handed out and never ran: an error ended its pass)
        try:
            self.cancel()
        except Exception:  # pragma: nocover
            pass


class _Tracker:
    """The pitch tracker (+ its post-processing) of one block of audio that is in HBM, started on a side thread:
    it needs nothing but the audio and runs beside the features -> VAD -> CMVN -> delta chain of its batch.
    `job.result()` is the device block of the post-processed pitch (the caller owns it); the audio block must
    outlive the job.  (Round 6 also started the tracker of batch k + 1 of the second streamed pass when batch k
    begins, its audio being resident: 1.158 against 1.179 s per 125 h - the per-batch chain is bound by the
    download and the host work around it, not by the wait for the tracker; not kept.)"""
    def __init__(self, config, rate, frame_shift, frame_length, d_wave, soff, stats=None, start=True):
        DB = _backend.DeviceBuffer
        params = {k: v for k, v in config['pitch'].items() if k not in ('processor', 'postprocessing')}
        params['sample_rate'] = rate
        params['frame_shift'] = frame_shift
        params['frame_length'] = frame_length
        self.rate, self.soff = rate, soff
        self.pproc = pproc = _processor_class('kaldi_pitch')(**params)
        self.post = post = _processor_class('kaldi_pitch_post')(**config['pitch']['postprocessing'])
        pplan = _backend.get_plan(pproc._build_options())
        self.pfoff = pfoff = _PipelineRun._frame_offsets(pplan, soff)
        qplan = _backend.get_plan(post._build_options())
        self.pdim = pdim = qplan.post_ndims(2)

        def track():
            d_raw = d_pitch = None
            try:
                d_raw = DB(max(int(pfoff[-1]) * 2 * 4, 16))
                pplan.run_device(d_wave, soff, pfoff, d_raw.ptr)
                d_pitch = DB(max(int(pfoff[-1]) * pdim * 4, 16))
                qplan.run_post_device(d_raw.ptr, 2, pfoff, d_pitch.ptr, noise_call=_NOISE_CALL)
                if stats is not None:
                    stats.add(gpu_ms=_call_ms(pplan) + _call_ms(qplan))
            except BaseException:
                # a call that failed half way may have kernels enqueued that still write these blocks:
                # the plain free() waits for the device before the pool can hand them to another thread
                for block in (d_pitch, d_raw):
                    if block is not None:
                        block.free()
                raise
            d_raw.free(synced=True)   # (both calls returned: their streams are synchronised)
            return d_pitch

        self.job = _backend.side_pool().submit(track) if start else None   # (not started: the description only)


class RunStats:
    """What a pipeline run cost, summed over its batches and threads (pass one as ``stats=`` to
    :func:`extract_features_streamed`): bytes over the host link in each direction, seconds this process
    waited for uploads / downloads, milliseconds of kernels (HIP events of every plan call).  bench.py prints
    them beside the wall clock: the link floor and the GPU time of BASELINE config 5."""
    _FIELDS = ('bytes_up', 'bytes_down', 'upload_wait_s', 'download_wait_s', 'gpu_ms', 'batches', 'utterances')

    def __init__(self):
        self._lock = threading.Lock()
        for name in self._FIELDS:
            setattr(self, name, 0)

    def add(self, **amounts):
        with self._lock:
            for name, value in amounts.items():
                setattr(self, name, getattr(self, name) + value)

    def as_dict(self):
        return {name: getattr(self, name) for name in self._FIELDS}


def _utterance_properties(utt, sample_rate):
    """The entries of an utterance's properties that are its own (reference pipeline.py:598-611), made when the
    properties are first read (Features._of_batch)"""
    audio = {'file': (os.path.abspath(utt.audio_file) if isinstance(utt.audio_file, str) else None),
             'sample_rate': sample_rate}
    if utt.tstart is not None:
        audio['tstart'] = utt.tstart
        audio['tstop'] = utt.tstop
    audio['duration'] = utt.duration
    return {'audio': audio, 'speaker': utt.speaker} if utt.speaker else {'audio': audio}


class _Group:
    """The utterances of one sample rate of a batch and the device blocks they own while they go through the
    stages.  A block has ONE owner at any time - this table, the tracker's thread, or the _ResidentWaves - and
    one place where it is given back: `swap` / `drop` (the launches that used it have been waited for: no device
    wait), or `abort` (an error: something may still be enqueued, plain free() waits for the device)."""
    def __init__(self, rate, idx):
        self.rate, self.idx = rate, idx
        self.blocks = {}
        self.soff = self.foff = self.pfoff = None
        self.dim = self.pdim = 0
        self.pitch_job = None

    def hold(self, name, block):
        assert name not in self.blocks
        self.blocks[name] = block
        return block

    def ptr(self, name):
        return self.blocks[name].ptr

    def swap(self, name, block):
        """`block` takes the place of the block held under `name`, which is given back"""
        self.blocks.pop(name).free(synced=True)
        self.blocks[name] = block

    def drop(self, name):
        self.blocks.pop(name).free(synced=True)

    def give(self, name):
        """hands the block to another owner"""
        return self.blocks.pop(name)

    def abort(self):
        if self.pitch_job is not None:   # (the tracker reads the audio: first let it end, whatever its outcome)
            job, self.pitch_job = self.pitch_job, None
            try:
                self.blocks['pitch'] = job.result()
            except BaseException:   # noqa: BLE001 (its own blocks were released by the job)
                pass
        for name in list(self.blocks):
            self.blocks.pop(name).free()


def _call_ms(plan):
    """HIP-event time of the plan's last call for RunStats.  With several batches in flight on one cached plan the
    marks may already belong to the next call when they are read (-1, or a partial time): the sum is an estimate
    then, never negative"""
    return max(plan.last_kernel_ms(0), 0.0)


def _offsets(counts):
    off = np.zeros(len(counts) + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    return off


def _classes_of(*keys):
    """Utterances that agree in all of the non-negative integer arrays `keys` form a class: returns (the index of
    one member per class, the class number of every utterance) - np.unique over a mixed-radix code of the keys,
    instead of a dictionary lookup per utterance"""
    code = np.zeros(keys[0].shape[0], dtype=np.int64)
    for key in keys:
        key = np.asarray(key, dtype=np.int64)
        code = code * (int(key.max()) + 1 if key.size else 1) + key
    _, first, inverse = np.unique(code, return_index=True, return_inverse=True)
    return first, inverse


def _extract_features(config, utterances, warps, log, tolerance=2, stats_hook=None,
                      stats_only=False, resident=None, batch_id=None, device_out=None, stats=None,
                      stages=None, utterance_properties=True, defer=False):
    """`defer`: return ``(features, finish)`` with the copy of the result block still on its way; `finish()`
    waits for it and must be called before the data are read (extract_features_streamed: the copy of batch k
    crosses the link beside the launches of batch k + 1)"""
    from shennong_amd.utils import paused_gc
    with paused_gc():   # (thousands of small objects per batch, none of them garbage: see utils.paused_gc)
        run = _PipelineRun(config, utterances, warps, log, tolerance, resident, batch_id, stats, stages,
                           utterance_properties)
        try:
            return run.execute(stats_hook, stats_only, device_out, defer)
        except BaseException:
            run.abort()
            raise


class _PipelineRun:
    """One batch through the whole pipeline with the intermediate features resident in HBM: the waveforms go up
    once, every stage is one batched launch on device blocks (features, energy -> VAD, CMVN statistics and
    apply, delta, pitch and its post-processing, column concatenation), the final matrices come down once.
    Stage order, arithmetic and properties are those of reference pipeline.py:525-643; one method per stage:

        audio -> [pitch, started on a side thread] -> features -> [vad] -> cmvn -> delta -> join

    `stats_only` (first pass of :func:`extract_features_streamed`): stop after the CMVN accumulation and return
    ``(group name of every utterance, per-utterance statistics [n, 2, dim + 1])``; the pitch stage, which the
    statistics do not depend on, is skipped.  `resident` (a _ResidentWaves) keeps the uploaded waveforms of batch
    `batch_id` in HBM after that pass and hands them to the next one.  `device_out` (a list;
    :func:`shennong_amd.distributed.extract_features_sharded`): the final matrices STAY in HBM - one
    ``(DeviceBuffer [rows, ndims], names in row order, ndims)`` per sample rate is appended and the caller owns
    the blocks; the returned Features carry times and properties over data that were never downloaded.
    `stages`: None = what the configuration names; a subset (``extract_features_warp``: features and delta
    only).  Every launch goes through an entry point that synchronises its stream before it returns and the one
    asynchronous copy is waited for before its block is released: blocks are given back without the device-wide
    wait DeviceBuffer.free() otherwise makes (see _Group)."""
    def __init__(self, config, utterances, warps, log, tolerance, resident, batch_id, stats, stages,
                 utterance_properties):
        self.config, self.warps, self.log, self.tolerance = config, warps, log, tolerance
        self.resident, self.batch_id, self.stats = resident, batch_id, stats
        self.utterance_properties = utterance_properties
        self.features_name = [k for k in config.keys() if k in valid_features()][0]
        enabled = (lambda name: name in config) if stages is None else \
            (lambda name: name in config and name in stages)
        self.with_cmvn, self.with_delta, self.with_pitch = enabled('cmvn'), enabled('delta'), enabled('pitch')
        if self.with_cmvn and config['cmvn']['by_speaker'] and not utterances.has_speakers():
            raise ValueError(
                'cmvn normalization by speaker requested '
                'but no speaker information provided')
        self.view = utterances
        self.utts = list(utterances)
        self.n = len(self.utts)
        self.names = getattr(utterances, 'names', None) or [u.name for u in self.utts]
        self.speakers = getattr(utterances, 'speakers', None) or [u.speaker for u in self.utts]
        self.cache = {}       # properties per processing history (see _Meta)
        self.groups = []      # one _Group per sample rate
        self.classes = []     # the distinct _Meta of the main features ...
        self.cls_of = np.zeros(self.n, dtype=np.int64)   # ... and which of them every utterance has
        self.pclasses, self.pcls_of = [], np.zeros(self.n, dtype=np.int64)   # the same for the pitch features
        self.rate_of = [None] * self.n
        self.frame_length = self.frame_shift = None

    def abort(self):
        for group in self.groups:
            group.abort()

    def _count(self, **amounts):
        if self.stats is not None:
            self.stats.add(**amounts)

    def execute(self, stats_hook, stats_only, device_out, defer=False):
        self.stage_audio()
        for group in self.groups:
            proc = _processor_class(self.features_name)(**self.config[self.features_name])
            proc.sample_rate = group.rate
            if self.frame_length is None:
                self.frame_length, self.frame_shift = proc.frame_length, proc.frame_shift
            if self.with_pitch and not stats_only:
                self.stage_pitch_start(group)
            self.stage_features(group, proc)
            # (a hook that brings the statistics of the whole corpus - the second pass of the streamed pipeline -
            # makes this batch's own statistics, and the VAD weights that only they use, unnecessary)
            own_stats = self.with_cmvn and not getattr(stats_hook, 'replaces', False)
            if own_stats and self.config['cmvn']['with_vad']:
                self.stage_vad(group)
            if group.pitch_job is not None:
                pass   # (the tracker still reads the audio: released where it is waited for, stage_join)
            elif stats_only and self.resident is not None and self.resident.offer(
                    (self.batch_id, group.rate), group.blocks['wave'], group.soff):
                block = group.give('wave')
                if self.with_pitch and _TRACK_IN_FIRST_PASS:
                    # the tracker of this batch starts NOW, on the audio that stays in HBM: the first pass waits
                    # for uploads, the second for downloads - the GPU is idle in the first
                    self.resident.offer_track(
                        (self.batch_id, group.rate),
                        _Tracker(self.config, group.rate, self.frame_shift, self.frame_length, block.ptr,
                                 group.soff, self.stats).job)
            else:
                group.drop('wave')
        if self.with_cmvn:
            found = self.stage_cmvn(stats_hook, stats_only)
            if stats_only:
                return found
        if self.with_delta:
            self.stage_delta()
        return self.stage_join(device_out, defer)

    # ---- audio: one int16 block per sample rate in HBM ------------------------------------------------------
    def stage_audio(self):
        from shennong_amd.audio import Audio
        from shennong_amd.processor.base import check_signal
        utts, n, log = self.utts, self.n, self.log
        pinned = getattr(self.view, 'pinned', None)
        if pinned is not None:
            # the run's audio is one stretch of a page-locked block that was checked when it was made (mono,
            # one rate, int16): it goes up from where it is
            import logging
            if log.isEnabledFor(logging.INFO):
                self._log_summary(1)
            group = _Group(pinned.sample_rate, np.arange(n))
            self.rate_of = [pinned.sample_rate] * n
            self.groups.append(group)
            held = self.resident.take((self.batch_id, group.rate)) if self.resident is not None else None
            if held is not None:
                block, group.soff = held
                group.hold('wave', block)
                if self.view.prefetch is not None:
                    self.view.prefetch.cancel()
                return
            t0 = time.perf_counter()
            ahead, self.view.prefetch = self.view.prefetch or _Prefetch(self.view), None
            block, group.soff = ahead.take()
            group.hold('wave', block)
            self._count(bytes_up=ahead.nbytes, upload_wait_s=time.perf_counter() - t0)
            return
        metadata = {}
        meta_of = []
        for utt in utts:
            key = utt.audio_file if isinstance(utt.audio_file, str) else id(utt.audio_file)
            found = metadata.get(key)
            if found is None:   # (the index scanned the header of every file when it was made)
                found = metadata[key] = getattr(utt, '_scan', None) or Audio.scan(utt.audio_file)
            meta_of.append(found)
        self._log_summary(len(metadata))
        if not all(m.nchannels == 1 for m in meta_of):
            raise ValueError('all audio files are not mono')
        self.rate_of = [m.sample_rate for m in meta_of]
        samplerates = sorted(set(self.rate_of))
        if len(samplerates) > 1:
            log.warning(
                'several sample rates found in audio files: %s, features '
                'extraction pipeline will work but this may not be a good '
                'idea to work on heterogeneous data',
                ', '.join(str(s) + 'Hz' for s in samplerates))
        for rate in samplerates:
            idx = np.asarray([i for i in range(n) if self.rate_of[i] == rate], dtype=np.int64)
            group = _Group(rate, idx)
            self.groups.append(group)
            held = self.resident.take((self.batch_id, rate)) if self.resident is not None else None
            if held is not None:
                block, group.soff = held
                group.hold('wave', block)
                continue
            mine = [utts[i] for i in idx.tolist()]
            if all(isinstance(u.audio_file, str) for u in mine):
                # WAV files: read side by side straight into page-locked staging memory (16-bit mono PCM natively,
                # other sample types through Audio.load + astype), uploaded from there
                from shennong_amd.audio import load_int16_block, sample_range
                metas = [meta_of[i] for i in idx.tolist()]
                group.soff = _offsets([sample_range(m.nsamples, m.sample_rate, u.tstart, u.tstop)[1]
                                       for u, m in zip(mine, metas)])
                total = int(group.soff[-1])
                staged, token = _backend.STAGING.array((max(total, 1),), np.int16)
                try:
                    load_int16_block(mine, metas, staged, group.soff)
                    t0 = time.perf_counter()
                    group.hold('wave', _backend.DeviceBuffer(max(2 * total, 16))).upload(staged[:total])
                    self._count(bytes_up=2 * total, upload_wait_s=time.perf_counter() - t0)
                finally:
                    del staged
                    _backend.STAGING.release(token)
                continue
            proc = _processor_class(self.features_name)(**self.config[self.features_name])
            proc.sample_rate = rate
            waves, checked = [], set()
            for utt in mine:
                audio = utt.load_audio()
                if (audio.nchannels, audio.sample_rate) not in checked:  # (one check per kind of signal)
                    check_signal(proc, audio)
                    checked.add((audio.nchannels, audio.sample_rate))
                waves.append(audio.astype(np.int16).data)
            group.soff = _offsets([w.shape[0] for w in waves])
            t0 = time.perf_counter()
            group.hold('wave', _backend.upload_rows(waves, np.int16))  # (page-locked staging: full link rate)
            self._count(bytes_up=2 * int(group.soff[-1]), upload_wait_s=time.perf_counter() - t0)

    def _log_summary(self, nfiles):
        import datetime
        utts = self.utts
        speakers = ('' if not self.view.has_speakers() else ' from {} speakers'.format(
            len(set(u.speaker for u in utts))))
        self.log.info('get %s utterances%s in %s audio files, total duration: %s',
                      len(utts), speakers, nfiles, datetime.timedelta(seconds=self.view.duration()))

    @staticmethod
    def _frame_offsets(plan, soff):
        lengths = np.diff(soff)
        uniq, inverse = np.unique(lengths, return_inverse=True)
        frames = np.asarray([plan.num_frames(x) for x in uniq.tolist()], dtype=np.int64)
        return _offsets(frames[inverse])

    # ---- pitch: needs nothing but the audio, runs on a side thread (its own streams) while this one takes the
    # audio through the features, VAD, CMVN and delta; waited for where the columns are joined ------------------
    def stage_pitch_start(self, group):
        cache, rate = self.cache, group.rate
        made = self.resident.take_track((self.batch_id, rate)) if self.resident is not None else None
        tracker = _Tracker(self.config, rate, self.frame_shift, self.frame_length, group.ptr('wave'), group.soff,
                           self.stats, start=made is None)
        group.pfoff, group.pdim, group.pitch_job = tracker.pfoff, tracker.pdim, made or tracker.job
        pproc, post, pfoff, pdim = tracker.pproc, tracker.post, tracker.pfoff, tracker.pdim
        key = ('pitch', rate)
        if key not in cache:
            cache[key] = pproc.get_properties()
        counts = np.diff(pfoff)
        uniq, inverse = np.unique(counts, return_inverse=True)
        base = len(self.pclasses)
        for t in uniq.tolist():
            tkey = ('times', 'pitch', rate, t)
            if tkey not in cache:
                cache[tkey] = pproc.times(t)
            self.pclasses.append(_Meta(cache[key], 2, t, cache[tkey], key).derive(
                cache, 'post', post.get_properties, ndims=pdim))
        self.pcls_of[group.idx] = base + inverse

    # ---- the main features ------------------------------------------------------------------------------------
    def stage_features(self, group, proc):
        features_name, rate, cache, utts = self.features_name, group.rate, self.cache, self.utts
        plan = _backend.get_plan(proc._build_options())
        group.dim = dim = plan.ndims
        group.foff = foff = self._frame_offsets(plan, group.soff)
        group.hold('feat', _backend.DeviceBuffer(max(int(foff[-1]) * dim * 4, 16)))
        vt = wlist = None
        if self.warps and features_name != 'spectrogram':
            wlist = np.asarray([self.warps[utts[i].name] for i in group.idx.tolist()], dtype=np.float64)
            vt = wlist.astype(np.float32)
        self.log.debug('extract %s on %d utterances at %d Hz', features_name, len(group.idx), rate)
        plan.run_device(group.ptr('wave'), group.soff, foff, group.ptr('feat'), vtln_warps=vt,
                        noise_call=_NOISE_CALL)
        self._count(gpu_ms=_call_ms(plan))
        # one _Meta per distinct (warp, frame count)
        counts = np.diff(foff)
        if wlist is None:
            first, inverse = _classes_of(counts)
        else:   # (classes by the caller's floats, which the properties record - not by their float32 images)
            first, inverse = _classes_of(np.unique(wlist, return_inverse=True)[1], counts)
        base = len(self.classes)
        for k in first.tolist():
            t = int(counts[k])
            warp = None if features_name == 'spectrogram' else float(wlist[k]) if wlist is not None else 1.0
            key = (features_name, rate, warp)
            if key not in cache:
                cache[key] = proc.get_properties(**({} if warp is None else {'vtln_warp': warp}))
            tkey = ('times', features_name, rate, t)
            if tkey not in cache:
                cache[tkey] = proc.times(t)
            self.classes.append(_Meta(cache[key], dim, t, cache[tkey], key))
        self.cls_of[group.idx] = base + inverse

    # ---- energy -> VAD: the weights of the CMVN statistics --------------------------------------------------
    def stage_vad(self, group):
        DB = _backend.DeviceBuffer
        energy = _processor_class('energy')()
        energy.frame_length = self.frame_length
        energy.frame_shift = self.frame_shift
        energy.sample_rate = group.rate
        eplan = _backend.get_plan(energy._build_options())
        foff = group.foff
        if not np.array_equal(self._frame_offsets(eplan, group.soff), foff):
            raise ValueError('energy and features differ in number of frames')
        group.hold('energy', DB(max(int(foff[-1]) * 4, 16)))
        eplan.run_device(group.ptr('wave'), group.soff, foff, group.ptr('energy'), noise_call=_NOISE_CALL)
        vad = _processor_class('vad')(**self.config['cmvn']['vad'])
        vplan = _backend.get_plan(vad._build_options())
        group.hold('vad', DB(max(int(foff[-1]) * 4, 16)))
        vplan.run_post_device(group.ptr('energy'), 1, foff, group.ptr('vad'))
        self._count(gpu_ms=_call_ms(eplan) + _call_ms(vplan))
        group.drop('energy')

    # ---- CMVN: statistics of every utterance in one launch per sample rate, summed per speaker (or kept per
    # utterance) on the host in utterance order; one apply launch per sample rate ------------------------------
    def stage_cmvn(self, stats_hook, stats_only):
        config, n, cache = self.config, self.n, self.cache
        dims = set(group.dim for group in self.groups)
        if len(dims) != 1:  # pragma: nocover (one processor, one dimension)
            raise ValueError('features have inconsistent dimensions')
        dim = dims.pop()
        if config['cmvn']['by_speaker']:
            names = list(dict.fromkeys(self.speakers))
            number = {name: g for g, name in enumerate(names)}
            group_of = np.fromiter(map(number.__getitem__, self.speakers), np.int32, n)
        else:
            names = self.names
            group_of = np.arange(n, dtype=np.int32)
        cplan = _backend.get_plan(_abi.default_options(_abi.KIND_CMVN))
        replaced = getattr(stats_hook, 'replaces', False) and not stats_only
        per_utt = np.zeros((0 if replaced else n, 2, dim + 1), dtype=np.float64)
        for group in self.groups:
            if replaced:
                break
            local = np.zeros((len(group.idx), 2, dim + 1), dtype=np.float64)
            cplan.cmvn_accumulate_device(
                group.ptr('feat'), dim, group.foff, local,
                d_weights=group.ptr('vad') if 'vad' in group.blocks else None,
                groups=np.arange(len(group.idx), dtype=np.int32))
            per_utt[group.idx] = local
            if 'vad' in group.blocks:
                group.drop('vad')
        if stats_only:
            for group in self.groups:
                group.drop('feat')
            return [names[g] for g in group_of.tolist()], per_utt
        # (utterance order, like the reference's accumulate loop: np.add.at adds the rows one after the other; a
        # hook that brings the statistics of the whole corpus - second pass of the streamed pipeline - needs none)
        stats = np.zeros((len(names), 2, dim + 1), dtype=np.float64)
        if not replaced:
            np.add.at(stats, group_of, per_utt)
        if stats_hook is not None:
            # several processes share the utterances of a speaker: their partial statistics are
            # summed here (shennong_amd.distributed.extract_features_sharded)
            stats = stats_hook(names, stats)
        low = np.flatnonzero(stats[:, 0, -1] < 1.0)
        if low.size:
            raise ValueError(
                'insufficient accumulation of stats for CMVN, '
                'must be >= 1.0 but is {}'.format(stats[low[0], 0, -1]))
        for group in self.groups:
            d_out = _backend.DeviceBuffer(max(int(group.foff[-1]) * dim * 4, 16))
            cplan.cmvn_apply_device(
                group.ptr('feat'), dim, group.foff, stats, d_out.ptr,
                groups=group_of[group.idx], norm_vars=True)
            group.swap('feat', d_out)
        # (the utterances of a CMVN group that share a _Meta take this step once)
        first, inverse = _classes_of(self.cls_of, group_of)
        classes = []
        for k in first.tolist():
            g = int(group_of[k])
            classes.append(self.classes[int(self.cls_of[k])].derive(
                cache, ('cmvn', g),
                lambda m, g=g: CmvnPostProcessor(dim, stats=stats[g]).get_properties(m)))
        self.classes, self.cls_of = classes, inverse

    # ---- delta --------------------------------------------------------------------------------------------
    def stage_delta(self):
        delta = _processor_class('delta')(**self.config['delta'])
        dplan = _backend.get_plan(delta._build_options())
        odim = None
        for group in self.groups:
            odim = dplan.post_ndims(group.dim)
            d_out = _backend.DeviceBuffer(max(int(group.foff[-1]) * odim * 4, 16))
            dplan.run_post_device(group.ptr('feat'), group.dim, group.foff, d_out.ptr)
            self._count(gpu_ms=_call_ms(dplan))
            group.swap('feat', d_out)
            group.dim = odim
        self.classes = [m.derive(self.cache, 'delta', delta.get_properties, ndims=odim) for m in self.classes]

    # ---- join: pitch columns (the number of frames can differ by a few because of the downsampling in the
    # pitch tracker: same tolerance as Kaldi's paste-feats), then the only device -> host copy ------------------
    def stage_join(self, device_out, defer=False):
        utts, n, cache, log, tolerance = self.utts, self.n, self.cache, self.log, self.tolerance
        results = [None] * n
        pending = []
        for group in self.groups:
            idx = group.idx
            if group.pitch_job is not None:
                job, group.pitch_job = group.pitch_job, None
                try:
                    group.hold('pitch', job.result())
                except BaseException:
                    group.blocks.pop('wave').free()   # (a tracker that failed half way may still have readers enqueued)
                    raise
                group.drop('wave')
                # one trim / merge per pair of (features history, pitch history): same frame counts, times and
                # histories; the trimmed frame count and times depend on the two time axes only
                first, inverse = _classes_of(self.cls_of[idx], self.pcls_of[idx])
                rows_of, classes, trims = [], [], {}
                for k in first.tolist():
                    m, o = self.classes[int(self.cls_of[idx[k]])], self.pclasses[int(self.pcls_of[idx[k]])]
                    tkey = (m.nframes, id(m.times), o.nframes, id(o.times))
                    trim = trims.get(tkey)
                    if trim is None:
                        trim = trims[tkey] = Features._concatenate_meta(
                            m.nframes, m.ndims, m.times, {}, o.nframes, o.times, {}, tolerance, log)[:2]
                    r, times = trim
                    rows_of.append(r)
                    classes.append(m.derive(
                        cache, ('concat', o.key),
                        lambda m, o=o: Features._concatenate_meta(
                            1, m.ndims, m.times[:1], m.properties, 1, m.times[:1], o.properties,
                            tolerance, log)[2],
                        ndims=m.ndims + o.ndims, nframes=r, times=times))
                base = len(self.classes)
                self.classes = self.classes + classes
                self.cls_of[idx] = base + inverse
                ooff = _offsets(np.asarray(rows_of, dtype=np.int64)[inverse])
                odim = group.dim + group.pdim
                d_out = _backend.DeviceBuffer(max(int(ooff[-1]) * odim * 4, 16))
                _backend.concat_columns_device(
                    group.ptr('feat'), group.dim, group.foff, group.ptr('pitch'), group.pdim,
                    group.pfoff, d_out.ptr, ooff)
                group.drop('pitch')
                group.swap('feat', d_out)
                group.dim, group.foff = odim, ooff
            rows = int(group.foff[-1])
            if device_out is not None:
                host = np.empty((rows, group.dim), dtype=np.float32)   # (never written, never read)
                if host.size:
                    _backend.check_finite_device(group.ptr('feat'), host.size)
                device_out.append((group.give('feat'), [utts[i].name for i in idx.tolist()], group.dim))
            else:
                host = _backend.result_array((rows, group.dim), np.float32)
                if host.size:
                    # (Features.validate's data check, once for the batch and before it leaves HBM; the copy
                    # then runs while the per-utterance objects below are made - they only need to know WHERE
                    # their rows will be)
                    _backend.check_finite_device(group.ptr('feat'), host.size)
                    pending.append((group.blocks['feat'].download_async(host), group, host.nbytes))
                else:
                    group.drop('feat')
            cuts = group.foff.tolist()
            if len(self.groups) == 1:   # (views of the one downloaded array)
                results = [host[a:b] for a, b in zip(cuts, cuts[1:])]
            else:
                for i, a, b in zip(idx.tolist(), cuts, cuts[1:]):
                    results[i] = host[a:b]
        # what is this utterance's own (its audio, its speaker) is made when its properties are first read; the
        # processors' part of the properties and the times are shared by every utterance with the same history /
        # frame count and copied when first read (Features._of_batch).  Times are generated, hence sorted; the
        # data were checked above: no per-utterance validate
        from itertools import repeat
        from operator import attrgetter
        metas = list(map(self.classes.__getitem__, self.cls_of.tolist()))
        times = map(attrgetter('times'), metas)
        extras = zip(repeat(_utterance_properties), utts, self.rate_of) if self.utterance_properties \
            else repeat(None)
        out = FeaturesCollection(zip(self.names, map(Features._of_batch, results, times, metas, extras)))
        def finish():
            """waits for the copies into the result block and gives the device blocks back (once)"""
            t0 = time.perf_counter()
            while pending:
                wait, group, nbytes = pending.pop()
                try:
                    wait()
                except BaseException:
                    group.abort()   # (a copy that failed may still be writing: the plain free() waits)
                    raise
                group.drop('feat')
                self._count(bytes_down=nbytes)
            self._count(download_wait_s=time.perf_counter() - t0)
        self._count(batches=1, utterances=n)
        if defer:
            return out, finish
        finish()
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
