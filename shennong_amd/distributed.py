"""Multi-GPU driver: utterances shard embarrassingly over the ranks of one node, one process per GPU.

The reference's only parallelism is a joblib thread pool over utterances sharing one address space
(reference shennong/processor/base.py:104-107); its multi-GPU counterpart is:
  * ``shard_utterances``  - length-balanced static partition (no data-path collective),
  * ``gather_features``   - the one exchange step: variable-length gather of the float32 Features blocks
                            to a root rank as point-to-point sends (``gatherv``): every peer uses its own
                            direct xGMI link to the root instead of a ring.
  * ``allreduce_cmvn_stats`` / ``apply_cmvn_sharded`` - per-speaker CMVN (reference
    shennong/pipeline.py:580-603 accumulates every utterance of a speaker before applying): the only
    real reduction on the path.  The [n_speakers, 2, dim+1] float64 blocks are all-gathered and summed
    in rank order, so the result does not depend on the collective's internal reduction order.
Transport: every function takes a ``group``.  An ``shennong_amd.comm.RcclComm`` runs the exchange
steps over RCCL through the C ABI (``snf_comm_*``: device pointers, no framework).  ``group=None`` is
that transport too whenever the process was started by a launcher (``WORLD_SIZE`` in the environment):
one ``RcclComm.from_env()`` per process, created on first use; without a launcher it is the single-process
identity.  Any other object with the same five members - ``rank``, ``world_size``,
``all_gather_object(obj)``, ``gather_features(local, dst)``, ``allreduce(float64 array, op)`` - is used as
it is: that is how the multi-process logic is tested on a box without GPUs (``tests/tools``: a gloo process
group behind this interface, and the real ``RcclComm`` over a socket-backed stand-in of ``snf_comm_*``).
This package never imports torch.
"""

import os


import numpy as np


def _agree(transport, error):
    """Collective error check: every rank reports whether its local step failed; if any did, ALL ranks
    raise (a rank that raised alone would leave the others blocked in the next collective).  Payloads are
    tagged: a rank that failed BEFORE the statistics exchange meets its peers' ('stats', ...) gather with
    its ('status', ...) - the peers raise there (`reduce_named_stats`) and come back here, so this rank
    reports once more to meet them."""
    payload = ('status', None if error is None else '%s: %s' % (type(error).__name__, error))
    reports = transport.all_gather_object(payload)
    if any(rep[0] == 'stats' for rep in reports):
        reports = transport.all_gather_object(payload)
    failed = [(r, rep[1]) for r, rep in enumerate(reports) if rep[0] == 'status' and rep[1] is not None]
    if failed:
        if error is not None:
            raise error
        raise RuntimeError('rank %d failed: %s' % failed[0])


class _SingleTransport:
    """One process, no launcher: every exchange step is the identity"""
    rank, world_size = 0, 1

    def all_gather_object(self, obj):
        return [obj]

    def gather_features(self, local, dst=0):
        return dict(local)

    def allreduce(self, array, op='sum'):
        return np.array(array, dtype=np.float64, copy=True)


_ENV_COMM = None


def _transport(group):
    """The object that carries out the exchange steps for `group` (see the module docstring)"""
    global _ENV_COMM
    if group is not None:
        missing = [m for m in ('rank', 'world_size', 'all_gather_object', 'gather_features', 'allreduce')
                   if not hasattr(group, m)]
        if missing:
            raise TypeError('group %r is not a transport: it lacks %s (pass an shennong_amd.comm.RcclComm)'
                            % (group, ', '.join(missing)))
        return group
    if _ENV_COMM is not None:
        return _ENV_COMM
    if int(os.environ.get('WORLD_SIZE', '0')) > 0 and 'RANK' in os.environ:
        from shennong_amd import _backend
        from shennong_amd.comm import RcclComm
        if int(os.environ['WORLD_SIZE']) > 1 and _backend.device_count() < 1:
            # (rounds 1-3 fell back to an initialised torch process group here; the package no longer imports
            # torch: a launched job without GPUs has to pass its transport explicitly)
            raise RuntimeError(
                'WORLD_SIZE=%s but no GPU is visible: the default transport is RCCL (shennong_amd.comm.RcclComm); '
                'pass group=<an object with rank, world_size, all_gather_object, gather_features, allreduce> '
                'to run the exchange steps over something else (tests/tools/torch_transport.py wraps a gloo '
                'process group that way)' % os.environ['WORLD_SIZE'])
        _ENV_COMM = RcclComm.from_env()
        return _ENV_COMM
    return _SingleTransport()


def shard_utterances(lengths, world_size):
    """Greedy longest-first partition of utterance indices by length (sample counts or durations in
    seconds: any non-negative numbers).

    Returns `world_size` lists of indices (each sorted ascending); deterministic, every index appears
    exactly once, the per-rank totals differ by at most the longest utterance."""
    lengths = np.asarray(lengths)
    if lengths.dtype.kind not in 'iu':
        lengths = lengths.astype(np.float64)  # (durations: truncating them would send every
                                              # sub-second utterance to rank 0)
    order = np.argsort(-lengths, kind='stable')
    totals = np.zeros(world_size, dtype=lengths.dtype if lengths.size else np.int64)
    shards = [[] for _ in range(world_size)]
    for idx in order:
        r = int(np.argmin(totals))
        shards[r].append(int(idx))
        totals[r] += lengths[idx]
    return [sorted(s) for s in shards]


def gather_features(local, dst=0, group=None):
    """Gathers per-rank ``{name: float32 [nframes, ndims]}`` dicts on rank `dst`.

    Names and shapes travel as a small all-gathered object; the matrices travel as ONE contiguous
    float32 buffer per peer, sent point-to-point to `dst`.  Returns the merged dict on `dst`, None
    elsewhere."""
    return _transport(group).gather_features(local, dst)


def process_all_sharded(processor, utterances, dst=0, group=None, **kwargs):
    """``processor.process_all`` over the utterances owned by this rank, then the gather.

    Every rank passes the same `utterances`; the shards are balanced by the durations the utterance index
    already holds and every rank decodes ONLY its own shard.  Over RCCL the feature rows never visit the
    host on the way: the kernel's output buffer is handed to ``snf_comm_gatherv`` as it is and the root
    makes one download of everything.  Rank `dst` gets the full FeaturesCollection (same keys and values
    as a single-process ``process_all``), the others get None."""
    from shennong_amd.comm import RcclComm
    from shennong_amd.features import FeaturesCollection
    transport = _transport(group)
    rank, world = transport.rank, transport.world_size
    utts = list(utterances)
    shards = shard_utterances([u.duration for u in utts], world)
    mine = shards[rank]
    signals = [utts[i].load_audio() for i in mine]
    per_utt = {k: [v[utts[i].name] for i in mine] for k, v in kwargs.items()}
    if isinstance(transport, RcclComm) and set(kwargs) <= {'vtln_warp'} and hasattr(processor, '_build_options'):
        merged = _gather_device_resident(processor, [utts[i].name for i in mine], signals,
                                         per_utt.get('vtln_warp'), transport, dst)
    else:
        feats = processor._process_batch(signals, **per_utt) if mine else []
        merged = gather_features({utts[i].name: f.data for i, f in zip(mine, feats)}, dst=dst, group=transport)
    if merged is None:
        return None
    feats = processor._wrap_batch([merged[u.name] for u in utts],
                                  **{k: [v[u.name] for u in utts] for k, v in kwargs.items()})
    return FeaturesCollection((u.name, f) for u, f in zip(utts, feats))


def _gather_device_resident(processor, names, signals, warps, comm, dst):
    """This rank's shard through ``plan.run_device`` and the rows straight from that buffer to the root:
    ``{name: float32 [nframes, ndims]}`` of all ranks on `dst`, None elsewhere"""
    from shennong_amd import _backend
    from shennong_amd.processor.base import check_signal
    for signal in signals:
        check_signal(processor, signal)
    plan = _backend.get_plan(processor._build_options())
    ndims = plan.ndims
    waves = [np.ascontiguousarray(s.astype(np.int16).data) for s in signals]
    soff = np.zeros(len(waves) + 1, dtype=np.int64)
    np.cumsum([w.shape[0] for w in waves], out=soff[1:])
    nframes = [int(plan.num_frames(w.shape[0])) for w in waves]
    foff = np.zeros(len(waves) + 1, dtype=np.int64)
    np.cumsum(nframes, out=foff[1:])
    meta = comm.all_gather_object((list(names), nframes))
    counts = [sum(nf) * ndims for _, nf in meta]
    d_wave = d_out = d_all = None
    try:
        # (threaded gather + copy through page-locked staging, like the single-process pipeline)
        d_wave = _backend.upload_rows(waves, np.int16, device=comm.device) if waves else None
        d_out = _backend.DeviceBuffer(max(int(foff[-1]) * ndims * 4, 16), device=comm.device)
        if foff[-1] > 0:
            plan.run_device(d_wave.ptr, soff, foff, d_out.ptr, vtln_warps=warps)
        if comm.rank == dst:
            d_all = _backend.DeviceBuffer(max(4 * sum(counts), 16), device=comm.device)
        comm.gatherv_device(d_out.ptr, int(foff[-1]) * ndims, d_all.ptr if d_all else None, counts, dst)
        if comm.rank != dst:
            return None
        # Features.validate's data check on the gathered block while it is in HBM, then ONE copy to the host
        host = _backend.result_array((sum(counts),), np.float32)
        if host.size:
            try:
                _backend.check_finite_device(d_all.ptr, host.size, device=comm.device)
            except ValueError:
                raise ValueError('features are not valid (non-finite values)') from None
            d_all.download(host)
    finally:
        for buf in (d_wave, d_out, d_all):
            if buf is not None:
                buf.free(synced=True)   # (launch, gather and download have all completed)
    merged, pos = {}, 0
    for names_r, nframes_r in meta:
        for name, nf in zip(names_r, nframes_r):
            merged[name] = host[pos:pos + nf * ndims].reshape(nf, ndims)  # (views of the one block)
            pos += nf * ndims
    return merged


def allreduce_cmvn_stats(stats, group=None):
    """Sums float64 CMVN statistics blocks [n_speakers, 2, dim + 1] over the ranks: one float64
    ncclAllReduce on the device (``snf_comm_allreduce_f64``).  Identical on every rank."""
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    return np.asarray(_transport(group).allreduce(stats, 'sum')).reshape(stats.shape)


def apply_cmvn_sharded(local_feats, utt2speak=None, norm_vars=True, weights=None,
                       skip_dims=None, group=None, _plan=None):
    """Per-speaker CMVN when the utterances of a speaker are spread over the ranks.

    `local_feats` is THIS rank's FeaturesCollection (as ``process_all`` over its shard returns);
    `utt2speak` maps utterance name -> speaker (None: one normalisation over the whole distributed
    collection, the reference's ``apply_cmvn(by_collection=True)``).  Every rank gets its own shard
    normalised with the statistics of the complete speakers: local statistics launch ->
    ``allreduce_cmvn_stats`` -> local apply launch.  Returns ``(FeaturesCollection, stats dict)``."""
    from shennong_amd import _abi, _backend
    from shennong_amd.features import Features, FeaturesCollection
    from shennong_amd.postprocessor.cmvn import CmvnPostProcessor, _fake_stats_for_dims
    keys = list(local_feats.keys())
    speak = [None if utt2speak is None else utt2speak[k] for k in keys]
    meta = _transport(group).all_gather_object(
        (sorted(set(speak), key=str), sorted(set(local_feats[k].ndims for k in keys))))
    speakers = sorted(set(s for m in meta for s in m[0]), key=str)
    dims = sorted(set(d for m in meta for d in m[1]))
    if len(dims) != 1:
        raise ValueError(
            'features in the collection must have consistent dimensions '
            'but dimensions are: {}'.format(dims))
    dim = dims[0]
    index = {s: i for i, s in enumerate(speakers)}
    groups = np.asarray([index[s] for s in speak], dtype=np.int32)
    plan = _plan or _backend.get_plan(_abi.default_options(_abi.KIND_CMVN))
    mats = [np.asarray(local_feats[k].data, dtype=np.float32) for k in keys]
    stats = np.zeros((len(speakers), 2, dim + 1), dtype=np.float64)
    if mats:
        plan.cmvn_accumulate(
            mats, stats, groups=groups,
            weights=None if weights is None else [weights[k] for k in keys])
    stats = allreduce_cmvn_stats(stats, group=group)
    for s, i in index.items():
        if stats[i, 0, -1] < 1.0:
            raise ValueError(
                'insufficient accumulation of stats for CMVN, '
                'must be >= 1.0 but is {}'.format(stats[i, 0, -1]))
    applied = stats
    if skip_dims:
        applied = np.stack([_fake_stats_for_dims(st, skip_dims) for st in stats])
    datas = plan.cmvn_apply(mats, applied, groups=groups, norm_vars=norm_vars) if mats else []
    out = FeaturesCollection()
    for u, k in enumerate(keys):
        proc = CmvnPostProcessor(dim, stats=stats[groups[u]])
        out[k] = Features(datas[u], local_feats[k].times,
                          properties=proc.get_properties(local_feats[k]))
    return out, {s: stats[i] for s, i in index.items()}


def reduce_named_stats(names, stats, group=None):
    """Sums ``stats[k]`` (float64 [len(names), 2, dim + 1]) of equally named entries over the ranks.

    Ranks may know different name lists (each holds its own utterances): the union of the names is
    agreed on with one small object all-gather, the blocks are summed with `allreduce_cmvn_stats`
    (rank-ordered, deterministic).  Returns the blocks of THIS rank's `names`, in its order."""
    gathered = _transport(group).all_gather_object(
        ('stats', list(names), int(stats.shape[-1]) if len(names) else None))
    # a peer that failed before it got here is in `_agree`: everybody stops (and meets it there again)
    for rep in gathered:
        if rep[0] != 'stats':
            raise RuntimeError('another rank failed before the statistics exchange: %s' % (rep[1],))
    gathered = [rep[1:] for rep in gathered]
    widths = sorted(set(w for _, w in gathered if w is not None))
    if not widths:
        return stats
    if len(widths) != 1:
        raise ValueError('features have inconsistent dimensions across ranks: {}'.format(widths))
    union = sorted(set(n for ns, _ in gathered for n in ns), key=str)
    index = {n: i for i, n in enumerate(union)}
    full = np.zeros((len(union), 2, widths[0]), dtype=np.float64)
    for k, name in enumerate(names):
        full[index[name]] = stats[k]
    full = allreduce_cmvn_stats(full, group=group)
    return np.stack([full[index[name]] for name in names]) if len(names) else full[:0]


def extract_features_sharded(configuration, utterances, warps=None, dst=0, group=None, log=None):
    """``pipeline.extract_features`` over the ranks of one node: every rank passes the same
    `utterances`, works on its length-balanced shard with the device-resident pipeline, and the CMVN
    statistics of speakers whose utterances landed on several ranks are summed across the ranks before
    they are applied (by-utterance CMVN needs no exchange).  Rank `dst` gets the complete
    FeaturesCollection (point-to-point gather of the matrices, the properties travel as objects), the
    others get None."""
    from shennong_amd import pipeline
    from shennong_amd.features import Features, FeaturesCollection
    from shennong_amd.logger import get_logger
    from shennong_amd.utterances import Utterances
    log = log or get_logger('pipeline', 'warning')
    transport = _transport(group)
    rank, world = transport.rank, transport.world_size
    config = pipeline._init_config(configuration, log=log)
    if warps:
        warps = pipeline._init_warps(warps, config, utterances, log)
    utts = list(utterances)
    shards = shard_utterances([u.duration for u in utts], world)
    mine = [utts[i] for i in shards[rank]]
    by_speaker = 'cmvn' in config and config['cmvn']['by_speaker']
    hook = (lambda names, stats: reduce_named_stats(names, stats, group=group)) if by_speaker else None
    # the statistics hook is a collective: a rank whose local extraction fails BEFORE reaching it would
    # leave the others waiting, so every rank first validates what it can locally and the ranks agree
    # on the outcome; the same after the extraction, before the gather
    error = None
    try:
        from shennong_amd.audio import Audio
        if not all(Audio.scan(u.audio_file).nchannels == 1 for u in mine):  # (what the pipeline refuses
            raise ValueError('all audio files are not mono')                # first; headers only)
    except Exception as exc:  # noqa: BLE001
        error = exc
    _agree(transport, error)
    local = FeaturesCollection()
    blocks = []   # RCCL: the final matrices of this rank stay in HBM until the gather has sent them
    from shennong_amd.comm import RcclComm
    on_device = isinstance(transport, RcclComm)
    try:
        if mine:
            local = pipeline._extract_features(
                config, Utterances(mine), {u.name: warps[u.name] for u in mine} if warps else None, log,
                stats_hook=hook, device_out=blocks if on_device else None)
        elif hook is not None:  # still take part in the reduction
            hook([], np.zeros((0, 2, 1), dtype=np.float64))
    except Exception as exc:  # noqa: BLE001
        error = exc
    try:
        _agree(transport, error)
        if on_device:
            merged = _gather_blocks(transport, local, blocks, dst)
        else:
            merged = gather_features({k: v.data for k, v in local.items()}, dst=dst, group=group)
    finally:
        for d_buf, _, _ in blocks:
            d_buf.free(synced=True)
    meta = transport.all_gather_object({k: (v.times, v.properties) for k, v in local.items()})
    if merged is None:
        return None
    out = FeaturesCollection()
    everything = {k: v for m in meta for k, v in m.items()}
    for u in utts:
        times, properties = everything[u.name]
        out[u.name] = Features(merged[u.name], times, properties=properties, validate=False)
    return out


def _gather_blocks(comm, local, blocks, dst):
    """The device-resident gather of `extract_features_sharded`: every rank's final [rows, ndims] blocks (one
    per sample rate, still in HBM) go point to point to `dst` (``snf_comm_gatherv``, one call per block
    position), which downloads everything once.  ``{name: float32 [nframes, ndims]}`` on `dst` (views of the
    one downloaded block), None elsewhere."""
    from shennong_amd import _backend
    mine = [(names, [int(local[n].shape[0]) for n in names], ndims) for _, names, ndims in blocks]
    meta = comm.all_gather_object(mine)
    rounds = max(len(m) for m in meta)
    sizes = [[sum(m[k][1]) * m[k][2] if k < len(m) else 0 for m in meta] for k in range(rounds)]
    total = sum(sum(row) for row in sizes)
    d_all = host = None
    placeholder = None
    try:
        if comm.rank == dst:
            d_all = _backend.DeviceBuffer(max(4 * total, 16), device=comm.device)
        base = 0
        for k in range(rounds):
            count = sizes[k][comm.rank]
            if k < len(blocks):
                send_ptr = blocks[k][0].ptr
            else:   # (this rank has no block at this position: it still takes part with a count of 0)
                if placeholder is None:
                    placeholder = _backend.DeviceBuffer(16, device=comm.device)
                send_ptr = placeholder.ptr
            comm.gatherv_device(send_ptr, count, d_all.ptr + 4 * base if d_all else None, sizes[k], dst)
            base += sum(sizes[k])
        if comm.rank != dst:
            return None
        host = _backend.result_array((total,), np.float32)
        if total:
            d_all.download(host)
    finally:
        for buf in (d_all, placeholder):
            if buf is not None:
                buf.free(synced=True)
    merged, pos = {}, 0
    for k in range(rounds):
        for m in meta:
            if k < len(m):
                names, nframes, ndims = m[k]
                for name, nf in zip(names, nframes):
                    merged[name] = host[pos:pos + nf * ndims].reshape(nf, ndims)
                    pos += nf * ndims
    return merged


def extract_features_streamed_sharded(configuration, utterances, sink, warps=None,
                                      max_batch_duration=None, group=None, log=None):
    """``pipeline.extract_features_streamed`` over the ranks of one node (BASELINE config 5): every
    rank passes the same `utterances`, streams its length-balanced shard batch by batch through the
    device-resident pipeline and hands the batches to ITS OWN `sink` (e.g. one
    ``KaldiStreamWriter('feats.<rank>.ark')`` per rank: no features travel between the ranks).  The
    only exchange is the sum of the per-speaker CMVN statistics after the first pass (a few KB,
    `reduce_named_stats`).  Returns the number of utterances this rank wrote."""
    from shennong_amd import pipeline
    from shennong_amd.logger import get_logger
    from shennong_amd.utterances import Utterances
    log = log or get_logger('pipeline', 'warning')
    transport = _transport(group)
    rank, world = transport.rank, transport.world_size
    config = pipeline._init_config(configuration, log=log)
    if warps:
        warps = pipeline._init_warps(warps, config, utterances, log)
    utts = list(utterances)
    shards = shard_utterances([u.duration for u in utts], world)
    mine = [utts[i] for i in sorted(shards[rank])]
    by_speaker = 'cmvn' in config and config['cmvn']['by_speaker']
    if by_speaker and not utterances.has_speakers():
        raise ValueError(
            'cmvn normalization by speaker requested '
            'but no speaker information provided')

    def reduce(names, stats):
        return reduce_named_stats(names, stats, group=group)

    if not mine:
        if by_speaker:  # still take part in the reduction
            reduce([], np.zeros((0, 2, 1), dtype=np.float64))
        return 0
    return pipeline.extract_features_streamed(
        configuration, Utterances(mine), sink,
        warps={u.name: warps[u.name] for u in mine} if warps else None,
        max_batch_duration=max_batch_duration, stats_reduce=reduce if by_speaker else None, log=log)
