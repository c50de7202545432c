"""The N > 1 path on CPU: world_size-2 processes exercise sharding, the Features gather and the CMVN
statistics exchange (the GPU kernels are not involved; rank-local 'features' are synthetic matrices), over
two transports (tests/tools/transports.py): a gloo process group behind the transport interface, and the
product's own ``RcclComm.from_env()`` - TCP rendezvous, unique-id broadcast, object all-gather, gatherv
counts and offsets, float64 all-reduce - over a socket-backed stand-in of the ``snf_comm_*`` entry points."""

import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from shennong_amd.distributed import shard_utterances

TOOLS = os.path.join(ROOT, 'tests', 'tools')
TRANSPORTS = ('gloo', 'rccl_stub')


def _open(kind, rank, world, port):
    for path in (ROOT, os.path.join(ROOT, 'tests'), TOOLS):
        if path not in sys.path:
            sys.path.insert(0, path)
    import transports
    return transports.open_transport(kind, rank, world, port)


def _run_pair(target, kind, tmp_path, world=2):
    """spawns `world` worker processes and returns what each wrote to its result file"""
    import multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=target, args=(kind, r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
    for p in procs:
        if p.is_alive():
            p.kill()
    assert [p.exitcode for p in procs] == [0] * world
    return [open(tmp_path / f'out{r}').read() for r in range(world)]


def test_shard_utterances_balanced():
    rng = np.random.default_rng(0)
    lengths = rng.integers(16000, 96000, size=1001)
    for world in (1, 2, 4, 8):
        shards = shard_utterances(lengths, world)
        assert sorted(i for s in shards for i in s) == list(range(1001))
        totals = [int(lengths[s].sum()) for s in shards]
        assert max(totals) - min(totals) <= lengths.max()
    assert shard_utterances([], 2) == [[], []]
    assert shard_utterances([5, 5, 5], 4) == [[0], [1], [2], []]
    # durations in seconds are not truncated (sub-second utterances would all land on rank 0)
    assert shard_utterances([0.3, 0.4, 0.5, 0.6, 0.2], 2) == [[0, 3, 4], [1, 2]]


def _worker(kind, rank, world, port, tmpdir):
    transport, close = _open(kind, rank, world, port)
    from shennong_amd.distributed import gather_features, shard_utterances
    rng = np.random.default_rng(7)
    nframes = rng.integers(0, 50, size=11)
    mats = {f'utt{i}': np.random.default_rng(100 + i).standard_normal(
        (int(n), 5)).astype(np.float32) for i, n in enumerate(nframes)}
    shards = shard_utterances(nframes, world)
    local = {f'utt{i}': mats[f'utt{i}'] for i in shards[rank]}
    merged = gather_features(local, dst=0, group=transport)
    ok = True
    if rank == 0:
        ok = sorted(merged) == sorted(mats) and all(
            np.array_equal(merged[k], mats[k]) for k in mats)
    else:
        ok = merged is None
    # a rank with nothing to send must not deadlock the gather
    merged = gather_features(local if rank == 0 else {}, dst=0, group=transport)
    if rank == 0:
        ok = ok and sorted(merged) == sorted(local)
    # a group that is not a transport is refused with its missing members named
    try:
        gather_features(local, group=object())
        ok = False
    except TypeError as exc:
        ok = ok and 'all_gather_object' in str(exc)
    close()
    open(os.path.join(tmpdir, f'out{rank}'), 'w').write('1' if ok else '0')


@pytest.mark.timeout(120)
@pytest.mark.parametrize('kind', TRANSPORTS)
def test_gather_features(tmp_path, kind):
    assert _run_pair(_worker, kind, tmp_path) == ['1', '1']


@pytest.mark.timeout(120)
def test_gather_features_three_ranks(tmp_path):
    """three ranks through the communicator class: the root's receive offsets are prefix sums over more
    than one peer, and a middle rank may be the one with nothing to send"""
    assert _run_pair(_worker, 'rccl_stub', tmp_path, world=3) == ['1', '1', '1']


class _OraclePlan:
    """Stand-in for the HIP CMVN plan so that the collective logic can run on CPU ranks (test
    infrastructure: same call surface as shennong_amd._backend.Plan.cmvn_*)"""
    def cmvn_accumulate(self, mats, stats, weights=None, groups=None):
        from oracle import oracle as orc
        for u, m in enumerate(mats):
            g = 0 if groups is None else int(groups[u])
            orc.cmvn_accumulate(m, weights=None if weights is None else weights[u], stats=stats[g])
        return stats

    def cmvn_apply(self, mats, stats, groups=None, norm_vars=True, reverse=False):
        from oracle import oracle as orc
        return [orc.cmvn_apply(m, stats[0 if groups is None else int(groups[u])],
                               norm_vars=norm_vars, reverse=reverse) for u, m in enumerate(mats)]


def _cmvn_case():
    from shennong_amd import Features, FeaturesCollection
    rng = np.random.default_rng(11)
    nframes = rng.integers(3, 60, size=9)
    coll = FeaturesCollection()
    for i, n in enumerate(nframes):
        data = (rng.standard_normal((int(n), 4)) * (1 + i % 3) + i).astype(np.float32)
        coll[f'utt{i}'] = Features(data, np.arange(int(n), dtype=np.float64))
    utt2speak = {f'utt{i}': f'spk{i % 3}' for i in range(9)}
    return coll, utt2speak, nframes


def _cmvn_worker(kind, rank, world, port, tmpdir):
    transport, close = _open(kind, rank, world, port)
    from oracle import oracle as orc
    from shennong_amd import FeaturesCollection
    from shennong_amd.distributed import (
        allreduce_cmvn_stats, apply_cmvn_sharded, shard_utterances)
    coll, utt2speak, nframes = _cmvn_case()
    shards = shard_utterances(nframes, world)
    local = FeaturesCollection({f'utt{i}': coll[f'utt{i}'] for i in shards[rank]})
    ok = True
    # the collective alone: rank-ordered sum of float64 blocks
    mine = np.full((3, 2, 5), float(rank + 1)) * np.arange(30).reshape(3, 2, 5)
    tot = allreduce_cmvn_stats(mine, group=transport)
    ok = ok and np.array_equal(tot, 3.0 * np.arange(30).reshape(3, 2, 5))
    for mapping in (utt2speak, None):
        got, stats = apply_cmvn_sharded(local, mapping, group=transport, _plan=_OraclePlan())
        ok = ok and list(got.keys()) == list(local.keys())
        for k in local.keys():
            spk = None if mapping is None else mapping[k]
            members = [u for u in coll.keys() if mapping is None or mapping[u] == spk]
            want_stats = np.zeros((2, 5))
            for u in members:
                orc.cmvn_accumulate(coll[u].data, stats=want_stats)
            ok = ok and np.allclose(stats[spk], want_stats, rtol=1e-13, atol=0)
            ok = ok and np.array_equal(got[k].data, orc.cmvn_apply(coll[k].data, stats[spk]))
            ok = ok and got[k].properties['cmvn']['stats'].shape == (2, 5)
    # a rank without any utterance still takes part in the reduction
    got, stats = apply_cmvn_sharded(local if rank == 0 else FeaturesCollection(), None,
                                    group=transport, _plan=_OraclePlan())
    ok = ok and len(got) == (len(local) if rank == 0 else 0)
    close()
    open(os.path.join(tmpdir, f'out{rank}'), 'w').write('1' if ok else '0')


@pytest.mark.timeout(120)
@pytest.mark.parametrize('kind', TRANSPORTS)
def test_cmvn_sharded(tmp_path, kind):
    assert _run_pair(_cmvn_worker, kind, tmp_path) == ['1', '1']


def _named_stats_worker(kind, rank, world, port, tmpdir):
    transport, close = _open(kind, rank, world, port)
    from shennong_amd.distributed import reduce_named_stats
    rng = np.random.default_rng(5)
    all_stats = {f's{k}': rng.random((2, 4)) for k in range(4)}
    # rank 0 holds s0, s1, s2; rank 1 holds s2, s3 (each with its own partial sums)
    names = [['s0', 's1', 's2'], ['s2', 's3']][rank]
    mine = np.stack([all_stats[n] * (rank + 1) for n in names])
    got = reduce_named_stats(names, mine, group=transport)
    weight = {'s0': 1, 's1': 1, 's2': 3, 's3': 2}
    ok = got.shape == mine.shape and all(
        np.allclose(got[k], all_stats[n] * weight[n], rtol=1e-15) for k, n in enumerate(names))
    # a rank without any utterance still takes part
    got = reduce_named_stats(names if rank == 0 else [], mine if rank == 0 else np.zeros((0, 2, 1)),
                             group=transport)
    ok = ok and (np.array_equal(got, mine) if rank == 0 else got.shape[0] == 0)
    close()
    open(os.path.join(tmpdir, f'out{rank}'), 'w').write('1' if ok else '0')


@pytest.mark.timeout(120)
@pytest.mark.parametrize('kind', TRANSPORTS)
def test_reduce_named_stats(tmp_path, kind):
    """the exchange step of the multi-rank pipeline (per-speaker CMVN statistics over ranks that know
    different speaker lists)"""
    assert _run_pair(_named_stats_worker, kind, tmp_path) == ['1', '1']


def _streamed_worker(kind, rank, world, port, tmpdir):
    transport, close = _open(kind, rank, world, port)
    from conftest import GOLDEN
    from shennong_amd import Utterances, pipeline
    from shennong_amd.distributed import extract_features_streamed_sharded
    wav = os.path.join(GOLDEN, 'test.wav')
    index = Utterances([(f'u{i}', wav, f's{i % 3}', 0.1 * (i % 4), 0.1 * (i % 4) + 0.3 + 0.1 * (i % 5))
                        for i in range(1, 10)])

    def fake(config, utterances, warps, log, tolerance=2, stats_hook=None, stats_only=False, **_resident):
        # stand-in for the device pipeline (no GPU on the CPU ranks): the "statistics" of utterance
        # u<i> are i, the "features" of an utterance are its speaker's global statistics
        utts = list(utterances)
        per_utt = np.stack([np.full((2, 3), float(int(u.name[1:]))) for u in utts])
        if stats_only:
            return [u.speaker for u in utts], per_utt
        names = list(dict.fromkeys(u.speaker for u in utts))
        stats = stats_hook(names, np.zeros((len(names), 2, 3)))
        out = {u.name: float(stats[names.index(u.speaker)][0, 0]) for u in utts}
        return (out, lambda: None) if _resident.get('defer') else out

    pipeline._extract_features = fake
    config = pipeline.get_default_config('mfcc', with_cmvn=True)
    out = {}
    n = extract_features_streamed_sharded(config, index, out.update, max_batch_duration=1.0, group=transport)
    want = {f's{k}': float(sum(i for i in range(1, 10) if i % 3 == k)) for k in range(3)}
    ok = n == len(out) and 0 < n < 9
    ok = ok and all(v == want[f's{int(k[1:]) % 3}'] for k, v in out.items())
    counts = transport.all_gather_object(sorted(out))
    ok = ok and sorted(counts[0] + counts[1]) == [f'u{i}' for i in range(1, 10)]
    close()
    open(os.path.join(tmpdir, f'out{rank}'), 'w').write('1' if ok else f'0 {n} {out} {counts}')  # noqa


@pytest.mark.timeout(120)
@pytest.mark.parametrize('kind', TRANSPORTS)
def test_streamed_sharded(tmp_path, kind):
    """streamed extraction over two ranks: each rank streams its shard into its own sink, the
    speakers' statistics are summed over all batches of all ranks before the second pass"""
    assert _run_pair(_streamed_worker, kind, tmp_path) == ['1', '1']


def _sharded_worker(kind, rank, world, port, tmpdir):
    transport, close = _open(kind, rank, world, port)
    from conftest import GOLDEN
    from shennong_amd import Features, FeaturesCollection, Utterances, _backend, pipeline
    from shennong_amd.distributed import extract_features_sharded
    wav = os.path.join(GOLDEN, 'test.wav')
    index = Utterances([(f'u{i}', wav, f's{i % 3}', 0.1 * (i % 4), 0.1 * (i % 4) + 0.3 + 0.1 * (i % 5))
                        for i in range(1, 10)])

    def want(name):
        k = int(name[1:])
        return np.full((3 + k % 5, 4), float(k), np.float32) + np.arange(4, dtype=np.float32)

    def fake(config, utterances, warps, log, tolerance=2, stats_hook=None, device_out=None, **_rest):
        # stand-in for the device pipeline: utterance u<k> yields a known matrix; the even and the odd
        # utterances stand for two sample rates (two blocks); with `device_out` the rows stay "in HBM"
        # (host memory behind the stand-in's DeviceBuffer) and the Features carry untouched placeholders
        coll = FeaturesCollection()
        utts = list(utterances)
        for parity in (0, 1):
            names = [u.name for u in utts if int(u.name[1:]) % 2 == parity]
            if not names:
                continue
            mats = [want(n) for n in names]
            if device_out is not None:
                flat = np.concatenate([m.reshape(-1) for m in mats])
                buf = _backend.DeviceBuffer(flat.nbytes)
                buf.upload(flat)
                device_out.append((buf, names, 4))
            for n, m in zip(names, mats):
                data = m if device_out is None else np.full_like(m, np.nan)
                coll[n] = Features(data, np.arange(m.shape[0], dtype=np.float64), properties={'name': n},
                                   validate=False)
        return coll

    pipeline._extract_features = fake
    config = pipeline.get_default_config('mfcc', with_cmvn=False)
    out = extract_features_sharded(config, index, dst=0, group=transport)
    ok = True
    if rank == 0:
        ok = sorted(out.keys()) == sorted(f'u{i}' for i in range(1, 10))
        ok = ok and all(np.array_equal(out[k].data, want(k)) for k in out.keys())
        ok = ok and all(out[k].properties == {'name': k} for k in out.keys())
        ok = ok and all(np.array_equal(out[k].times, np.arange(want(k).shape[0])) for k in out.keys())
    else:
        ok = out is None
    if kind == 'rccl_stub':   # the rows went through snf_comm_gatherv, one call per block position
        ok = ok and sum(1 for c in transport.fake.calls if c[0] == 'gatherv') == 2
    close()
    open(os.path.join(tmpdir, f'out{rank}'), 'w').write('1' if ok else '0')


@pytest.mark.timeout(120)
@pytest.mark.parametrize('kind', TRANSPORTS)
def test_extract_features_sharded(tmp_path, kind):
    """the whole-pipeline driver over two ranks: shards, agreement, gather (device-resident blocks over the
    communicator class; host dictionaries over the gloo transport), times and properties as objects"""
    assert _run_pair(_sharded_worker, kind, tmp_path) == ['1', '1']


def _mixed_failure_worker(kind, rank, world, port, tmpdir):
    transport, close = _open(kind, rank, world, port)
    from shennong_amd import distributed
    error = None
    try:
        if rank == 1:
            raise ValueError('all audio files are not mono')   # fails before the statistics exchange
        distributed.reduce_named_stats(['s0'], np.ones((1, 2, 3)), group=transport)
    except Exception as exc:  # noqa: BLE001
        error = exc
    try:
        distributed._agree(transport, error)
        outcome = 'no error'
    except Exception as exc:  # noqa: BLE001
        outcome = '%s: %s' % (type(exc).__name__, exc)
    close()
    open(os.path.join(tmpdir, f'out{rank}'), 'w').write(outcome)


@pytest.mark.timeout(120)
@pytest.mark.parametrize('kind', TRANSPORTS)
def test_failure_before_the_statistics_exchange_stops_every_rank(tmp_path, kind):
    """one rank fails before it reaches the by-speaker statistics exchange while the other is inside it: the
    tagged payloads let the healthy rank stop there and both meet again in `_agree` (no rank is left
    unpacking a status as statistics or waiting for a peer that has gone)"""
    out = _run_pair(_mixed_failure_worker, kind, tmp_path)
    assert out[1] == 'ValueError: all audio files are not mono'
    assert out[0].startswith('RuntimeError: another rank failed before the statistics exchange')


def _comm_class_worker(kind, rank, world, port, tmpdir):
    """what the communicator class itself does, call by call, on the stand-in (mirrors _rccl_pair_worker,
    which needs two GPUs)"""
    comm, close = _open(kind, rank, world, port)
    from shennong_amd import _backend
    ok = (comm.rank, comm.world_size) == (rank, world)
    # the id that rank 0 made reached the peer through the rendezvous (the stand-in's port travels in it)
    ok = ok and comm.fake.calls[-1] == ('init', world, rank, rank)
    ok = ok and (comm.fake.calls[0][0] == 'unique_id') == (rank == 0)
    # variable-length gather to each root in turn: rank r sends (r + 1) * 1000 + 7 floats of value r + 0.5
    for root in (0, 1):
        counts = [1007, 2007]
        mine = np.full(counts[rank], rank + 0.5, dtype=np.float32)
        d_send = _backend.DeviceBuffer(mine.nbytes, device=rank)
        d_send.upload(mine)
        d_recv = _backend.DeviceBuffer(4 * sum(counts), device=rank) if rank == root else None
        comm.gatherv_device(d_send.ptr, mine.size, d_recv.ptr if d_recv else None, counts, root)
        if rank == root:
            got = np.empty(sum(counts), dtype=np.float32)
            d_recv.download(got)
            ok = ok and np.array_equal(got, np.concatenate([np.full(1007, 0.5), np.full(2007, 1.5)]).astype(np.float32))
    # the root's own count must match what it sends (argument error, raised before anything moves)
    if rank == 0:
        try:
            comm.gatherv_device(d_send.ptr, 5, d_send.ptr, [6, 0], 0)
            ok = False
        except ValueError as exc:
            ok = ok and 'recv_counts[root]' in str(exc)
    # float64 all-reduce against numpy, sum and max; ordering of successive collectives
    base = np.arange(48, dtype=np.float64).reshape(4, 2, 6) / 7.0
    ok = ok and np.array_equal(comm.allreduce(base * (rank + 1), 'sum'), base * 1 + base * 2)
    ok = ok and comm.allreduce(np.array([float(rank)]), 'max')[0] == 1.0
    ok = ok and comm.allreduce(np.zeros(0), 'sum').size == 0
    ok = ok and comm.all_gather_object(('r', rank)) == [('r', 0), ('r', 1)]
    merged = comm.gather_features({'u%d' % rank: np.full((3 + rank, 2), rank, np.float32)}, dst=0)
    if rank == 0:
        ok = ok and sorted(merged) == ['u0', 'u1'] and merged['u1'].shape == (4, 2) and (merged['u1'] == 1).all()
        ok = ok and ('gatherv', 'root', [6, 8]) in comm.fake.calls
    else:
        ok = ok and merged is None and ('gatherv', 'send', 8) in comm.fake.calls
    close()
    open(os.path.join(tmpdir, f'out{rank}'), 'w').write('1' if ok else '0 %s' % (comm.fake.calls,))


@pytest.mark.timeout(120)
def test_rccl_comm_class_two_processes(tmp_path):
    """``RcclComm.from_env()`` in two processes on CPU: rendezvous, id broadcast, gatherv counts / offsets to
    either root, all-reduce sum / max, object all-gather (shennong_amd/comm.py end to end; the C side is
    the socket-backed stand-in of tests/tools/fake_comm.py, which restates csrc/comm.cpp)"""
    assert _run_pair(_comm_class_worker, 'rccl_stub', tmp_path) == ['1', '1']


def test_rendezvous_rejects_another_jobs_token(tmp_path):
    """a peer that holds a different SNF_COMM_TOKEN never becomes a member: rank 0 drops its hello and
    keeps waiting for the real peer (here: until its timeout)"""
    import threading
    for path in (ROOT, TOOLS):
        if path not in sys.path:
            sys.path.insert(0, path)
    import fake_comm
    from shennong_amd import _backend, comm as comm_mod
    real_lib, real_buf = _backend._LIB, _backend.DeviceBuffer
    try:
        fake_comm.install()
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        errors = {}

        def rank0():
            os.environ['SNF_COMM_TOKEN'] = 'job-a'
            try:
                comm_mod.RcclComm(0, 2, device=0, port=port, timeout=1.5)
            except Exception as exc:  # noqa: BLE001
                errors[0] = exc
        t = threading.Thread(target=rank0)
        t.start()
        key_b = __import__('hashlib').sha256(b'shennong_amd.comm/1:job-b').digest()
        import time
        time.sleep(0.3)
        conn = socket.create_connection(('127.0.0.1', port), timeout=5)
        comm_mod._send_msg(conn, comm_mod._HELLO.pack(comm_mod._MAGIC, 1), key_b)
        t.join(10)
        conn.close()
        assert isinstance(errors.get(0), (TimeoutError, socket.timeout, OSError)), errors
    finally:
        _backend._LIB, _backend.DeviceBuffer = real_lib, real_buf
        os.environ.pop('SNF_COMM_TOKEN', None)


def test_rendezvous_port_next_to_the_launchers():
    """the transport's own rendezvous never takes MASTER_PORT (the launcher's store lives there) and
    stays inside the valid port range"""
    from shennong_amd.comm import rendezvous_port
    for master in (1024, 29500, 64518, 64519, 65535):
        port = rendezvous_port(master)
        assert port != master and 0 < port <= 65535


def test_rendezvous_port_override(monkeypatch):
    from shennong_amd.comm import rendezvous_port
    monkeypatch.setenv('SNF_COMM_PORT', '40123')
    assert rendezvous_port(29500) == 40123


def test_rendezvous_frames_are_authenticated_capped_and_plain_data(monkeypatch):
    """what crosses the rendezvous sockets: an HMAC over every frame (only holders of the job token are
    peers), a size cap checked before anything is allocated, and an unpickler that accepts plain data only"""
    import pickle
    import numpy as np
    from shennong_amd import comm
    monkeypatch.setenv('SNF_COMM_TOKEN', 'job-a')
    key_a = comm._job_key()
    monkeypatch.setenv('SNF_COMM_TOKEN', 'job-b')
    key_b = comm._job_key()
    assert key_a != key_b
    meta = [(['u0', 'u1'], [(3, 13), (0, 13)]), None, 'ValueError: x',
            {'u0': (np.arange(3) * 0.01, {'pipeline': [{'name': 'mfcc', 'columns': [0, 12]}], 'warp': np.float32(1)})}]
    left, right = socket.socketpair()
    try:
        comm._send_msg(left, pickle.dumps(meta), key_a)
        back = comm._loads(comm._recv_msg(right, key_a))
        assert back[:3] == meta[:3] and np.array_equal(back[3]['u0'][0], meta[3]['u0'][0])
        assert back[3]['u0'][1] == meta[3]['u0'][1]
        comm._send_msg(left, b'hello', key_b)                      # another job's token
        with pytest.raises(ConnectionError, match='failed authentication'):
            comm._recv_msg(right, key_a)
        comm._send_msg(left, b'x' * 64, key_a)                     # longer than the reader allows
        with pytest.raises(ConnectionError, match='exceeds the limit'):
            comm._recv_msg(right, key_a, limit=16)
    finally:
        left.close()
        right.close()

    class Payload:
        def __reduce__(self):
            return (os.getenv, ('HOME',))
    with pytest.raises(pickle.UnpicklingError, match='only plain data'):
        comm._loads(pickle.dumps(Payload()))


# ---- RCCL transport through the C ABI (needs a GPU; the test box has one: world size 1) ---------------
@pytest.mark.gpu
def test_rccl_comm_world_of_one():
    import numpy as np
    from shennong_amd import _backend, distributed
    from shennong_amd.comm import RcclComm
    if _backend.device_count() < 1:
        pytest.skip('no HIP device visible')
    comm = RcclComm(0, 1)
    assert (comm.rank, comm.world_size) == (0, 1)
    assert comm.all_gather_object({'a': 1}) == [{'a': 1}]
    stats = np.arange(24, dtype=np.float64).reshape(4, 2, 3)
    assert np.array_equal(comm.allreduce(stats, 'sum'), stats)
    assert np.array_equal(distributed.allreduce_cmvn_stats(stats, group=comm), stats)
    assert comm.allreduce(np.array([3.5]), 'max')[0] == 3.5
    comm.barrier()
    rng = np.random.default_rng(0)
    local = {'u%d' % i: rng.standard_normal((n, 5)).astype(np.float32) for i, n in enumerate((3, 0, 17))}
    merged = distributed.gather_features(local, dst=0, group=comm)
    assert merged.keys() == local.keys()
    assert all(np.array_equal(merged[k], local[k]) for k in local)
    assert distributed.gather_features({}, dst=0, group=comm) == {}
    # device pointers in and out
    data = rng.standard_normal(1000).astype(np.float32)
    d_send, d_recv = _backend.DeviceBuffer(data.nbytes), _backend.DeviceBuffer(data.nbytes)
    d_send.upload(data)
    comm.gatherv_device(d_send.ptr, data.size, d_recv.ptr, [data.size], 0)
    back = np.empty_like(data)
    d_recv.download(back)
    assert np.array_equal(back, data)
    # process_all over a world of one: the rows go from the kernel's output buffer to the "root" on the
    # device and come down once; same collection as the single-process call
    from shennong_amd import Audio, Utterances, synth
    from shennong_amd.processor import MfccProcessor
    waves = synth.ragged_utterances(77, 5, min_s=0.2, max_s=0.6)
    index = Utterances([(f'u{i}', Audio(w, 16000)) for i, w in enumerate(waves)])
    proc = MfccProcessor(dither=0)
    warps = {f'u{i}': [1.0, 0.9, 1.1][i % 3] for i in range(5)}
    assert distributed.process_all_sharded(proc, index, group=comm, vtln_warp=warps) == \
        proc.process_all(index, vtln_warp=warps)
    assert distributed.process_all_sharded(proc, index, group=comm) == proc.process_all(index)
    comm.close()


@pytest.mark.gpu
def test_extract_features_sharded_world_of_one():
    """extract_features_sharded through RcclComm on one GPU: the final blocks (two sample rates = two blocks)
    go from the pipeline's device buffers through snf_comm_gatherv and come down once; same collection as
    pipeline.extract_features"""
    from shennong_amd import Audio, Utterances, _backend, distributed, pipeline, synth
    from shennong_amd.comm import RcclComm
    from shennong_amd.logger import get_logger
    if _backend.device_count() < 1:
        pytest.skip('no HIP device visible')
    comm = RcclComm(0, 1)
    waves = synth.ragged_utterances(321, 8, min_s=0.3, max_s=0.8)
    items = [(f'u{i}', Audio(w, 16000), f's{i % 3}') for i, w in enumerate(waves)]
    items += [(f'v{i}', Audio(w[:len(w) // 2], 8000), f's{i % 3}') for i, w in enumerate(waves[:3])]
    index = Utterances(items)
    quiet = get_logger('test', 'error')
    for cfg in (pipeline.get_default_config('mfcc', with_cmvn=True, with_delta=True, with_pitch='kaldi'),
                pipeline.get_default_config('filterbank', with_cmvn=False, with_delta=False, with_pitch=False)):
        if 'mfcc' in cfg:
            cfg['mfcc']['dither'] = 0
            cfg['cmvn']['with_vad'] = False
            cfg['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
        else:
            cfg['filterbank']['dither'] = 0
        want = pipeline.extract_features(cfg, index, log=quiet)
        got = distributed.extract_features_sharded(cfg, index, group=comm, log=quiet)
        assert got.keys() == want.keys()
        for k in want.keys():
            assert got[k] == want[k], k
    comm.close()


def _rccl_pair_worker(rank, port, tmpdir):
    sys.path.insert(0, ROOT)
    import numpy as np
    from shennong_amd import _backend
    from shennong_amd.comm import RcclComm
    os.environ['SNF_COMM_TOKEN'] = 'test-%d' % port
    _backend.set_device(rank)
    comm = RcclComm(rank, 2, device=rank, port=port)
    ok = True
    # variable-length gather to each root in turn: rank r sends (r + 1) * 1000 + 7 floats of value r + 0.5
    for root in (0, 1):
        counts = [1007, 2007]
        mine = np.full(counts[rank], rank + 0.5, dtype=np.float32)
        d_send = _backend.DeviceBuffer(mine.nbytes, device=rank)
        d_send.upload(mine)
        d_recv = _backend.DeviceBuffer(4 * sum(counts), device=rank) if rank == root else None
        comm.gatherv_device(d_send.ptr, mine.size, d_recv.ptr if d_recv else None, counts, root)
        if rank == root:
            got = np.empty(sum(counts), dtype=np.float32)
            d_recv.download(got)
            ok = ok and np.array_equal(got, np.concatenate([np.full(1007, 0.5), np.full(2007, 1.5)]).astype(np.float32))
    # float64 all-reduce against numpy, sum and max
    base = np.arange(48, dtype=np.float64).reshape(4, 2, 6) / 7.0
    ok = ok and np.array_equal(comm.allreduce(base * (rank + 1), 'sum'), base * 1 + base * 2)
    ok = ok and comm.allreduce(np.array([float(rank)]), 'max')[0] == 1.0
    ok = ok and comm.all_gather_object(('r', rank)) == [('r', 0), ('r', 1)]
    merged = comm.gather_features({'u%d' % rank: np.full((3 + rank, 2), rank, np.float32)}, dst=0)
    if rank == 0:
        ok = ok and sorted(merged) == ['u0', 'u1'] and merged['u1'].shape == (4, 2) and (merged['u1'] == 1).all()
    comm.barrier()
    comm.close()
    open(os.path.join(tmpdir, f'ok{rank}'), 'w').write('1' if ok else '0')


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_rccl_two_ranks(tmp_path):
    """RCCL with two ranks (two processes, two GPUs): offsets of the variable-length gather and the float64
    all-reduce against numpy.  Skipped on a one-GPU box (the round's test box); runs wherever two GPUs are
    visible, so the 8-GPU bench is not the first execution of csrc/comm.cpp's receive loop."""
    import multiprocessing as mp
    from shennong_amd import _backend
    if _backend.device_count() < 2:
        pytest.skip('needs two GPUs')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_rccl_pair_worker, args=(r, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert [p.exitcode for p in procs] == [0, 0]
    assert [open(tmp_path / f'ok{r}').read() for r in range(2)] == ['1', '1']


def _gather_leg_worker(kind, rank, world, port, tmpdir):
    """one rank of bench.py's gather leg on CPU: the socket stand-in in host mode, a 'hot path' that fills its
    block with rank-dependent numbers; kind 'hang': the second gather of rank 1 never returns"""
    for path in (ROOT, TOOLS):
        if path not in sys.path:
            sys.path.insert(0, path)
    import json
    import ctypes as C
    import numpy as np
    import fake_comm
    import bench
    from shennong_amd import _backend
    from shennong_amd.comm import RcclComm
    lib = fake_comm.install(hang_after=1 if kind == 'hang' and rank == 1 else None)
    sys.stdout = open(os.path.join(tmpdir, f'stdout{rank}'), 'w')
    group = RcclComm(rank, world, device=0, port=port, timeout=4.0, connect=False)
    own = 1000 + 24 * rank    # (ragged blocks, as under --scaling strong)
    bufs = [_backend.DeviceBuffer(4 * own), _backend.DeviceBuffer(4 * own)]
    block = (np.arange(own, dtype=np.float32) + 1000.0 * rank)

    def run_pass(dst, stream):
        C.memmove(dst, block.ctypes.data, block.nbytes)

    def download(buf, count):
        return np.frombuffer(C.string_at(buf.ptr, 4 * int(count)), dtype=np.float32).copy()
    stream = C.c_void_p()
    lib.snf_stream_create(C.byref(stream))
    watch = bench.Watchdog()
    out = bench.gather_leg(lib, group, watch, run_pass, bufs, own, stream, _backend.DeviceBuffer, steps=3, inner=4,
                           warmup=1, job_frames=123456.0, kernel_ms=0.9, comm_timeout=1.5, download=download)
    open(os.path.join(tmpdir, f'out{rank}'), 'w').write(json.dumps([out, watch.stuck]))
    line = {'metric': 'x', 'value': 1.0, 'with_gather': out} if rank == 0 else None
    bench.finish(line, watch, group)


@pytest.mark.timeout(120)
@pytest.mark.parametrize('kind', ['fine', 'hang'])
def test_bench_gather_leg_and_watchdog(tmp_path, kind):
    """bench.py's gather leg (own stream, two buffers, counts over the sockets, EVERY rank's
    block verified by digest) at world size 2 on CPU - and the same with a transport that stalls: the leg comes
    back with an error inside its bound on both ranks, rank 0 still prints its line, and both processes exit
    although each holds a thread that will never finish"""
    import json
    import time
    t0 = time.time()
    results = [json.loads(r) for r in _run_pair(_gather_leg_worker, kind, tmp_path)]
    line = json.loads(open(tmp_path / 'stdout0').read())
    assert open(tmp_path / 'stdout1').read() == ''
    assert line['value'] == 1.0
    if kind == 'fine':
        for out, stuck in results:
            assert not stuck and 'error' not in out, out
            assert out['passes_per_step'] == 4 and out['rccl_ranks_seen'] == 2
            assert out['gather_bytes_per_pass_at_root'] == 4 * 1024
            assert set(out['predicted_ms']) == {'153_GBps_per_link', '76.5_GBps_per_link'}
        assert results[0][0]['gathered_blocks_ok'] is True and results[0][0]['gathered_blocks_checked'] == 2
        assert line['with_gather']['gathered_blocks_ok'] is True
    else:
        assert time.time() - t0 < 60
        for out, stuck in results:
            assert stuck and 'did not finish within' in out['error'], out
        assert 'TimeoutError' in line['with_gather']['error']


def test_bench_digest_and_prediction():
    for path in (ROOT, TOOLS):
        if path not in sys.path:
            sys.path.insert(0, path)
    import numpy as np
    import bench
    a = np.arange(12, dtype=np.float32)
    assert bench.block_digest(a) == bench.block_digest(a.reshape(3, 4)) != bench.block_digest(a[::-1].copy())[:1] + [0]
    assert bench.block_digest(a[:0]) == [0, 0]
    b = a.copy()
    b[5] = np.nextafter(b[5], np.float32(100))
    assert bench.block_digest(a) != bench.block_digest(b)
    # 8 ranks x 2 980 000 x 40 floats: 476.8 MB per peer = 3.12 ms at 153 GB/s, above the 0.93 ms kernel
    counts = [2980000 * 40] * 8
    pred = bench.predicted_gather_ms(counts, 0.93)
    assert abs(pred['153_GBps_per_link'] - 3.116) < 0.01 and abs(pred['76.5_GBps_per_link'] - 6.233) < 0.01
    assert bench.predicted_gather_ms([100, 100], 0.93)['153_GBps_per_link'] == 0.93


def _sharded_on_gpu_worker(rank, port, tmpdir):
    """one of two ranks that share GPU 0: the REAL device pipeline on this rank's shard, the exchange through the
    device-mode socket stand-in of snf_comm_* (device pointers in and out, staged through host memory)"""
    for path in (ROOT, TOOLS):
        if path not in sys.path:
            sys.path.insert(0, path)
    import numpy as np
    import fake_comm
    from conftest import GOLDEN
    from shennong_amd import Audio, Utterances, _backend, pipeline, synth
    from shennong_amd.comm import RcclComm
    from shennong_amd.distributed import (
        extract_features_sharded, extract_features_streamed_sharded, process_all_sharded)
    from shennong_amd.processor import MfccProcessor
    os.environ['SNF_COMM_TOKEN'] = 'gpu-stub-%d' % port
    _backend.set_device(0)
    fake = fake_comm.install(device=True)
    comm = RcclComm(rank, 2, device=0, port=port)
    wav = os.path.join(GOLDEN, 'test.wav')
    waves = synth.utterances(3, 6, 20000)
    items = [(f'w{i}', wav, f's{i % 3}', 0.1 * (i % 4), 0.1 * (i % 4) + 0.4 + 0.1 * (i % 5)) for i in range(1, 8)]
    index = Utterances(items)
    memory = Utterances([(f'm{i}', Audio(waves[i, :12000 + 1500 * i].copy(), 16000, validate=False), f's{i % 2}')
                         for i in range(6)])
    config = pipeline.get_default_config('mfcc', with_pitch='kaldi', with_cmvn=True, with_delta=True)
    config['mfcc']['dither'] = 0
    config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
    config['cmvn']['by_speaker'] = True
    ok, why = True, []

    def same(got, want, what, exact):
        good = sorted(got) == sorted(want)
        for k in want:
            if good:
                a, b = got[k].data, want[k].data
                # (a speaker's statistics are the sum of per-RANK partial sums here and of per-utterance terms in
                # the one-process run: float64 sums in another order, the normalised float32 within an ulp or two)
                good = a.shape == b.shape and (np.array_equal(a, b) if exact else np.allclose(a, b, rtol=1e-5, atol=1e-5))
                good = good and np.array_equal(got[k].times, want[k].times)
        if not good:
            why.append(what)
        return good
    for name, utts in (('files', index), ('memory', memory)):
        got = extract_features_sharded(config, utts, dst=0, group=comm)
        if rank == 0:
            want = pipeline.extract_features(config, utts)
            ok = same(got, want, 'sharded ' + name, False) and ok
            ok = ok and all(got[k].properties['pipeline'] == want[k].properties['pipeline'] for k in want)
        else:
            ok = ok and got is None
    # streamed: every rank streams its shard into its own sink; statistics summed over ranks and batches
    out = {}
    count = extract_features_streamed_sharded(config, memory, out.update, max_batch_duration=1.6, group=comm)
    names = comm.all_gather_object(sorted(out))
    ok = ok and count == len(out) and sorted(names[0] + names[1]) == sorted(u.name for u in memory)
    whole = pipeline.extract_features(config, memory)
    ok = same(out, {k: whole[k] for k in out}, 'streamed sharded', False) and ok
    # process_all over the ranks (no statistics: bit for bit)
    proc = MfccProcessor(dither=0)
    got = process_all_sharded(proc, memory, dst=1, group=comm)
    if rank == 1:
        ok = same(got, proc.process_all(memory), 'process_all sharded', True) and ok
    else:
        ok = ok and got is None
    calls = [c[0] for c in fake.calls]
    ok = ok and 'gatherv' in calls and 'allreduce' in calls
    comm.close()
    open(os.path.join(tmpdir, f'ok{rank}'), 'w').write('1' if ok else '0 %s' % why)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_sharded_pipelines_two_ranks_on_one_gpu(gpu, tmp_path):
    """extract_features_sharded / extract_features_streamed_sharded / process_all_sharded with TWO rank processes on
    the one GPU of the test box: the real device pipeline per shard, device-resident blocks handed to the
    communicator class, the by-speaker statistics reduced across ranks - everything but RCCL itself, which the
    device-mode stand-in replaces (blocks staged through host memory).  Results equal the one-process pipeline."""
    import multiprocessing as mp
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_sharded_on_gpu_worker, args=(r, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    for p in procs:
        if p.is_alive():
            p.kill()
    assert [p.exitcode for p in procs] == [0, 0]
    assert [open(tmp_path / f'ok{r}').read() for r in range(2)] == ['1', '1']
