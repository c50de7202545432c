#!/usr/bin/env python
"""Benchmark of the speech-features hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is `--inner` (default 120, `config.passes_per_step`) passes of the hot path over one batch of
synthetic input that is already resident in HBM: ``FilterbankProcessor(num_bins=40, dither=0)`` (25 ms /
10 ms, 16 kHz) over 10 000 unique synthetic 3 s utterances per GPU (BASELINE.json configs[1]; 2 980 000
frames per GPU per pass).  One pass is one 0.9 ms kernel launch: the repeat makes the timed region of the
default 20 steps ~2.2 s (long enough for the driver's GPU-busy sampler); every rate
counts every pass, `roofline.kernel_ms` is the mean HIP-event time of ONE launch.  The passes of a timed
region are enqueued on one stream through the asynchronous `*_device` entry points (a pair of `snf_event_*`
marks around each) and the region ends with one synchronisation - the way a pipeline drives the path.
With N > 1 (launched through torch.distributed.run, used as a process spawner only: torch is never
imported) the utterances shard across ranks with no data-path collective.  `--scaling weak` (default): every
GPU owns 10 000 utterances; `--scaling strong`: ONE 10 000-utterance corpus is dealt over the ranks by
`shard_utterances`.  Two values: `value` (compute only) - its barriers and its max-over-ranks go over the
job's rendezvous SOCKETS (shennong_amd.comm.RcclComm.host_barrier / host_allreduce), so nothing of it depends on
the transport the second value measures - and `value_with_gather`: a second timed region of the same K steps in
which EVERY pass is followed by the `snf_comm_gatherv` (RCCL: ncclSend / ncclRecv pairs, device pointers) of
every rank's [frames, 40] block to rank 0, on a stream of its own with two output buffers, so that the exchange
of pass k runs beside the kernel of pass k + 1 (its passes per step are sized from two probe passes so that the
region lasts about three seconds whatever the exchange costs: `with_gather.passes_per_step`); every rank's
block is verified on the root by digest; `rccl_ranks_seen` is what the RCCL communicator itself reports
(ncclCommCount).  That leg runs LAST and under a watchdog (`--comm-timeout`): RCCL's bootstrap, every enqueue
and the end of the region are bounded, and a transport that stalls costs `with_gather` (= {"error": ...}), not
the line - the process then leaves through os._exit behind the printed line instead of waiting on a stream that
will never drain.  `--transport stub` (a dress rehearsal, NEVER a measurement; the line says so): the ranks are
processes that share GPU 0 and the exchange is the socket stand-in of tests/tools/fake_comm.py staged through
host memory - everything around the collective (sharding, counts, double buffering, digests, this JSON line)
then executes on a one-GPU box.

Rank 0 prints ONE JSON line.  `value` = frames of all ranks per second of the slowest rank.
`roofline` prices the dominant kernel against the 8 TB/s HBM peak with the algorithmic bytes of
SURVEY.md §8(d) (480 B per fbank-40 frame).  `cpu_baseline` times the CPU oracle (a scalar C port of
the Kaldi algorithm; the reference's pykaldi cannot run offline) on a bounded sample of the same
utterances on this box's host cores.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = ('frames/s (25 ms/10 ms, 16 kHz) for fbank-40 + MFCC-13 at 1/2/4/8 GPUs')
BYTES_PER_FRAME = {'fbank40': 480, 'mfcc13': 372, 'mfcc13_delta': 476, 'spectrogram': 1348}
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--utts', type=int, default=10000, help='utterances per GPU')
    ap.add_argument('--seconds', type=float, default=3.0, help='utterance duration')
    ap.add_argument('--cpu-sample', type=int, default=10000,
                    help='utterances timed on the CPU oracle (0 = skip)')
    ap.add_argument('--settle', type=int, default=50,
                    help='untimed launches before the W warm-up steps: the GPU idles during the host-side '
                         'set-up and its clocks take ~30 launches to come back (10: 0.97 ms per step, '
                         '50 / 200 / 1000: 0.92 ms; gpurun_out/settle.txt)')
    ap.add_argument('--inner', type=int, default=120,
                    help='passes over the batch inside ONE step (stated in config.passes_per_step): 20 steps '
                         'of one 0.9 ms pass are an 18 ms timed region, too short for the driver\'s GPU-busy '
                         'sampler; 120 passes per step make it ~2.2 s.  Rates count every pass.')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='N > 1: weak = --utts utterances per GPU; strong = --utts utterances in all, dealt over '
                         'the ranks by shennong_amd.distributed.shard_utterances')
    ap.add_argument('--transport', choices=('rccl', 'stub'), default='rccl',
                    help='N > 1: rccl = one GPU per rank, RCCL over xGMI; stub = every rank on GPU 0, the exchange '
                         'through tests/tools/fake_comm.py (rehearsal of the N > 1 code path on a one-GPU box; the '
                         'line is labelled and is not a measurement)')
    ap.add_argument('--comm-timeout', type=float, default=60.0,
                    help='seconds any single wait on the transport may take (sockets, RCCL bootstrap, an enqueue); '
                         'the gather region as a whole gets 30 s + twice its estimated duration')
    ap.add_argument('--stream-hours', type=float, default=125.0,
                    help='hours of audio of the streamed-pipeline leg (BASELINE config 5: 1 000 h / 8 GPUs = 125 h '
                         'per GPU; the 10 000 waves of the batch are reused round robin, 1 000 speakers)')
    ap.add_argument('--no-extra', action='store_true',
                    help='skip the MFCC / spectrogram side measurements')
    return ap.parse_args()


def effective_cores():
    """Host cores this process may actually use: min(affinity, cgroup cpu.max quota)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def make_batch(first_id, count, nsamples):
    """Unique seeded utterances, generated by a small process pool (untimed)."""
    import numpy as np
    from concurrent.futures import ProcessPoolExecutor
    from shennong_amd import synth
    workers = max(1, min(effective_cores() // max(1, int(os.environ.get('WORLD_SIZE', '1'))), 16))
    chunk = (count + workers - 1) // workers
    jobs = [(first_id + i, min(chunk, count - i)) for i in range(0, count, chunk)]
    if workers == 1 or count < 64:
        parts = [synth.utterances(a, c, nsamples) for a, c in jobs]
    else:
        with ProcessPoolExecutor(workers) as pool:
            parts = list(pool.map(_gen, [(a, c, nsamples) for a, c in jobs]))
    return np.concatenate(parts, axis=0)


def _gen(args):
    from shennong_amd import synth
    return synth.utterances(*args)


def _gen_ids(args):
    import numpy as np
    from shennong_amd import synth
    ids, nsamples = args
    return np.concatenate([synth.utterances(i, 1, nsamples) for i in ids], axis=0)


def make_batch_ids(ids, nsamples):
    """The utterances `ids` of the seeded corpus (a rank's shard of it), same generator as make_batch"""
    import numpy as np
    from concurrent.futures import ProcessPoolExecutor
    workers = max(1, min(effective_cores() // max(1, int(os.environ.get('WORLD_SIZE', '1'))), 16))
    chunk = max(1, (len(ids) + workers - 1) // workers)
    jobs = [(list(ids[i:i + chunk]), nsamples) for i in range(0, len(ids), chunk)]
    if workers == 1 or len(ids) < 64:
        parts = [_gen_ids(j) for j in jobs]
    else:
        with ProcessPoolExecutor(workers) as pool:
            parts = list(pool.map(_gen_ids, jobs))
    return np.ascontiguousarray(np.concatenate(parts, axis=0))


def cpu_baseline(opts, waves, nthreads):
    """frames/s of the CPU oracle on `waves` [n, nsamples] with one utterance per task over `nthreads`
    POSIX threads (the reference's model: joblib threads over utterances, processor/base.py:104-107)."""
    from oracle import oracle as orc
    orc.lib()
    t0 = time.perf_counter()
    out = orc.compute_batch(opts, waves, nthreads)
    dt = time.perf_counter() - t0
    return out.shape[0] / dt, dt, out.shape[0]


def end_to_end(plan, waves, frames_per_utt, chunk=500, reps=3, n_streams=3):
    """PCIe-inclusive rate of the same workload (SURVEY.md 8d "end-to-end over the batch"): the int16
    waves start in page-locked HOST memory, the float32 features end there.  The batch is cut into
    chunks that go round `n_streams` streams: chunk i + 1 is uploaded while the kernel of chunk i runs and
    its features come down (copy engines and compute overlap; 500 utterances x 3 streams measured best: 21.2 ms
    per 10 000 utterances against 24.7 with 1 000 x 2; the link alone would need 16.8).  Never the headline `value`."""
    import ctypes as C
    import numpy as np
    from shennong_amd import _backend
    L = _backend.lib()
    n_utts, nsamples = waves.shape
    chunk = min(chunk, n_utts)
    n_chunks = n_utts // chunk
    ndims = plan.ndims
    in_bytes, out_bytes = chunk * nsamples * 2, chunk * frames_per_utt * ndims * 4
    h_in, h_out = C.c_void_p(), C.c_void_p()
    _backend.check(L.snf_host_malloc(C.byref(h_in), n_chunks * in_bytes))
    _backend.check(L.snf_host_malloc(C.byref(h_out), n_chunks * out_bytes))
    C.memmove(h_in, waves.ctypes.data, n_chunks * in_bytes)
    streams, d_in, d_out = [], [], []
    for _ in range(n_streams):
        st = C.c_void_p()
        _backend.check(L.snf_stream_create(C.byref(st)))
        streams.append(st)
        d_in.append(_backend.DeviceBuffer(in_bytes))
        d_out.append(_backend.DeviceBuffer(out_bytes))
    soff = np.arange(chunk + 1, dtype=np.int64) * nsamples
    foff = np.arange(chunk + 1, dtype=np.int64) * frames_per_utt

    def one_pass():
        for i in range(n_chunks):
            k = i % n_streams
            _backend.check(L.snf_memcpy_h2d_async(
                C.c_void_p(d_in[k].ptr), C.c_void_p(h_in.value + i * in_bytes), in_bytes, streams[k]))
            plan.run_device(d_in[k].ptr, soff, foff, d_out[k].ptr, stream=streams[k].value)
            _backend.check(L.snf_memcpy_d2h_async(
                C.c_void_p(h_out.value + i * out_bytes), C.c_void_p(d_out[k].ptr), out_bytes, streams[k]))
        for st in streams:
            _backend.check(L.snf_stream_synchronize(st))
    one_pass()
    t0 = time.perf_counter()
    for _ in range(reps):
        one_pass()
    dt = (time.perf_counter() - t0) / reps
    first = np.empty((frames_per_utt, ndims), dtype=np.float32)
    C.memmove(first.ctypes.data, h_out, first.nbytes)
    for k in range(n_streams):
        d_in[k].free()
        d_out[k].free()
        L.snf_stream_destroy(streams[k])
    L.snf_host_free(h_in)
    L.snf_host_free(h_out)
    frames = n_chunks * chunk * frames_per_utt
    return {'frames_per_s': frames / dt, 'wall_ms': dt * 1e3, 'chunks': n_chunks,
            'utterances_per_chunk': chunk, 'host_to_device_GBps': n_chunks * in_bytes / dt / 1e9,
            'device_to_host_GBps': n_chunks * out_bytes / dt / 1e9, 'finite': bool(np.isfinite(first).all()),
            'streams': n_streams,
            # the two directions run side by side and the upload carries twice the bytes: it is the bound one
            # (the `*_GBps` above are each direction's bytes over the WHOLE wall clock, not link rates)
            'link_floor_ms': max(n_chunks * in_bytes, n_chunks * out_bytes) / 57e9 * 1e3,
            'note': 'page-locked host buffers, %d streams: upload / kernel / download of consecutive '
                    'chunks overlap; PCIe-inclusive, never the headline value; host_to_device_GBps / '
                    'device_to_host_GBps = the bytes of a direction over the whole wall clock (the upload, '
                    '2 B per sample against 1.6 B per sample of features, is the direction that binds: '
                    '`link_floor_ms` at 57 GB/s)' % n_streams}


# ---- N > 1: the one collective of the job, measured under a watchdog ------------------------------------------
XGMI_LINK_GBPS = (153.0, 76.5)   # one xGMI link, GB/s: the figure quoted for MI355X (7 links x ~153 GB/s per GPU) and
                                 # half of it (should that figure count both directions); the gather is priced at both


class Watchdog:
    """Calls that may depend on OTHER processes (RCCL's bootstrap, an enqueue into a transport whose peer died,
    the end of a region with a collective in it) run on a thread that can be abandoned: ctypes releases the
    interpreter lock, the caller waits `seconds` and no longer.  Once something has been abandoned the process
    holds a thread - possibly a HIP stream - that will never finish: `stuck` tells the caller to leave through
    os._exit when it has printed what it has."""
    def __init__(self, bind=None):
        self.stuck = False
        self.bind = bind   # called first on every such thread: hipSetDevice is per thread, a fresh one is on GPU 0

    def call(self, fn, seconds, what):
        import threading
        box = {}

        def target():
            try:
                if self.bind is not None:
                    self.bind()
                box['value'] = fn()
            except BaseException as exc:   # noqa: BLE001 (handed to the caller below)
                box['error'] = exc
        thread = threading.Thread(target=target, name='snf-bench-watch', daemon=True)
        thread.start()
        thread.join(seconds)
        if thread.is_alive():
            self.stuck = True
            raise TimeoutError('%s did not finish within %.0f s' % (what, seconds))
        if 'error' in box:
            raise box['error']
        return box.get('value')


class Marks:
    """pairs of HIP events (snf_event_*) around the passes of one timed region"""
    def __init__(self, L, n):
        import ctypes as C
        from shennong_amd import _backend
        self.L, self.pairs = L, []
        for _ in range(n):
            a, b = C.c_void_p(), C.c_void_p()
            _backend.check(L.snf_event_create(C.byref(a)))
            _backend.check(L.snf_event_create(C.byref(b)))
            self.pairs.append((a, b))

    def ms(self):
        import ctypes as C
        from shennong_amd import _backend
        out, ms = [], C.c_float()
        for a, b in self.pairs:
            _backend.check(self.L.snf_event_elapsed_ms(a, b, C.byref(ms)))
            out.append(float(ms.value))
        return out

    def free(self):
        for a, b in self.pairs:
            self.L.snf_event_destroy(a)
            self.L.snf_event_destroy(b)


def block_digest(block):
    """(sum, xor) of the 32-bit patterns of a float32 block: what a rank says about the block it sent and what
    the root finds where that block should have landed"""
    import numpy as np
    bits = np.ascontiguousarray(block).reshape(-1).view(np.uint32)
    if bits.size == 0:
        return [0, 0]
    return [int(bits.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(bits))]


def predicted_gather_ms(counts, kernel_ms, root=0):
    """Per pass, from link arithmetic: every peer sends its block over its OWN xGMI link to the
    root, the links run side by side, the exchange of pass k runs beside the kernel of pass k + 1 - a pass costs
    max(kernel, largest peer block / link rate); the root's HBM takes the sum of the blocks at a small fraction
    of its 8 TB/s.  One figure per assumed link rate."""
    peer_bytes = max([4 * c for r, c in enumerate(counts) if r != root] or [0])
    return {'%g_GBps_per_link' % bw: max(float(kernel_ms), peer_bytes / (bw * 1e9) * 1e3) for bw in XGMI_LINK_GBPS}


def gather_leg(L, group, watch, run_pass, bufs, own_floats, stream, DeviceBuffer, steps, inner, warmup,
               job_frames, kernel_ms, comm_timeout, download):
    """`value_with_gather`: K steps in which every pass is followed by the gather of every rank's block to rank 0.

    `group`: an RcclComm made with connect=False (its sockets are up); `run_pass(dst_ptr, stream)` enqueues one
    pass of the hot path into a device block of `own_floats` floats; `bufs`: two such blocks (DeviceBuffer);
    `download(DeviceBuffer, count) -> float32 array`.  Returns the `with_gather` object of the line: numbers, or
    {'error': ...} - whatever happens in here (an exception, a transport that stalls: see Watchdog) the caller
    keeps what it measured before."""
    import ctypes as C
    import numpy as np
    from shennong_amd import _backend
    rank, world = group.rank, group.world_size
    check = _backend.check
    d_all = cstream = None
    events = []
    try:
        watch.call(group.connect, comm_timeout, 'RCCL bootstrap (ncclCommInitRank)')
        counts = [int(c) for c in group.host_allreduce(
            np.array([float(own_floats) if r == rank else 0.0 for r in range(world)]), 'sum')]
        if rank == 0:
            d_all = DeviceBuffer(max(sum(counts) * 4, 16))
        cstream = C.c_void_p()
        check(L.snf_stream_create(C.byref(cstream)))
        for _ in range(4):
            ev = C.c_void_p()
            check(L.snf_event_create(C.byref(ev)))
            events.append(ev)
        ready, drained = events[:2], events[2:]
        used = [False, False]

        def enqueue(count, marks=None):
            """`count` passes: kernel of pass k on the compute stream into buffer k mod 2 (once that buffer's last
            exchange has drained), its exchange on the communication stream behind it - beside the next kernel"""
            for k in range(count):
                b = k & 1
                if used[b]:
                    check(L.snf_stream_wait_event(stream, drained[b]))
                if marks is not None:
                    L.snf_event_record(marks.pairs[k][0], stream)
                run_pass(bufs[b].ptr, stream)
                if marks is not None:
                    L.snf_event_record(marks.pairs[k][1], stream)
                check(L.snf_event_record(ready[b], stream))
                check(L.snf_stream_wait_event(cstream, ready[b]))
                group.gatherv_device(bufs[b].ptr, own_floats, d_all.ptr if d_all else None, counts, 0,
                                     stream=cstream.value)
                check(L.snf_event_record(drained[b], cstream))
                used[b] = True

        def drain(seconds, what):
            """both streams empty, or TimeoutError: polled (snf_stream_query), never a blocking wait on work
            that a peer may never complete"""
            deadline = time.perf_counter() + seconds
            for st in (stream, cstream):
                while True:
                    rc = L.snf_stream_query(st)
                    if rc == 0:
                        break
                    if rc < 0:
                        check(rc)
                    if time.perf_counter() > deadline:
                        watch.stuck = True
                        raise TimeoutError('%s did not finish within %.0f s' % (what, seconds))
                    time.sleep(0.0002)

        def barrier():
            check(L.snf_device_synchronize())
            group.host_barrier()

        # passes per step of THIS region: as many as the compute-only region when the exchange is cheap, fewer
        # when it is not (7 x 477 MB per pass must not turn the run into minutes): two probe passes, the slowest
        # rank's time, a region of about three seconds
        def probe():
            enqueue(1)
            drain(comm_timeout, 'the first gather')
            barrier()
            t0 = time.perf_counter()
            enqueue(2)
            drain(comm_timeout, 'the probe gathers')
            return (time.perf_counter() - t0) / 2
        probe_s = watch.call(probe, 3 * comm_timeout, 'the probe passes of the gather')
        probe_s = float(group.host_allreduce(np.array([probe_s]), 'max')[0])
        g_inner = int(max(1, min(inner, 3.0 / (steps * max(probe_s, 1e-6)))))
        budget = 30.0 + 2.0 * probe_s * g_inner * (steps + min(warmup, 2))

        def region():
            for _ in range(min(warmup, 2)):
                enqueue(min(g_inner, 8))
            drain(budget, 'the warm-up of the gather region')
            marks = Marks(L, steps * g_inner)
            barrier()
            t0 = time.perf_counter()
            enqueue(steps * g_inner, marks)
            drain(budget, 'the gather region')
            dt = time.perf_counter() - t0
            ms = marks.ms()
            marks.free()
            return dt, ms
        g_elapsed, g_kernel_ms = watch.call(region, budget + 10.0, 'the gather region')
        g_elapsed = float(group.host_allreduce(np.array([g_elapsed]), 'max')[0])
        out = {'value': job_frames * g_inner * steps / g_elapsed,
               'ms_per_step': g_elapsed / steps * 1e3,
               'ms_per_pass': g_elapsed / steps / g_inner * 1e3,
               'passes_per_step': g_inner, 'probe_ms_per_pass': probe_s * 1e3,
               'gather_bytes_per_pass_at_root': int(sum(counts) - counts[0]) * 4,
               'kernel_ms': float(np.mean(g_kernel_ms)),
               'predicted_ms': predicted_gather_ms(counts, kernel_ms),
               'overlap': 'the exchange of pass k (own stream, two output buffers) runs beside the kernel of pass k + 1',
               'rccl_ranks_seen': group.ranks_seen()}
        # what arrived: EVERY rank's block, whole - each rank digests the block it sent (the two buffers hold the
        # same features), the root digests what lies where that block should have landed
        digests = group.all_gather_object(block_digest(download(bufs[(steps * g_inner - 1) & 1], own_floats)))
        if rank == 0:
            got = download(d_all, sum(counts))
            found, pos = [], 0
            for r in range(world):
                found.append(block_digest(got[pos:pos + counts[r]]))
                pos += counts[r]
            out['gathered_blocks_ok'] = bool(found == [list(d) for d in digests])
            out['gathered_blocks_checked'] = world
        return out
    except BaseException as exc:   # noqa: BLE001 (the compute-only value must not be lost with the collective)
        if isinstance(exc, (KeyboardInterrupt, SystemExit)):
            raise
        return {'error': '%s: %s' % (type(exc).__name__, exc), 'watchdog_abandoned_a_call': watch.stuck}
    finally:
        if not watch.stuck:   # (a stream that never drains is not waited for, its buffers are not given back)
            for ev in events:
                L.snf_event_destroy(ev)
            if cstream is not None:
                L.snf_stream_destroy(cstream)
            if d_all is not None:
                d_all.free()


def finish(line, watch, comm):
    """Rank 0 prints THE line; then every rank leaves - orderly when nothing was abandoned, through os._exit
    when the watchdog gave up on a call into the transport: a thread, possibly a HIP stream, of this process
    will then never finish, and neither would the interpreter's or the HIP runtime's shutdown behind it"""
    if line is not None:
        print(json.dumps(line), flush=True)
    if watch.stuck:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    if comm is not None:
        try:
            comm.host_barrier()   # (nobody closes the sockets under a peer that is still reading them)
        except Exception:  # noqa: BLE001 (a peer that left through the watchdog's exit)
            pass
        comm.close()


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import numpy as np
    from shennong_amd import _abi, _backend
    from shennong_amd.processor import (
        FilterbankProcessor, MfccProcessor, SpectrogramProcessor)
    from shennong_amd.postprocessor import DeltaPostProcessor

    if _backend.device_count() < 1:
        raise SystemExit('bench.py needs an MI355X: no HIP device visible')
    stub = args.transport == 'stub'
    device = 0 if stub else local_rank   # (stub: a rehearsal, every rank is a process on GPU 0)
    _backend.set_device(device)
    comm = None
    watch = Watchdog(bind=_backend.bind_device)
    if world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ:  # launched by torch.distributed.run
        if stub:
            sys.path.insert(0, os.path.join(ROOT, 'tests', 'tools'))
            import fake_comm
            # (SNF_STUB_HANG_AFTER=n on rank 1: its n + 1-th gather never returns - the watchdog's rehearsal)
            hang = os.environ.get('SNF_STUB_HANG_AFTER')
            fake_comm.install(device=True, hang_after=int(hang) if hang and rank == 1 else None)
        from shennong_amd.comm import RcclComm
        # the sockets only: RCCL itself is brought up by the gather leg, last and under the watchdog.  The sockets
        # wait long (ranks reach a barrier tens of seconds apart when the host is busy generating 8 x 10 000
        # utterances): what bounds a stalled TRANSPORT is the watchdog, not these timeouts
        comm = RcclComm.from_env(device=device, connect=False, timeout=max(args.comm_timeout, 600.0))

    nsamples = int(args.seconds * 16000)
    if args.scaling == 'strong' and world > 1:
        # ONE corpus of --utts utterances for the whole job: every rank takes its shard of it (equal lengths:
        # the greedy partition deals them round robin) and generates only those
        from shennong_amd.distributed import shard_utterances
        mine = shard_utterances([nsamples] * args.utts, world)[rank]
        n_utts = len(mine)
        waves = make_batch_ids(mine, nsamples)
    else:
        n_utts = args.utts
        waves = make_batch(rank * n_utts, n_utts, nsamples)          # [n_utts, nsamples] int16
    fbank = FilterbankProcessor(num_bins=40, dither=0)
    opts = fbank._build_options()
    plan = _backend.get_plan(opts)
    frames_per_utt = plan.num_frames(nsamples)
    total_frames = frames_per_utt * n_utts
    soff = np.arange(n_utts + 1, dtype=np.int64) * nsamples
    foff = np.arange(n_utts + 1, dtype=np.int64) * frames_per_utt

    d_wave = _backend.DeviceBuffer(waves.nbytes)
    d_wave.upload(waves)
    d_out = _backend.DeviceBuffer(total_frames * 40 * 4)

    def sync_all():
        # (the barrier goes over the rendezvous sockets, not over RCCL: see the module docstring)
        _backend.check(_backend.lib().snf_device_synchronize())
        if comm is not None:
            comm.host_barrier()
            _backend.check(_backend.lib().snf_device_synchronize())

    inner = max(1, args.inner)

    # The timed passes are ENQUEUED on one stream without waiting for each (the *_device entry points are
    # asynchronous on a caller's stream: this is how a pipeline drives the path); every pass sits between
    # two timing marks on that stream, read after the one synchronisation that ends the timed region.
    import ctypes as C
    L = _backend.lib()
    stream = C.c_void_p()
    _backend.check(L.snf_stream_create(C.byref(stream)))

    def timed_region(p, d_dst):
        """K steps of `inner` passes of plan `p` between two barriers + synchronisations -> (seconds of the
        slowest rank, kernel ms of every pass)"""
        marks = Marks(L, args.steps * inner)
        sync_all()
        t0 = time.perf_counter()
        for a, b in marks.pairs:
            L.snf_event_record(a, stream)
            p.run_device(d_wave.ptr, soff, foff, d_dst.ptr, stream=stream.value)
            L.snf_event_record(b, stream)
        _backend.check(L.snf_stream_synchronize(stream))
        dt = time.perf_counter() - t0
        if comm is not None:
            dt = float(comm.host_allreduce(np.array([dt]), 'max')[0])
        sync_all()
        ms = marks.ms()
        marks.free()
        return dt, ms

    def one_pass():
        plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)

    def step():  # one step = `inner` passes of the hot path over the resident batch
        for _ in range(inner):
            plan.run_device(d_wave.ptr, soff, foff, d_out.ptr, stream=stream.value)

    for _ in range(args.settle):  # settle clocks / first-touch before the W warmup steps (untimed)
        one_pass()
    for _ in range(args.warmup):
        step()
    elapsed, kernel_ms = timed_region(plan, d_out)

    ms_per_step = elapsed / args.steps * 1e3
    # frames of the whole job per pass: N x 10 000 utterances (weak) or the one corpus (strong)
    job_frames = total_frames * world
    if comm is not None and world > 1:
        job_frames = int(comm.host_allreduce(np.array([float(total_frames)]), 'sum')[0])
    value = job_frames * inner * args.steps / elapsed
    kms = float(np.mean(kernel_ms))

    # ---- the other half of the metric: MfccProcessor() (13 cepstra) over the same batch, timed the same
    # way right behind the fbank-40 steps (the same K steps between the same barriers) -----------------
    mfcc_plan = _backend.get_plan(MfccProcessor(dither=0)._build_options())
    d_mfcc = _backend.DeviceBuffer(total_frames * 13 * 4)
    for _ in range(args.settle):  # (the GPU idled while the plan was made: settle like the fbank steps)
        mfcc_plan.run_device(d_wave.ptr, soff, foff, d_mfcc.ptr)
    for _ in range(args.warmup * inner):
        mfcc_plan.run_device(d_wave.ptr, soff, foff, d_mfcc.ptr, stream=stream.value)
    mfcc_elapsed, mfcc_kernel_ms = timed_region(mfcc_plan, d_mfcc)
    value_mfcc13 = job_frames * inner * args.steps / mfcc_elapsed
    mfcc_kms = float(np.mean(mfcc_kernel_ms))
    achieved = total_frames * BYTES_PER_FRAME['fbank40'] / (kms * 1e-3) / 1e9

    # ---- parity spot-check of the timed output against the oracle (rank 0, first utterances) ----
    extra = {}
    if rank == 0:
        from oracle import oracle as orc
        got = np.empty((frames_per_utt * 2, 40), dtype=np.float32)
        d_out.download(got)
        want = np.concatenate([orc.compute(opts, waves[i]) for i in range(2)])
        err = np.abs(got - want) / np.maximum(np.abs(want), 1.0)
        extra['parity_max_rel_err_vs_oracle'] = float(err.max())

    # ---- side measurements (not the headline value).  N = 1 only: they describe one GPU, and N ranks
    # running them side by side would only fight over the host's cores (pipeline legs, copy threads) ----
    if world > 1:
        extra['side_measurements'] = 'N = 1 only (python bench.py --gpus 1)'
    if not args.no_extra and world == 1:
        def time_plan(p, run, frames, nbytes, reps=10, settle=10):
            # (the GPU idles while the host prepares each case: like the headline loop, settle the
            # clocks with untimed launches first - the first launches after an idle period run ~25 %
            # slower)
            for _ in range(settle):
                run()
            _backend.check(_backend.lib().snf_device_synchronize())
            t0 = time.perf_counter()
            ks = []
            for _ in range(reps):
                run()
                ks.append(p.last_kernel_ms(0))
            dt = (time.perf_counter() - t0) / reps
            # the kernels of the last call: name and HIP-event time of each, the dominant one first
            slots = []
            for k in range(1, 9):
                name = p.kernel_name(k)
                if name is None:
                    break
                slots.append((float(p.last_kernel_ms(k)), name))
            slots.sort(reverse=True)
            return {'frames_per_s': frames / dt, 'ms_per_step': dt * 1e3,
                    'kernel_ms': float(np.mean(ks)),
                    'kernel': slots[0][1] if slots else None,
                    'kernels_ms': {name: round(ms, 4) for ms, name in slots},
                    'algorithmic_bytes_per_frame': nbytes,
                    'hbm_frac': frames * nbytes / (np.mean(ks) * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # fbank-40 on the round-2 kernel (fbank512_kernel) on the same box, same run: the launcher reads
        # SNF_FBANK512_OLD at every call
        os.environ['SNF_FBANK512_OLD'] = '1'
        try:
            extra['fbank40_round2_kernel'] = time_plan(plan, one_pass, total_frames, BYTES_PER_FRAME['fbank40'])
        finally:
            del os.environ['SNF_FBANK512_OLD']
        extra['fbank40_round3_kernel'] = time_plan(plan, one_pass, total_frames, BYTES_PER_FRAME['fbank40'])
        extra['fbank40_round2_kernel']['note'] = ('fbank512_kernel (the per-utterance / centred / VTLN forms run on '
                                                  'it), same batch, calls waited for one by one')
        extra['fbank40_round3_kernel']['note'] = ('fbank512b_kernel, the kernel of the headline `value` (flat '
                                                  'batches; its round-3 structure with the round-4 arithmetic), '
                                                  'timed like the side lines: calls waited for one by one')
        # the reference's DEFAULT frame options: dither = 1.0 (shennong/processor/base.py:122); the noise is
        # drawn in the kernel (counter-based generator, device_fft.h), same batch, same buffers
        dith_plan = _backend.get_plan(FilterbankProcessor(num_bins=40)._build_options())
        extra['fbank40_dither1'] = time_plan(
            dith_plan, lambda: dith_plan.run_device(d_wave.ptr, soff, foff, d_out.ptr),
            total_frames, BYTES_PER_FRAME['fbank40'])
        extra['fbank40_dither1']['over_dither0'] = (
            extra['fbank40_dither1']['kernel_ms'] / extra['fbank40_round3_kernel']['kernel_ms'])
        one_pass()  # (d_out holds the dither-0 features again)
        extra['mfcc13'] = time_plan(
            mfcc_plan, lambda: mfcc_plan.run_device(d_wave.ptr, soff, foff, d_mfcc.ptr),
            total_frames, BYTES_PER_FRAME['mfcc13'])
        # fbank-80 (a common front end of neural recognisers): banks of 65 ... 128 bins run the 64-bin kernel
        # twice, over the two halves of the bank (round 6; the generic kernel until then, 7 x slower per frame)
        p80 = _backend.get_plan(FilterbankProcessor(num_bins=80, dither=0)._build_options())
        d_o80 = _backend.DeviceBuffer(total_frames * 80 * 4)
        extra['fbank80'] = time_plan(
            p80, lambda: p80.run_device(d_wave.ptr, soff, foff, d_o80.ptr), total_frames, 320 + 4 * 80)
        d_o80.free()
        # Kaldi's "hires" MFCC (40 bins, 40 cepstra): more than the 16 cepstra of the fused form - the filterbank
        # kernel writes [log energy | log-mel] rows, mfcc_dct_kernel forms the cepstra (round 6; the generic kernel
        # until then, 14 ms per 2.98 M frames)
        p4040 = _backend.get_plan(MfccProcessor(num_bins=40, num_ceps=40, dither=0)._build_options())
        d_o4040 = _backend.DeviceBuffer(total_frames * 40 * 4)
        extra['mfcc40_hires'] = time_plan(
            p4040, lambda: p4040.run_device(d_wave.ptr, soff, foff, d_o4040.ptr), total_frames, 320 + 4 * 40)
        d_o4040.free()
        # the DCT-II of the same plan as an MFMA chain instead of the vector-pipe form that ships (the
        # table layout is chosen when the plan is built: a private plan outside the cache)
        os.environ['SNF_DCT_MFMA'] = '1'
        try:
            mfma_plan = _backend.Plan(MfccProcessor(dither=0)._build_options())
        finally:
            del os.environ['SNF_DCT_MFMA']
        extra['mfcc13_dct_mfma'] = time_plan(
            mfma_plan, lambda: mfma_plan.run_device(d_wave.ptr, soff, foff, d_mfcc.ptr),
            total_frames, BYTES_PER_FRAME['mfcc13'])
        del mfma_plan
        delta_plan = _backend.get_plan(DeltaPostProcessor()._build_options())
        d_delta = _backend.DeviceBuffer(total_frames * 39 * 4)
        extra['delta_13_to_39'] = time_plan(
            delta_plan, lambda: delta_plan.run_post_device(d_mfcc.ptr, 13, foff, d_delta.ptr),
            total_frames, 208)
        # BASELINE config 3: MFCC-13 + delta + delta-delta through one plan (append_deltas): the shipped
        # form runs the MFCC kernel and the delta kernel back to back (580 B/frame: the [T, 13] block makes
        # one round trip through HBM), the one-launch form of round 2 (476 B/frame) is SNF_FUSED_DELTA=1
        chain_opts = MfccProcessor(dither=0)._build_options()
        chain_opts.append_deltas = 1
        chain_plan = _backend.get_plan(chain_opts)
        extra['mfcc13_delta'] = time_plan(
            chain_plan, lambda: chain_plan.run_device(d_wave.ptr, soff, foff, d_delta.ptr),
            total_frames, BYTES_PER_FRAME['mfcc13_delta'])
        os.environ['SNF_FUSED_DELTA'] = '1'
        try:
            fused_plan = _backend.Plan(chain_opts)
        finally:
            del os.environ['SNF_FUSED_DELTA']
        extra['mfcc13_delta_fused'] = time_plan(
            fused_plan, lambda: fused_plan.run_device(d_wave.ptr, soff, foff, d_delta.ptr),
            total_frames, BYTES_PER_FRAME['mfcc13_delta'])
        del fused_plan
        d_delta.free()
        extra['end_to_end'] = end_to_end(plan, waves, frames_per_utt)
        spec_utts = n_utts  # BASELINE config 2: all 10 000 utterances (3.06 GB of output)
        spec_plan = _backend.get_plan(SpectrogramProcessor(dither=0)._build_options())
        d_spec = _backend.DeviceBuffer(frames_per_utt * spec_utts * 257 * 4)
        extra['spectrogram257'] = time_plan(
            spec_plan, lambda: spec_plan.run_device(
                d_wave.ptr, soff[:spec_utts + 1], foff[:spec_utts + 1], d_spec.ptr),
            frames_per_utt * spec_utts, BYTES_PER_FRAME['spectrogram'])
        d_spec.free()
        # PLP-13 and Kaldi pitch on the whole batch (BASELINE's size; the pitch tracker keeps 19 GB of
        # NCCF / back-pointer scratch in HBM for it)
        from shennong_amd.processor import PlpProcessor, KaldiPitchProcessor
        sub = n_utts
        plp_plan = _backend.get_plan(PlpProcessor(dither=0)._build_options())
        d_plp = _backend.DeviceBuffer(frames_per_utt * sub * 13 * 4)
        extra['plp13'] = time_plan(
            plp_plan, lambda: plp_plan.run_device(
                d_wave.ptr, soff[:sub + 1], foff[:sub + 1], d_plp.ptr),
            frames_per_utt * sub, 372, reps=5)
        d_plp.free()
        pitch_plan = _backend.get_plan(KaldiPitchProcessor()._build_options())
        pf = pitch_plan.num_frames(nsamples)
        pfoff = np.arange(sub + 1, dtype=np.int64) * pf
        d_pitch = _backend.DeviceBuffer(pf * sub * 2 * 4)
        extra['pitch'] = time_plan(
            pitch_plan, lambda: pitch_plan.run_device(
                d_wave.ptr, soff[:sub + 1], pfoff, d_pitch.ptr),
            pf * sub, 328, reps=3, settle=2)
        d_pitch.free()
        # the whole pipeline (BASELINE config 5: fbank + pitch + delta + CMVN by speaker), device
        # resident between the stages; wall clock of extract_features, i.e. INCLUDING the one upload,
        # the one download and the per-utterance Python bookkeeping (Features objects, properties)
        from shennong_amd import Audio, Utterances, pipeline
        from shennong_amd.logger import get_logger
        pn = min(n_utts, 1000)
        index = Utterances([(f'u{i}', Audio(waves[i], 16000, validate=False), f's{i % 20}')
                            for i in range(pn)])
        cfg = pipeline.get_default_config('filterbank', with_pitch='kaldi', with_cmvn=True,
                                          with_delta=True)
        cfg['filterbank']['num_bins'] = 40
        cfg['filterbank']['dither'] = 0
        cfg['cmvn']['with_vad'] = False
        quiet = get_logger('bench', 'error')
        def time_pipeline(idx):
            pipeline.extract_features(cfg, idx, log=quiet)
            walls = []
            for _ in range(5):   # (one call is 8-14 ms of host and device work: the median of five)
                t0 = time.perf_counter()
                feats = pipeline.extract_features(cfg, idx, log=quiet)
                walls.append(time.perf_counter() - t0)
            return float(np.median(walls)), min(walls), feats
        dt, dt_min, feats = time_pipeline(index)
        dt_pin, dt_pin_min, _ = time_pipeline(index.pin())
        nfr = sum(f.nframes for f in feats.values())
        cols = int(next(iter(feats.values())).ndims)
        extra['pipeline_fbank_pitch_delta_cmvn'] = {
            'frames_per_s': nfr / dt, 'wall_s': dt, 'wall_s_min': dt_min, 'calls': 5,
            'pinned_wall_s': dt_pin, 'pinned_wall_s_min': dt_pin_min, 'pinned_frames_per_s': nfr / dt_pin,
            'link_floor_s': (pn * nsamples * 2 + nfr * cols * 4) / 57e9,
            'utterances': pn, 'columns': cols,
            'note': 'extract_features on 1 000 in-memory utterances, host to host; `pinned_*`: the same index after '
                    'Utterances.pin(); the upload, the pitch tracker (~3 ms, the features beside it) and the '
                    'download follow each other: `link_floor_s` is the two copies at 57 GB/s'}
        del feats

        # the same pipeline streamed in bounded batches at ONE GPU's share of BASELINE config 5: 1 000 h / 8 =
        # 125 h = 150 000 utterances of 3 s (the 10 000 waves of the batch reused round robin: host synthesis
        # is not what is measured; every utterance has its own name, 1 000 speakers round robin), CMVN by
        # speaker with the reference's default VAD-weighted statistics (pipeline.py:584-596): a first pass
        # (features + energy + VAD + statistics) that leaves the uploaded audio in HBM, features recomputed in
        # the second, results handed to a counting sink and dropped; default batches, one warm-up corpus
        import resource
        cfg5 = pipeline.get_default_config('filterbank', with_pitch='kaldi', with_cmvn=True, with_delta=True)
        cfg5['filterbank']['num_bins'] = 40
        cfg5['filterbank']['dither'] = 0
        cfg5['cmvn']['by_speaker'] = True     # (with_vad stays True: the reference's default)
        n5 = max(1200, int(round(args.stream_hours * 3600.0 / args.seconds)))
        audios = [Audio(waves[i], 16000, validate=False) for i in range(n_utts)]
        index2 = Utterances([(f'u{i:06d}', audios[i % n_utts], f's{i % 1000:04d}') for i in range(n5)])
        hours = n5 * args.seconds / 3600.0
        # the corpus loaded ONCE into one page-locked int16 block (Utterances.pin(): what a loader that decodes
        # files does for free; 14.4 GB for 125 h, untimed like the file reads it stands for): batches go up from
        # where they lie, the upload of batch k + 1 beside the kernels of batch k
        t0 = time.perf_counter()
        pinned5 = index2.pin()
        pin_s = time.perf_counter() - t0
        warm = Utterances(list(index2)[:min(n5, 2 * 4800)])   # (two batches of the default size: pools at size)
        pipeline.extract_features_streamed(cfg5, warm.pin(), lambda f: None, log=quiet)
        pipeline.extract_features_streamed(cfg5, warm, lambda f: None, log=quiet)
        del warm
        _, hbm_total = _backend.mem_info()

        def run_streamed(index, njobs, watch_memory=False):
            seen = {'utts': 0, 'batches': 0, 'peak_device': 0, 'frames': 0, 'mem_info_s': 0.0}

            def counting_sink(feats):
                seen['utts'] += len(feats)
                seen['batches'] += 1
                seen['frames'] += sum(f.nframes for f in feats.values())
                if watch_memory:   # (hipMemGetInfo costs milliseconds: sampled in the run that is not the headline)
                    t0 = time.perf_counter()
                    free, total = _backend.mem_info()
                    seen['mem_info_s'] += time.perf_counter() - t0
                    seen['peak_device'] = max(seen['peak_device'], total - free)
            run_stats = pipeline.RunStats()
            t0 = time.perf_counter()
            written = pipeline.extract_features_streamed(cfg5, index, counting_sink, log=quiet, njobs=njobs,
                                                         stats=run_stats)
            return time.perf_counter() - t0, written, seen, run_stats.as_dict()
        tried = {}
        for njobs in (1, 2, 3):
            tried[njobs] = run_streamed(pinned5, njobs)
        best = min(tried, key=lambda k: tried[k][0])
        dt, written, seen, cost = tried[best]
        dt_page, _, seen_page, cost_page = run_streamed(index2, 1, watch_memory=True)
        link = 57e9   # bytes per second and direction the host link of this box family sustains (end_to_end above)
        extra['pipeline_streamed'] = {
            'hours_of_audio': hours, 'hours_of_audio_per_s': hours / dt, 'wall_s': dt,
            'frames_per_s': seen['frames'] / dt, 'njobs': best,
            'wall_s_by_njobs': {str(k): v[0] for k, v in tried.items()},
            'gpu_s': cost['gpu_ms'] / 1e3,
            'link_floor_s': {'overlapped': max(cost['bytes_up'], cost['bytes_down']) / link,
                             'serial': (cost['bytes_up'] + cost['bytes_down']) / link,
                             'bytes_up': cost['bytes_up'], 'bytes_down': cost['bytes_down'],
                             'assumed_bytes_per_s_per_direction': link},
            'upload_wait_s': cost['upload_wait_s'], 'download_wait_s': cost['download_wait_s'],
            'audio': 'Utterances.pin(): one page-locked int16 block (%.1f GB, page-locked in %.1f s, untimed)' % (
                n5 * nsamples * 2 / 1e9, pin_s),
            'pageable_hours_of_audio_per_s': hours / dt_page, 'pageable_wall_s': dt_page,
            'pageable_upload_wait_s': cost_page['upload_wait_s'],
            'utterances': n5, 'utterances_written': int(written), 'every_utterance_once': seen['utts'] == n5,
            'speakers': 1000, 'cmvn': 'by speaker, VAD-weighted (reference default)', 'columns': 123,
            'batches': seen['batches'], 'batch_s': pipeline.default_batch_duration(best),
            'peak_device_bytes': int(seen_page['peak_device']), 'device_total_bytes': int(hbm_total),
            'pageable_mem_info_s': seen_page['mem_info_s'],
            'host_max_rss_bytes': int(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss) * 1024,
            'passes': 2, 'audio_resident_between_passes': True,
            'note': 'BASELINE config 5 at one GPU\'s share (1 000 h / 8); wall clock includes the uploads, both '
                    'passes, the downloads and the per-utterance Features objects; `njobs` batches in flight '
                    '(the reference\'s parameter; best of 1 / 2 / 3); `pageable_*`: the same corpus as 150 000 '
                    'separate numpy arrays (gathered into staging memory batch by batch), njobs 1; `gpu_s`: HIP '
                    'events of every plan call of the run'}
        del index2, audios, pinned5

        # a corpus on DISK, in and out: 4 000 of the utterances written as 16-bit WAV files,
        # read back natively (snf_wav_read_pcm16: side by side into page-locked memory) by process_all and by the
        # streamed config-5 pipeline, whose batches go to a Kaldi archive of float matrices (KaldiStreamWriter)
        import shutil
        import tempfile
        import scipy.io.wavfile
        from shennong_amd.serializers import KaldiStreamWriter
        nfiles = min(n_utts, 4000)
        where = tempfile.mkdtemp(prefix='snf_bench_')
        try:
            for i in range(nfiles):
                scipy.io.wavfile.write(os.path.join(where, 'u%05d.wav' % i), 16000, waves[i])
            t0 = time.perf_counter()
            on_disk = Utterances([('u%05d' % i, os.path.join(where, 'u%05d.wav' % i), 's%03d' % (i % 50))
                                  for i in range(nfiles)])
            index_s = time.perf_counter() - t0
            fbank.process_all(on_disk)
            walls = []
            for _ in range(3):
                t0 = time.perf_counter()
                coll = fbank.process_all(on_disk)
                walls.append(time.perf_counter() - t0)
                del coll
            t0 = time.perf_counter()
            with KaldiStreamWriter(os.path.join(where, 'corpus.ark'), double=False) as writer:
                written = pipeline.extract_features_streamed(cfg5, on_disk, writer.write, njobs=2, log=quiet)
            corpus_s = time.perf_counter() - t0
            hours_disk = nfiles * args.seconds / 3600.0
            extra['corpus_on_disk'] = {
                'files': nfiles, 'hours_of_audio': hours_disk, 'index_us_per_file': index_s / nfiles * 1e6,
                'process_all_us_per_file': min(walls) / nfiles * 1e6,
                'process_all_hours_of_audio_per_s': hours_disk / min(walls),
                'wav_to_archive_s': corpus_s, 'wav_to_archive_hours_of_audio_per_s': hours_disk / corpus_s,
                'archive_bytes': os.path.getsize(os.path.join(where, 'corpus.ark')), 'utterances_written': int(written),
                'note': '16-bit mono WAV files (page cache warm) -> FilterbankProcessor.process_all; and -> the '
                        'streamed config-5 pipeline (CMVN by speaker, two passes, two batches in flight) -> '
                        'KaldiStreamWriter(double=False); directory: ' + tempfile.gettempdir()}
        finally:
            shutil.rmtree(where, ignore_errors=True)

        # FeaturesProcessor.process_all on the 10 000 in-memory utterances (the north_star's API surface,
        # reference processor/base.py:56-107): wall clock from utterances to a FeaturesCollection, host to host:
        # upload (960 MB: PCIe floor 22.4 ms at 42.8 GB/s) + kernel + download (477 MB) + one Features per
        # utterance.  Two forms of the same call: `Utterances.pin()` (the audio was loaded once into ONE
        # page-locked block - what a loader that decodes files can do for free -: pieces go up from where they
        # are) and plain pageable numpy arrays (every byte is first gathered into page-locked staging memory: the
        # call is then bound by the host's memory traffic, ~3.4 GB per call, not by the link).
        pa_index = Utterances([(f'u{i:05d}', Audio(waves[i], 16000, validate=False)) for i in range(n_utts)])

        def time_process_all(index, reps=7):
            import gc
            coll = fbank.process_all(index)
            walls = []
            for _ in range(reps):
                del coll       # (giving back the previous result - 10 000 Features, a 477 MB page-locked block - is
                gc.collect(0)  # not part of making the next one; the collector's young pass over them neither)
                t0 = time.perf_counter()
                coll = fbank.process_all(index)
                walls.append(time.perf_counter() - t0)
            assert len(coll) == n_utts
            return float(np.median(walls)), min(walls)
        pa_pinned = pa_index.pin()
        dt, dt_min = time_process_all(pa_pinned)
        dt_page, dt_page_min = time_process_all(pa_index)
        extra['process_all'] = {
            'frames_per_s': total_frames / dt, 'wall_ms': dt * 1e3, 'wall_ms_min': dt_min * 1e3,
            'utterances': n_utts, 'audio': 'Utterances.pin(): one page-locked int16 block, loaded once',
            'pageable_frames_per_s': total_frames / dt_page, 'pageable_wall_ms': dt_page * 1e3,
            'pageable_wall_ms_min': dt_page_min * 1e3,
            'pcie_floor_ms': n_utts * nsamples * 2 / 42.8e9 * 1e3,
            'note': 'FilterbankProcessor(num_bins=40, dither=0).process_all(utterances) -> FeaturesCollection, '
                    'host to host through the public API; `pageable_*`: the same call on plain numpy arrays'}
        del pa_pinned
        del pa_index
        # 8 kHz audio (256-sample frames): two frames per 16-lane row (fbank256x2_kernel)
        sub8 = min(n_utts, 4000)
        w8 = np.ascontiguousarray(waves[:sub8, :nsamples // 2])
        d_w8 = _backend.DeviceBuffer(w8.nbytes)
        d_w8.upload(w8)
        p8 = _backend.get_plan(FilterbankProcessor(sample_rate=8000, num_bins=40, dither=0)._build_options())
        f8 = p8.num_frames(nsamples // 2)
        soff8 = np.arange(sub8 + 1, dtype=np.int64) * (nsamples // 2)
        foff8 = np.arange(sub8 + 1, dtype=np.int64) * f8
        d_o8 = _backend.DeviceBuffer(f8 * sub8 * 40 * 4)
        extra['fbank40_8khz'] = time_plan(
            p8, lambda: p8.run_device(d_w8.ptr, soff8, foff8, d_o8.ptr), f8 * sub8, 160 + 2 * 80)
        extra['fbank40_8khz']['kernel'] = p8.kernel_name(1)
        d_w8.free()
        d_o8.free()
        # 44.1 kHz audio (1102-sample frames, 2048-point transform): the register-resident long-frame
        # kernel; 2000 x 3 s = 596 000 frames (the waves are only bytes here, re-cut to 132 300 samples)
        sub44 = min(n_utts, 2000)
        n44 = 132300
        w44 = np.ascontiguousarray(waves.reshape(-1)[:sub44 * n44].reshape(sub44, n44))
        d_w44 = _backend.DeviceBuffer(w44.nbytes)
        d_w44.upload(w44)
        p44 = _backend.get_plan(FilterbankProcessor(sample_rate=44100, num_bins=40, dither=0)._build_options())
        f44 = p44.num_frames(n44)
        soff44 = np.arange(sub44 + 1, dtype=np.int64) * n44
        foff44 = np.arange(sub44 + 1, dtype=np.int64) * f44
        d_o44 = _backend.DeviceBuffer(f44 * sub44 * 40 * 4)
        extra['fbank40_44khz'] = time_plan(
            p44, lambda: p44.run_device(d_w44.ptr, soff44, foff44, d_o44.ptr), f44 * sub44, 2 * 441 + 160)
        extra['fbank40_44khz']['kernel'] = p44.kernel_name(1)
        d_w44.free()
        d_o44.free()
        # 32 kHz audio (800-sample frames, padded to 1024): two frames per 1024-point complex transform
        # (fbank1024x2_kernel, round 6; until then the zero-extended 2048-point transform); 2000 x 3 s
        sub32 = min(n_utts, 2000)
        n32 = 96000
        w32 = np.ascontiguousarray(waves.reshape(-1)[:sub32 * n32].reshape(sub32, n32))
        d_w32 = _backend.DeviceBuffer(w32.nbytes)
        d_w32.upload(w32)
        p32 = _backend.get_plan(FilterbankProcessor(sample_rate=32000, num_bins=40, dither=0)._build_options())
        f32 = p32.num_frames(n32)
        soff32 = np.arange(sub32 + 1, dtype=np.int64) * n32
        foff32 = np.arange(sub32 + 1, dtype=np.int64) * f32
        d_o32 = _backend.DeviceBuffer(f32 * sub32 * 40 * 4)
        extra['fbank40_32khz'] = time_plan(
            p32, lambda: p32.run_device(d_w32.ptr, soff32, foff32, d_o32.ptr), f32 * sub32, 2 * 320 + 160)
        extra['fbank40_32khz']['kernel'] = p32.kernel_name(1)
        d_w32.free()
        d_o32.free()

    d_mfcc.free()
    # ---- CPU baseline (rank 0, N = 1 only) -----------------------------------------------------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        ncores = effective_cores()
        n_all = min(n_utts, max(args.cpu_sample, 32 * ncores))
        n_one = min(n_utts, max(64, args.cpu_sample // 8))
        cpu_baseline(opts, waves[:min(n_utts, ncores)], ncores)  # warm the thread pool / page cache
        v_all, dt_all, fr = cpu_baseline(opts, waves[:n_all], ncores)
        v_one, dt_one, fr_one = cpu_baseline(opts, waves[:n_one], 1)
        cpu = {'value': v_all, 'unit': 'frames/s', 'cores': ncores, 'kind': 'port',
               'sample': '%d of the %d synthetic 3 s utterances (%d frames, %.2f s), fbank-40 '
                         'dither 0, one utterance per task on %d POSIX threads; single thread on '
                         '%d utterances: %.0f frames/s (%.2f s)' % (
                             n_all, n_utts, fr, dt_all, ncores, n_one, v_one, dt_one)}
        extra['cpu_single_thread_frames_per_s'] = v_one
        extra['gpu_over_cpu_all_cores'] = value / v_all
        if 'process_all' in extra:   # the public API, host to host, against the same CPU baseline
            extra['process_all']['over_cpu_all_cores'] = extra['process_all']['frames_per_s'] / v_all
            extra['process_all']['pageable_over_cpu_all_cores'] = extra['process_all']['pageable_frames_per_s'] / v_all

    # HBM bytes per launch from the committed PMC passes (profiles/hbm_traffic.json), same workload only
    traffic = None
    traffic_note = None
    valu = None
    try:
        with open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json')) as fh:
            tr = json.load(fh)
        if tr['workload'] == {'kind': 'fbank40', 'utterances': n_utts, 'seconds': args.seconds}:
            traffic = tr['traffic_bytes_per_launch'] / (kms * 1e-3) / 1e9
            if 'valu' in tr:
                # report achieved/VALU beside achieved/HBM - the kernel is bound by
                # the f32 vector pipe (and LDS), not by HBM
                rate = tr['valu']['wave_instrs_per_launch'] / (kms * 1e-3)
                valu = {'achieved': rate / 1e9, 'peak': tr['valu']['peak_wave_instrs_per_s'] / 1e9,
                        'unit': 'G wave-instr/s', 'frac': rate / tr['valu']['peak_wave_instrs_per_s'],
                        'note': 'SQ_INSTS_VALU per launch (separate --pmc pass, profiles/hbm_traffic.json) / this '
                                'run\'s kernel time vs the data-sheet issue rate (1 wave64 instruction '
                                'per 2 clocks per SIMD at 2.4 GHz)'}
            traffic_note = ('%.3f GB per launch from separate rocprofv3 --pmc passes (2 x FETCH_SIZE + '
                            'WRITE_SIZE, profiles/hbm_traffic.json) / this run\'s kernel time; '
                            'algorithmic %.3f GB' % (tr['traffic_bytes_per_launch'] / 1e9,
                                                     total_frames * BYTES_PER_FRAME['fbank40'] / 1e9))
    except (OSError, KeyError, ValueError):
        pass

    # ---- the same K steps with the job's ONE collective inside the timed region: behind every pass the gather
    # of every rank's [frames, 40] block to rank 0 (snf_comm_gatherv: ncclSend / ncclRecv pairs, every peer over
    # its own xGMI link).  LAST, under the watchdog: nothing above depends on the transport ----
    gathered = None
    if comm is not None:
        d_out2 = _backend.DeviceBuffer(total_frames * 40 * 4)

        def download(buf, count):
            host = np.empty(int(count), dtype=np.float32)
            if host.size:
                buf.download(host)
            return host
        gathered = gather_leg(
            L, comm, watch, lambda dst, st: plan.run_device(d_wave.ptr, soff, foff, dst, stream=st.value),
            [d_out, d_out2], total_frames * 40, stream, _backend.DeviceBuffer, args.steps, inner, args.warmup,
            job_frames, kms, args.comm_timeout, download)
        if stub:
            gathered['transport'] = 'stub'

    if rank == 0:
        line = {
            'metric': METRIC, 'value': value, 'value_mfcc13': value_mfcc13, 'unit': 'frames/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'settle': args.settle, 'ms_per_step': ms_per_step,
            'ms_per_pass': ms_per_step / inner,
            'ms_per_step_mfcc13': mfcc_elapsed / args.steps * 1e3,
            'value_with_gather': gathered.get('value') if gathered else None,
            'rccl_ranks_seen': gathered.get('rccl_ranks_seen') if gathered else None,
            'with_gather': gathered,
            'higher_is_better': True, 'scaling': args.scaling if world > 1 else 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'FilterbankProcessor(num_bins=40, dither=0) log-mel, 25 ms/10 ms, '
                            '16 kHz, %d unique synthetic %.1f s utterances per GPU (%d frames), '
                            'int16 waves resident in HBM' % (n_utts, args.seconds, total_frames),
                'utterances_per_gpu': n_utts, 'frames_per_gpu': total_frames, 'frames_per_pass_all_gpus': job_frames,
                'passes_per_step': inner, 'frames_per_gpu_per_step': total_frames * inner,
                'untimed_launches_before_the_timed_region': args.settle + args.warmup * inner,
                'launches': 'the K x %d passes of a timed region are enqueued on one stream (asynchronous '
                            '*_device calls), one synchronisation at its end; a pair of HIP events around '
                            'every pass gives roofline.kernel_ms' % inner,
                'parallelism': 'utterance-sharded x%d, no data-path collective in `value`; `value_with_gather` '
                               'adds the gather of every Features block to rank 0 behind every pass (own stream, two buffers)' % world,
                'device': _backend.device_name(device)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_note': traffic_note, 'valu': valu,
                         'kernel': plan.kernel_name(1), 'kernel_ms': kms,
                         'algorithmic_bytes_per_frame': BYTES_PER_FRAME['fbank40']},
            'roofline_mfcc13': {
                'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'achieved': total_frames * BYTES_PER_FRAME['mfcc13'] / (mfcc_kms * 1e-3) / 1e9,
                'frac': total_frames * BYTES_PER_FRAME['mfcc13'] / (mfcc_kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'kernel': mfcc_plan.kernel_name(1), 'kernel_ms': mfcc_kms,
                'algorithmic_bytes_per_frame': BYTES_PER_FRAME['mfcc13']},
            'cpu_baseline': cpu,
            'extra': extra}
        if stub:
            # a dress rehearsal: N processes time-slice ONE GPU and the exchange crosses PCIe twice and a socket
            line['transport'] = ('stub: %d processes share GPU 0, snf_comm_* replaced by the host-staged socket '
                                 'stand-in of tests/tools/fake_comm.py' % world)
            line['not_a_measurement'] = True
    finish(line if rank == 0 else None, watch, comm)


if __name__ == '__main__':
    main()
