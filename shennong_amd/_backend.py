"""ctypes loader and thin call layer over ``libshennong_hip.so`` (the C ABI in
``include/shennong_amd.h``).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible, every compute
call raises.  Host-only helpers (frame counts, window function) work without a GPU.
"""

import ctypes as C
import os
import threading

import numpy as np

from shennong_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SHENNONG_AMD_LIB: developer knob for A/B runs of another build of the same ABI)
_LIB_PATH = os.environ.get('SHENNONG_AMD_LIB') or os.path.join(_HERE, 'libshennong_hip.so')
_LIB = None
_LOCK = threading.Lock()
_PLANS = {}
_DEVICE = int(os.environ.get('SHENNONG_AMD_DEVICE', '0'))

# every entry point declared in include/shennong_amd.h (checked by tests/test_abi.py)
EXPORTS = [
    'snf_version', 'snf_last_error', 'snf_device_count', 'snf_set_device',
    'snf_device_name', 'snf_device_synchronize', 'snf_num_frames',
    'snf_first_sample_of_frame', 'snf_window_size', 'snf_window_shift',
    'snf_padded_window_size', 'snf_window_function', 'snf_pitch_num_frames',
    'snf_plan_create', 'snf_plan_destroy', 'snf_plan_ndims', 'snf_plan_fast_path',
    'snf_plan_num_frames', 'snf_plan_run_batch', 'snf_plan_run_batch_device',
    'snf_post_ndims', 'snf_post_run_batch', 'snf_post_run_batch_device',
    'snf_cmvn_accumulate', 'snf_cmvn_apply', 'snf_cmvn_accumulate_device',
    'snf_cmvn_apply_device', 'snf_concat_columns_device', 'snf_count_nonfinite_device', 'snf_set_noise_call',
    'snf_malloc', 'snf_free', 'snf_memcpy_h2d', 'snf_memcpy_d2h', 'snf_memset',
    'snf_host_malloc', 'snf_host_free', 'snf_debug_fill_lds', 'snf_debug_pitch_scratch',
    'snf_stream_create', 'snf_stream_destroy', 'snf_stream_synchronize', 'snf_memcpy_h2d_async',
    'snf_memcpy_d2h_async', 'snf_comm_unique_id', 'snf_comm_init', 'snf_comm_rank', 'snf_comm_world_size',
    'snf_comm_destroy', 'snf_comm_gatherv', 'snf_comm_allreduce_f64',
    'snf_plan_last_kernel_ms', 'snf_plan_kernel_name', 'snf_set_oom_hook',
    'snf_event_create', 'snf_event_destroy', 'snf_event_record', 'snf_event_elapsed_ms', 'snf_mem_info',
    'snf_stream_wait_event', 'snf_stream_query', 'snf_event_synchronize', 'snf_wav_scan', 'snf_wav_read_pcm16']


_OOM_HOOK_TYPE = C.CFUNCTYPE(None)


def _release_pool():
    try:
        DEVICE_POOL.clear()
    except Exception:  # pragma: nocover  (a callback must not raise into C)
        pass


_OOM_HOOK = _OOM_HOOK_TYPE(_release_pool)


def lib():
    """Loads the HIP library once; raises loudly when it has not been built"""
    global _LIB
    if _LIB is not None:
        return _LIB
    with _LOCK:
        if _LIB is not None:
            return _LIB
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f'{_LIB_PATH} not found: build it with `make -C '
                f'{os.path.join(_HERE, "csrc")}` (or python -c "import '
                f'__graft_entry__ as g; g.build()"). shennong_amd has no CPU '
                f'fallback.')
        L = C.CDLL(_LIB_PATH)
        i64, i32, f32, vp = C.c_int64, C.c_int32, C.c_float, C.c_void_p
        pf, pi64 = C.POINTER(C.c_float), C.POINTER(C.c_int64)
        pi16 = C.POINTER(C.c_int16)
        FO, PO = C.POINTER(_abi.FrameOptions), C.POINTER(_abi.PitchOptions)
        OP = C.POINTER(_abi.Options)
        L.snf_version.restype = C.c_char_p
        L.snf_last_error.restype = C.c_char_p
        L.snf_device_name.argtypes = [i32, C.c_char_p, i32]
        L.snf_set_device.argtypes = [i32]
        L.snf_num_frames.argtypes = [FO, i64]
        L.snf_num_frames.restype = i64
        L.snf_first_sample_of_frame.argtypes = [FO, i64]
        L.snf_first_sample_of_frame.restype = i64
        L.snf_window_size.argtypes = [FO]
        L.snf_window_shift.argtypes = [FO]
        L.snf_padded_window_size.argtypes = [FO]
        L.snf_window_function.argtypes = [FO, pf]
        L.snf_pitch_num_frames.argtypes = [PO, i64]
        L.snf_pitch_num_frames.restype = i64
        L.snf_plan_create.argtypes = [OP, i32, C.POINTER(vp)]
        L.snf_plan_destroy.argtypes = [vp]
        L.snf_plan_destroy.restype = None
        L.snf_plan_ndims.argtypes = [vp]
        L.snf_plan_fast_path.argtypes = [vp]
        L.snf_plan_num_frames.argtypes = [vp, i64]
        L.snf_plan_num_frames.restype = i64
        L.snf_plan_run_batch.argtypes = [vp, pi16, pi64, i64, pf, pf, pi64]
        L.snf_plan_run_batch_device.argtypes = [
            vp, vp, pi64, i64, pf, vp, pi64, vp]
        L.snf_post_ndims.argtypes = [vp, i32]
        L.snf_post_run_batch.argtypes = [vp, pf, i32, pi64, i64, pf]
        L.snf_post_run_batch_device.argtypes = [vp, vp, i32, pi64, i64, vp, vp]
        pi32, pf64 = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        L.snf_cmvn_accumulate.argtypes = [
            vp, pf, i32, pi64, i64, pf, pi32, i32, pf64]
        L.snf_cmvn_apply.argtypes = [
            vp, pf, i32, pi64, i64, pf64, pi32, i32, i32, i32, pf]
        L.snf_cmvn_accumulate_device.argtypes = [
            vp, vp, i32, pi64, i64, vp, pi32, i32, pf64]
        L.snf_cmvn_apply_device.argtypes = [
            vp, vp, i32, pi64, i64, pf64, pi32, i32, i32, i32, vp]
        L.snf_concat_columns_device.argtypes = [
            i32, vp, i32, pi64, vp, i32, pi64, i64, vp, pi64]
        L.snf_count_nonfinite_device.argtypes = [i32, vp, C.c_uint64, C.POINTER(C.c_uint64)]
        L.snf_set_noise_call.argtypes = [C.c_uint64]
        L.snf_malloc.argtypes = [C.POINTER(vp), C.c_uint64]
        L.snf_free.argtypes = [vp]
        L.snf_memcpy_h2d.argtypes = [vp, vp, C.c_uint64]
        L.snf_memcpy_d2h.argtypes = [vp, vp, C.c_uint64]
        L.snf_memset.argtypes = [vp, i32, C.c_uint64]
        L.snf_host_malloc.argtypes = [C.POINTER(vp), C.c_uint64]
        L.snf_host_free.argtypes = [vp]
        L.snf_debug_fill_lds.argtypes = [C.c_uint32]
        L.snf_debug_pitch_scratch.argtypes = [vp] + [C.POINTER(vp)] * 4
        L.snf_stream_create.argtypes = [C.POINTER(vp)]
        L.snf_stream_destroy.argtypes = [vp]
        L.snf_stream_synchronize.argtypes = [vp]
        L.snf_stream_query.argtypes = [vp]
        L.snf_memcpy_h2d_async.argtypes = [vp, vp, C.c_uint64, vp]
        L.snf_event_create.argtypes = [C.POINTER(vp)]
        L.snf_event_destroy.argtypes = [vp]
        L.snf_event_record.argtypes = [vp, vp]
        L.snf_stream_wait_event.argtypes = [vp, vp]
        L.snf_event_synchronize.argtypes = [vp]
        L.snf_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
        L.snf_memcpy_d2h_async.argtypes = [vp, vp, C.c_uint64, vp]
        L.snf_comm_unique_id.argtypes = [vp]
        L.snf_comm_init.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
        L.snf_comm_rank.argtypes = [vp]
        L.snf_comm_world_size.argtypes = [vp]
        L.snf_comm_destroy.argtypes = [vp]
        L.snf_comm_gatherv.argtypes = [vp, vp, i64, vp, pi64, i32, vp]
        L.snf_comm_allreduce_f64.argtypes = [vp, vp, i64, i32, vp]
        L.snf_plan_last_kernel_ms.argtypes = [vp, i32]
        L.snf_plan_last_kernel_ms.restype = f32
        L.snf_plan_kernel_name.argtypes = [vp, i32]
        L.snf_plan_kernel_name.restype = C.c_char_p
        L.snf_set_oom_hook.argtypes = [_OOM_HOOK_TYPE]
        L.snf_mem_info.argtypes = [C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.snf_wav_scan.argtypes = [C.c_char_p, pi32, pi32, pi64, pi32, pi32]
        L.snf_wav_read_pcm16.argtypes = [C.POINTER(C.c_char_p), i64, pi64, pi64, pi16, pi64, i32, pi32]
        # the library's own allocations (plan scratch: ~19 GB for a 10 000-utterance pitch batch) reclaim
        # what DEVICE_POOL has parked before they give up
        L.snf_set_oom_hook(_OOM_HOOK)
        _LIB = L
    return _LIB


def check(rc):
    """Maps an ABI return code to the exception class the reference raises"""
    if rc == _abi.SNF_OK:
        return
    msg = lib().snf_last_error().decode(errors='replace')
    if rc == _abi.SNF_E_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def device_count():
    return int(lib().snf_device_count())


def set_device(device_id):
    """Selects the GPU used by plans and device buffers created afterwards (one process per GPU)
    and makes it the calling thread's current HIP device"""
    global _DEVICE
    _DEVICE = int(device_id)
    check(lib().snf_set_device(_DEVICE))


def bind_device(device_id=None):
    """hipSetDevice is per thread: every allocation / copy made outside a plan call binds the
    thread to the selected GPU first (a fresh thread starts on device 0)"""
    check(lib().snf_set_device(_DEVICE if device_id is None else int(device_id)))


def get_device():
    return _DEVICE


def mem_info(device_id=None):
    """(free, total) bytes of HBM on the selected GPU; the blocks parked in DEVICE_POOL count as used"""
    bind_device(device_id)
    free, total = C.c_uint64(0), C.c_uint64(0)
    check(lib().snf_mem_info(C.byref(free), C.byref(total)))
    return int(free.value), int(total.value)


def device_name(device_id=None):
    buf = C.create_string_buffer(256)
    check(lib().snf_device_name(
        _DEVICE if device_id is None else device_id, buf, 256))
    return buf.value.decode()


# ---- host-only helpers -------------------------------------------------------
def num_frames(frame_opts, nsamples):
    return int(lib().snf_num_frames(C.byref(frame_opts), int(nsamples)))


def first_sample_of_frame(frame_opts, frame):
    return int(lib().snf_first_sample_of_frame(C.byref(frame_opts), int(frame)))


def window_size(frame_opts):
    return int(lib().snf_window_size(C.byref(frame_opts)))


def window_shift(frame_opts):
    return int(lib().snf_window_shift(C.byref(frame_opts)))


def padded_window_size(frame_opts):
    return int(lib().snf_padded_window_size(C.byref(frame_opts)))


def window_function(frame_opts):
    out = np.zeros(max(window_size(frame_opts), 0), dtype=np.float32)
    check(lib().snf_window_function(
        C.byref(frame_opts), out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def pitch_num_frames(pitch_opts, nsamples):
    return int(lib().snf_pitch_num_frames(C.byref(pitch_opts), int(nsamples)))


# ---- plans -------------------------------------------------------------------
class Plan:
    """An immutable device-resident feature plan (window, mel banks, DCT... in HBM)"""
    def __init__(self, opts, device=None, quiet=False):
        self.opts = _abi.Options.from_buffer_copy(bytes(opts))
        self.device = _DEVICE if device is None else int(device)
        handle = C.c_void_p()
        check(lib().snf_plan_create(
            C.byref(self.opts), self.device, C.byref(handle)))
        self.handle = handle
        # the noise stream of a call (dither, delta-pitch noise) is numbered HERE, one number per call of this
        # plan whichever way the call takes - a small batch, the pieces of a large one (which run on clones),
        # a device-resident call - so that two calls never share a stream (see snf_set_noise_call)
        self._noise_calls = 0
        self._noise_lock = threading.Lock()
        if not quiet and lib().snf_plan_fast_path(handle) == 0:
            from shennong_amd.logger import get_logger
            get_logger('backend', 'warning').warning(
                'this option combination is not covered by the register-resident kernels (even frames '
                'that pad to 128 ... 512 samples with <= 64 mel bins, <= 16 cepstra and a power spectrum; '
                'frames that pad to 1024 samples, or even ones that pad to 2048, with <= 128 mel bins): the '
                'generic wave-per-frame kernel is used, 5 to 8 times slower per frame')

    def __del__(self):
        handle = getattr(self, 'handle', None)
        if handle and _LIB is not None:
            try:
                _LIB.snf_plan_destroy(handle)
            except Exception:  # pragma: nocover
                pass
            self.handle = None

    def _next_noise_call(self):
        with self._noise_lock:
            self._noise_calls += 1
            return self._noise_calls

    @property
    def ndims(self):
        return int(lib().snf_plan_ndims(self.handle))

    def num_frames(self, nsamples):
        return int(lib().snf_plan_num_frames(self.handle, int(nsamples)))

    def post_ndims(self, in_cols):
        return int(lib().snf_post_ndims(self.handle, int(in_cols)))

    def last_kernel_ms(self, which=0):
        return float(lib().snf_plan_last_kernel_ms(self.handle, which))

    def kernel_name(self, which):
        name = lib().snf_plan_kernel_name(self.handle, which)
        return name.decode() if name else None

    # -- Audio -> Features --
    def run(self, waves, vtln_warps=None, check_finite=False, wrap=None):
        """`waves`: list of 1-D int16 arrays -> list of float32 [nframes, ndims]; `check_finite`:
        ONE validation of the whole batch (what Features.validate checks per utterance).  With `wrap` the
        call returns ``wrap(matrices)`` instead - the caller's per-utterance objects, which a large batch
        builds while its pieces are still on their way (they only need to know WHERE their rows will be)"""
        if wrap is not None:
            return self._run(waves, vtln_warps, check_finite, wrap)
        return self._run(waves, vtln_warps, check_finite, lambda res: res)

    def _run(self, waves, vtln_warps, check_finite, wrap):
        n = len(waves)
        i16 = np.dtype(np.int16)
        if any(w.dtype != i16 or w.ndim != 1 for w in waves):
            waves = [np.ascontiguousarray(w, dtype=np.int16).reshape(-1) for w in waves]
        lengths = np.fromiter((w.shape[0] for w in waves), np.int64, n)
        soff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lengths, out=soff[1:])
        frames_of = {int(x): self.num_frames(int(x)) for x in np.unique(lengths)}
        nfr = np.fromiter((frames_of[int(x)] for x in lengths), np.int64, n)
        foff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(nfr, out=foff[1:])
        warp = None
        if vtln_warps is not None:
            warp = np.ascontiguousarray(vtln_warps, dtype=np.float32)
            if warp.shape[0] != n:
                raise ValueError('one vtln_warp per utterance is required')
            if np.all(warp == 1.0):
                warp = None
        if n == 0:
            return wrap([])
        cuts = self._tracker_chunks(soff)
        if cuts is not None:   # (a corpus whose tracker scratch would not fit: whole calls one after the other)
            res = []
            for a, b in zip(cuts, cuts[1:]):
                res += self._run(waves[a:b], None, check_finite, lambda part: part)
            return wrap(res)
        if int(soff[-1]) * 2 >= _LARGE_BATCH_BYTES and self.ndims > 0:
            return self._run_large(waves, soff, foff, nfr, warp, check_finite, wrap)
        wave, wave_token = stage_rows(waves, np.int16)
        # the per-utterance results are views of ONE array (cutting 4 000 fresh copies out of it
        # cost more than the launch and both transfers together)
        out = result_array((int(foff[-1]), self.ndims), np.float32)
        try:
            lib().snf_set_noise_call(self._next_noise_call())   # (thread-local, used up by the call below)
            check(lib().snf_plan_run_batch(
                self.handle, wave.ctypes.data_as(C.POINTER(C.c_int16)),
                soff.ctypes.data_as(C.POINTER(C.c_int64)), n,
                warp.ctypes.data_as(C.POINTER(C.c_float)) if warp is not None
                else None,
                out.ctypes.data_as(C.POINTER(C.c_float)),
                foff.ctypes.data_as(C.POINTER(C.c_int64))))
            if check_finite:
                _check_finite(out)
            res = []
            for u in range(n):
                if nfr[u] == 0:
                    # Kaldi returns an empty (0, 0) matrix when no frame fits
                    res.append(np.zeros((0, 0), dtype=np.float32))
                elif n == 1:
                    res.append(out)
                else:
                    res.append(out[foff[u]:foff[u + 1]])
        finally:
            del wave
            STAGING.release(wave_token)
        return wrap(res)

    def _tracker_chunks(self, soff):
        """Utterance cuts of a pitch batch that is too long for ONE call - the tracker keeps ~2.3 GB of scratch per
        hour of audio (back pointers and the resampled NCCF of every frame: 19 GB per 10 000 x 3 s), per plan and
        grow-only - or None when the batch fits `_MAX_TRACKER_HOURS` (or is not a pitch batch)"""
        if self.opts.kind != _abi.KIND_PITCH:
            return None
        limit = int(_MAX_TRACKER_HOURS * 3600.0 * float(self.opts.pitch.samp_freq))
        if int(soff[-1]) <= limit:
            return None
        cuts, n = [0], soff.shape[0] - 1
        while cuts[-1] < n:
            a = cuts[-1]
            b = int(np.searchsorted(soff, soff[a] + limit, side='right')) - 1
            cuts.append(min(max(b, a + 1), n))   # (an utterance longer than the limit is a call of its own)
        return cuts if len(cuts) > 2 else None

    def _clones(self, count):
        """(lock, `count` private plans with this plan's options on its device): the pieces of a large batch run
        through them side by side (a plan keeps ONE pair of offset tables in HBM, which a piece that is still in
        flight may be reading).  The caller holds the lock from its first enqueue until its streams are
        synchronised: a second large call with the same options - another thread of a joblib-style caller - would
        otherwise rewrite the tables under the first one's kernels.  Such calls are bound by the link, which
        they would share anyway; calls with other options run side by side."""
        key = (_abi.options_key(self.opts), self.device)
        with _LOCK:
            entry = _CLONES.get(key)
            if entry is None:
                while len(_CLONES) >= _MAX_CLONE_SETS:   # (oldest option set first; a set in use stays alive
                    _CLONES.pop(next(iter(_CLONES)))     # through its caller's references until that call ends)
                entry = _CLONES[key] = (threading.Lock(), [])
            lock, clones = entry
            missing = count - len(clones)
        # (plan creation uploads tables: not under the process-wide lock, which get_plan and the shared streams
        # of other threads need; the clones say nothing about the kernel they use - the parent already did)
        fresh = [Plan(self.opts, self.device, quiet=True) for _ in range(max(missing, 0))]
        with _LOCK:
            clones.extend(fresh[:max(count - len(clones), 0)])
            return lock, clones[:count]

    def run_pinned(self, corpus, vtln_warps=None, check_finite=False, wrap=None):
        """`run` over every utterance of a :class:`PinnedCorpus`, uploaded straight from its page-locked block"""
        soff = corpus.soff
        n = soff.shape[0] - 1
        lengths = np.diff(soff)
        frames_of = {int(x): self.num_frames(int(x)) for x in np.unique(lengths)}
        nfr = np.fromiter((frames_of[int(x)] for x in lengths), np.int64, n)
        foff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(nfr, out=foff[1:])
        warp = None
        if vtln_warps is not None:
            warp = np.ascontiguousarray(vtln_warps, dtype=np.float32)
            if warp.shape[0] != n:
                raise ValueError('one vtln_warp per utterance is required')
            if np.all(warp == 1.0):
                warp = None
        wrap = wrap or (lambda res: res)
        if n == 0:
            return wrap([])
        cuts = self._tracker_chunks(soff)
        if cuts is not None:
            res = []
            for a, b in zip(cuts, cuts[1:]):
                res += self.run_pinned(_CorpusSlice(corpus, a, b), None, check_finite, lambda part: part)
            return wrap(res)
        if self.ndims <= 0 or int(soff[-1]) * 2 < _LARGE_BATCH_BYTES:
            return self._run(corpus.views, vtln_warps, check_finite, wrap)
        return self._run_large(corpus.views, soff, foff, nfr, warp, check_finite, wrap, pinned=corpus)

    def _run_large(self, waves, soff, foff, nfr, warp, check_finite, wrap, pinned=None):
        """`run` for batches of tens of megabytes and more (process_all over a corpus).  The batch is cut into
        pieces of whole utterances (~16); each copy thread takes every fourth piece through the whole path on
        its own stream and its own clone of the plan: gather into page-locked memory -> upload -> kernel ->
        download into the page-locked result, all stream ordered, so that the upload of one piece, the kernel
        of another and the download of a third overlap (the link is full duplex) and the call costs little more
        than its upload.  The batch is validated while it is in HBM (one count over the whole output instead of
        two passes over the host copy); the per-utterance views are cut while the pieces are in flight.
        10 000 x 3 s utterances, fbank-40: upload, kernel, download one after the other took 36 ms (round 4)."""
        global _COPY_POOL
        n = len(waves)
        total = int(foff[-1])
        total_samples = int(soff[-1])
        ndims = self.ndims
        d_wave = d_out = None
        if pinned is not None:   # (the source IS page-locked: pieces go up from where they are)
            staged, token = pinned.block, None
        else:
            staged, token = STAGING.array((total_samples,), np.int16)
        try:
            d_wave = DeviceBuffer(max(total_samples * 2, 16), self.device)
            d_out = DeviceBuffer(max(total * ndims * 4, 16), self.device)
            out = result_array((total, ndims), np.float32)
            pieces = max(1, min(_COPY_PIECES * _COPY_THREADS, n, total_samples * 2 // (8 << 20)))
            if self.opts.kind == _abi.KIND_PITCH:
                # (the tracker's cost per utterance falls with the size of a call - 250 / 1 000 / 10 000 utterances:
                # 8 / 3 / 1.2 us - : fewer, larger pieces, still enough of them to run the uploads beside the
                # kernels.  10 000 x 3 s from pageable arrays: 16 pieces 59.6 ms, 8: 49.3, 4: 51.2, 2: 57.4, 1: 78.8)
                pieces = max(1, min(pieces, int(os.environ.get('SNF_PITCH_PIECES', '8'))))
            # (round 5, measured and dropped: a small first piece per thread, so that the link starts after 0.7 ms
            # of gathering instead of 5 - 31-34 against 30-31 ms: the call is bound by the host's memory traffic
            # - 960 MB gathered, read again by the upload, 477 MB written by the download - not by its head)
            cuts = [int(np.searchsorted(soff, total_samples * k // pieces)) for k in range(pieces + 1)]
            cuts[0], cuts[-1] = 0, n
            # (16-byte aligned device blocks: a piece starts on an even utterance boundary only if its sample
            # and frame offsets are multiples of 8 samples / 4 floats - move the cut forward until they are)
            for k in range(1, pieces):
                c = max(cuts[k], cuts[k - 1])
                while c < n and (int(soff[c]) % 8 or (int(foff[c]) * ndims) % 4):
                    c += 1
                cuts[k] = c
            threads = min(_COPY_THREADS, pieces)
            # TWO streams for the whole call: every upload goes on `up`, in the order the pieces are ready, and
            # nothing else does - the link's upload direction never waits for a kernel or a download; a piece's
            # kernel and the download of its rows go on `down` behind an event recorded after its upload.
            # (Measured on 10 000 x 3 s from a page-locked corpus, link alone 16.9 ms for both directions side by
            # side: a stream per thread carrying upload -> kernel -> download of its pieces in turn 24.9 ms - the
            # next upload queues behind the last download -, a stream per piece 28.3 ms.)  Every piece has a plan
            # clone of its own: a clone keeps ONE pair of offset tables, which a piece in flight may be reading.
            clone_lock, clones = self._clones(pieces)
            if _COPY_POOL is None:
                from concurrent.futures import ThreadPoolExecutor
                with _LOCK:
                    if _COPY_POOL is None:
                        _COPY_POOL = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix='snf-copy')
            base_in, base_out = staged.ctypes.data, out.ctypes.data
            L = lib()
            up, down = _shared_stream(self.device, 0), _shared_stream(self.device, 1)
            noise_call = self._next_noise_call()   # (ONE number for the call: every piece draws from its stream)
            events = []
            for _ in range(pieces):
                ev = C.c_void_p()
                check(L.snf_event_create(C.byref(ev)))
                events.append(ev)
            # (a piece's enqueues are not interleaved with another's; two locks: a thread that waits inside a
            # plan call - offset tables of a ragged piece are uploaded with a wait - must not hold up the uploads)
            order_up, order_down = threading.Lock(), threading.Lock()

            def enqueue(k):
                a, b = cuts[k], cuts[k + 1]
                s0, s1, f0, f1 = int(soff[a]), int(soff[b]), int(foff[a]), int(foff[b])
                with order_up:
                    if s1 > s0:
                        check(L.snf_memcpy_h2d_async(C.c_void_p(d_wave.ptr + 2 * s0), C.c_void_p(base_in + 2 * s0),
                                                     2 * (s1 - s0), C.c_void_p(up)))
                    check(L.snf_event_record(events[k], C.c_void_p(up)))
                with order_down:
                    check(L.snf_stream_wait_event(C.c_void_p(down), events[k]))
                    if f1 > f0:
                        clones[k].run_device(d_wave.ptr + 2 * s0, soff[a:b + 1] - s0, foff[a:b + 1] - f0,
                                             d_out.ptr + 4 * ndims * f0,
                                             vtln_warps=None if warp is None else warp[a:b], stream=down,
                                             noise_call=noise_call)
                        check(L.snf_memcpy_d2h_async(C.c_void_p(base_out + 4 * ndims * f0),
                                                     C.c_void_p(d_out.ptr + 4 * ndims * f0),
                                                     4 * ndims * (f1 - f0), C.c_void_p(down)))

            def run(w):
                bind_device(self.device)
                for k in range(w, pieces, threads):
                    a, b = cuts[k], cuts[k + 1]
                    if b <= a:
                        continue
                    s0, s1 = int(soff[a]), int(soff[b])
                    if s1 > s0:
                        np.concatenate(waves[a:b], out=staged[s0:s1])   # (numpy releases the interpreter lock)
                    enqueue(k)

            from concurrent.futures import wait
            futures = []
            clone_lock.acquire()
            try:
                if pinned is None:
                    futures = [_COPY_POOL.submit(run, w) for w in range(threads)]
                else:   # (nothing to gather: this thread enqueues the sixteen pieces, ~0.3 ms)
                    for k in range(pieces):
                        if cuts[k + 1] > cuts[k]:
                            enqueue(k)
                res = []
                for u in range(n):   # (cut while the pieces are in flight)
                    if nfr[u] == 0:
                        res.append(np.zeros((0, 0), dtype=np.float32))   # Kaldi: an empty (0, 0) matrix
                    elif n == 1:
                        res.append(out)
                    else:
                        res.append(out[foff[u]:foff[u + 1]])
                res = wrap(res)
            finally:
                # whatever happened: no thread is still enqueuing, nothing is in flight on these buffers afterwards
                wait(futures)
                L.snf_stream_synchronize(C.c_void_p(up))
                L.snf_stream_synchronize(C.c_void_p(down))
                clone_lock.release()
                for ev in events:
                    L.snf_event_destroy(ev)
            for future in futures:
                future.result()
            if check_finite and total:
                check_finite_device(d_out.ptr, total * ndims, self.device)
        except BaseException:
            for block in (d_wave, d_out):   # (free() waits for the device: whatever was enqueued is done
                if block is not None:       # before the blocks go back to the pool)
                    block.free()
            raise
        finally:
            del staged
            STAGING.release(token)
        d_wave.free(synced=True)   # every copy thread synchronised its stream
        d_out.free(synced=True)
        return res

    # -- Features -> Features --
    def run_post(self, mats, check_finite=False):
        """`mats`: list of float32 [nframes, cols] -> list of float32 [nframes, out_cols]"""
        n = len(mats)
        if n == 0:
            return []
        cols = mats[0].shape[1]
        nfr = np.fromiter((m.shape[0] for m in mats), np.int64, n)
        foff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(nfr, out=foff[1:])
        ocols = self.post_ndims(cols)
        if ocols <= 0:
            raise ValueError('bad column count for this post-processor')
        data, data_token = stage_rows(mats, np.float32)
        out = result_array((int(foff[-1]), ocols), np.float32)
        try:
            check(lib().snf_post_run_batch(
                self.handle, data.ctypes.data_as(C.POINTER(C.c_float)), cols,
                foff.ctypes.data_as(C.POINTER(C.c_int64)), n,
                out.ctypes.data_as(C.POINTER(C.c_float))))
            if check_finite:
                _check_finite(out)
            res = [out] if n == 1 else [out[foff[u]:foff[u + 1]] for u in range(n)]
        finally:
            del data
            STAGING.release(data_token)
        return res

    # -- CMVN (plan kind CMVN): statistics on the GPU, per-speaker sums on the host --
    @staticmethod
    def _pack(mats):
        n = len(mats)
        cols = mats[0].shape[1]
        nfr = np.fromiter((m.shape[0] for m in mats), np.int64, n)
        foff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(nfr, out=foff[1:])
        data, token = stage_rows(mats, np.float32)
        return data, cols, foff, token

    def cmvn_accumulate(self, mats, stats, weights=None, groups=None):
        """stats[groups[u]] += CMVN statistics of mats[u]; `stats` is float64
        [n_groups, 2, cols + 1] (accumulated into, in place)"""
        n = len(mats)
        if n == 0:
            return stats
        assert stats.dtype == np.float64 and stats.flags.c_contiguous
        assert stats.shape[1:] == (2, mats[0].shape[1] + 1)
        w = None
        if weights is not None:
            w = np.ascontiguousarray(np.concatenate(
                [np.asarray(x, dtype=np.float32).ravel() for x in weights]))
            if w.shape[0] != sum(m.shape[0] for m in mats):
                raise ValueError('one weight per frame is required')
        g = None if groups is None else np.ascontiguousarray(groups, np.int32)
        data, cols, foff, token = self._pack(mats)
        try:
            check(lib().snf_cmvn_accumulate(
                self.handle, data.ctypes.data_as(C.POINTER(C.c_float)), cols,
                foff.ctypes.data_as(C.POINTER(C.c_int64)), n,
                w.ctypes.data_as(C.POINTER(C.c_float)) if w is not None else None,
                g.ctypes.data_as(C.POINTER(C.c_int32)) if g is not None else None,
                stats.shape[0], stats.ctypes.data_as(C.POINTER(C.c_double))))
        finally:
            del data
            STAGING.release(token)
        return stats

    def cmvn_apply(self, mats, stats, groups=None, norm_vars=True,
                   reverse=False, check_finite=False):
        """Kaldi ApplyCmvn / ApplyCmvnReverse of mats[u] with stats[groups[u]]"""
        n = len(mats)
        if n == 0:
            return []
        stats = np.ascontiguousarray(stats, dtype=np.float64)
        assert stats.shape[1:] == (2, mats[0].shape[1] + 1)
        g = None if groups is None else np.ascontiguousarray(groups, np.int32)
        data, cols, foff, token = self._pack(mats)
        out = result_array(data.shape, np.float32)
        try:
            check(lib().snf_cmvn_apply(
                self.handle, data.ctypes.data_as(C.POINTER(C.c_float)), cols,
                foff.ctypes.data_as(C.POINTER(C.c_int64)), n,
                stats.ctypes.data_as(C.POINTER(C.c_double)),
                g.ctypes.data_as(C.POINTER(C.c_int32)) if g is not None else None,
                stats.shape[0], int(bool(norm_vars)), int(bool(reverse)),
                out.ctypes.data_as(C.POINTER(C.c_float))))
            if check_finite:
                _check_finite(out)
            res = [out] if n == 1 else [out[foff[u]:foff[u + 1]] for u in range(n)]
        finally:
            del data
            STAGING.release(token)
        return res

    def cmvn_accumulate_device(self, d_in, cols, foff, stats, d_weights=None, groups=None):
        """device-resident `cmvn_accumulate` (d_in / d_weights are device pointers)"""
        n = foff.shape[0] - 1
        g = None if groups is None else np.ascontiguousarray(groups, np.int32)
        check(lib().snf_cmvn_accumulate_device(
            self.handle, C.c_void_p(d_in), cols, foff.ctypes.data_as(C.POINTER(C.c_int64)), n,
            C.c_void_p(d_weights) if d_weights else None,
            g.ctypes.data_as(C.POINTER(C.c_int32)) if g is not None else None,
            stats.shape[0], stats.ctypes.data_as(C.POINTER(C.c_double))))
        return stats

    def cmvn_apply_device(self, d_in, cols, foff, stats, d_out, groups=None, norm_vars=True,
                          reverse=False):
        n = foff.shape[0] - 1
        stats = np.ascontiguousarray(stats, dtype=np.float64)
        g = None if groups is None else np.ascontiguousarray(groups, np.int32)
        check(lib().snf_cmvn_apply_device(
            self.handle, C.c_void_p(d_in), cols, foff.ctypes.data_as(C.POINTER(C.c_int64)), n,
            stats.ctypes.data_as(C.POINTER(C.c_double)),
            g.ctypes.data_as(C.POINTER(C.c_int32)) if g is not None else None,
            stats.shape[0], int(bool(norm_vars)), int(bool(reverse)), C.c_void_p(d_out)))

    # -- device-resident variants (benchmark / pipelines that keep data in HBM) --
    def run_device(self, d_wave, soff, foff, d_out, vtln_warps=None, stream=None, noise_call=None):
        """`noise_call`: the noise stream of this call when the plan dithers (see snf_set_noise_call);
        None = the plan's own call count, a new stream per call"""
        n = soff.shape[0] - 1
        warp = None
        if vtln_warps is not None:
            warp = np.ascontiguousarray(vtln_warps, dtype=np.float32)
        lib().snf_set_noise_call(int(noise_call) if noise_call else self._next_noise_call())  # (thread-local,
        check(lib().snf_plan_run_batch_device(                                                 # used up below)
            self.handle, C.c_void_p(d_wave),
            soff.ctypes.data_as(C.POINTER(C.c_int64)), n,
            warp.ctypes.data_as(C.POINTER(C.c_float)) if warp is not None
            else None,
            C.c_void_p(d_out), foff.ctypes.data_as(C.POINTER(C.c_int64)),
            C.c_void_p(stream) if stream else None))

    def run_post_device(self, d_in, in_cols, foff, d_out, stream=None, noise_call=None):
        n = foff.shape[0] - 1
        if noise_call:
            lib().snf_set_noise_call(int(noise_call))
        check(lib().snf_post_run_batch_device(
            self.handle, C.c_void_p(d_in), in_cols,
            foff.ctypes.data_as(C.POINTER(C.c_int64)), n, C.c_void_p(d_out),
            C.c_void_p(stream) if stream else None))


def concat_columns_device(d_a, cols_a, off_a, d_b, cols_b, off_b, d_out, off_out, device=None):
    """out[u] = [a[u][:rows], b[u][:rows]] on the device (see include/shennong_amd.h)"""
    n = off_a.shape[0] - 1
    p64 = C.POINTER(C.c_int64)
    check(lib().snf_concat_columns_device(
        _DEVICE if device is None else int(device), C.c_void_p(d_a), cols_a,
        off_a.ctypes.data_as(p64), C.c_void_p(d_b), cols_b, off_b.ctypes.data_as(p64), n,
        C.c_void_p(d_out), off_out.ctypes.data_as(p64)))


def check_finite_device(d_ptr, count, device=None):
    """Features.validate's data check on a block that is still in HBM (`count` floats)"""
    bad = C.c_uint64(0)
    check(lib().snf_count_nonfinite_device(
        _DEVICE if device is None else int(device), C.c_void_p(d_ptr), int(count), C.byref(bad)))
    if bad.value:
        raise ValueError('data contains non-finite numbers (nan of infinity)')


def get_plan(opts, device=None):
    """Plan cache keyed by (options bytes, device): options are copied by value at every
    ``process`` call like the reference does (processor/base.py:421-425)"""
    key = (_abi.options_key(opts), _DEVICE if device is None else int(device))
    with _LOCK:
        plan = _PLANS.get(key)
    if plan is None:
        plan = Plan(opts, device)
        with _LOCK:
            if len(_PLANS) > 64:
                _PLANS.clear()
            _PLANS.setdefault(key, plan)
            plan = _PLANS[key]
    return plan


def clear_plans():
    with _LOCK:
        _PLANS.clear()
        _CLONES.clear()


# ---- raw device memory (for hosts that keep batches resident in HBM) ----------
class _Staging:
    """Page-locked host staging buffers for the host-pointer entry points (snf_host_malloc), pooled:
    a batch is assembled directly in one (no page faults after the first use, full link rate), the
    result comes back into another and is cut into per-utterance arrays from there.  Below
    `_MIN_BYTES` plain numpy memory is used (a single utterance is not worth a pinned buffer); a
    failed pinned allocation falls back to plain memory as well - this only concerns the KIND of host
    memory, the arithmetic always runs on the GPU."""
    _MIN_BYTES = 1 << 20
    _MAX_POOLED = 4
    _MAX_POOLED_BYTES = 8 << 30

    def __init__(self):
        self._free = []  # (capacity, pointer)
        # re-entrant: a garbage-collection pass triggered by an allocation inside the lock may finalise a
        # _ResultBlock on this very thread, whose __del__ comes back here (release)
        self._lock = threading.RLock()

    def array(self, shape, dtype, allocate=True):
        """-> (numpy array of `shape` / `dtype`, token to pass to :func:`release`); with `allocate` false a
        pooled buffer is used if one fits, none is created (page-locking fresh memory costs ~0.25 ms/MB)"""
        dtype = np.dtype(dtype)
        count = int(np.prod(shape))
        nbytes = count * dtype.itemsize
        if nbytes < self._MIN_BYTES:
            return np.empty(shape, dtype=dtype), None
        token = None
        with self._lock:
            fits = [b for b in self._free if b[0] >= nbytes]
            if fits:
                token = min(fits)
                self._free.remove(token)
        if token is None and not allocate:
            return np.empty(shape, dtype=dtype), None
        if token is None:
            capacity = (nbytes + nbytes // 8 + (1 << 21) - 1) & ~((1 << 21) - 1)
            ptr = C.c_void_p()
            if lib().snf_host_malloc(C.byref(ptr), capacity) != 0 or not ptr.value:
                return np.empty(shape, dtype=dtype), None
            token = (capacity, ptr.value)
        raw = (C.c_char * nbytes).from_address(token[1])
        return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape), token

    def release(self, token):
        if token is None:
            return
        drop = None
        with self._lock:
            self._free.append(token)
            if len(self._free) > self._MAX_POOLED or \
                    sum(b[0] for b in self._free) > self._MAX_POOLED_BYTES:
                drop = max(self._free) if sum(b[0] for b in self._free) > self._MAX_POOLED_BYTES \
                    else min(self._free)
                self._free.remove(drop)
        if drop is not None:
            lib().snf_host_free(C.c_void_p(drop[1]))


STAGING = _Staging()


class _ResultBlock:
    """Owner of one pooled page-locked buffer that holds the results of a batch.  The per-utterance
    matrices are numpy views of it (``np.asarray(block)``: numpy keeps the block as their base); the
    buffer goes back to the pool when the last of them is gone."""
    _LIMIT = 8 << 30   # page-locked result memory handed out and not yet returned, at most
    _FRESH = 4 << 30   # ... below which a result may page-lock a new buffer (see result_array; 1 GiB until the
                       # streamed pipeline's four-hour batches: 0.7-1.5 GB of results per batch; 2 GiB until
                       # that pipeline kept the result block of batch k alive beside the one of batch k + 1)
    _held = 0
    _lock = threading.RLock()   # (re-entrant for the same reason as _Staging._lock)

    def __init__(self, shape, dtype, token, address):
        self._token = token
        self._nbytes = token[0]
        self.__array_interface__ = {
            'shape': tuple(int(x) for x in shape), 'typestr': np.dtype(dtype).str,
            'data': (address, False), 'version': 3}

    def __del__(self):
        token, self._token = self._token, None
        if token is not None:
            with _ResultBlock._lock:
                _ResultBlock._held -= self._nbytes
            try:
                STAGING.release(token)
            except Exception:  # pragma: nocover (interpreter shutdown)
                pass


class _PinnedOwner:
    """Owner of one page-locked allocation of its own (not pooled): numpy views made through
    ``np.asarray(owner)`` keep it alive, the memory is released with the last of them"""
    def __init__(self, shape, dtype):
        dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(shape)) * dtype.itemsize
        ptr = C.c_void_p()
        check(lib().snf_host_malloc(C.byref(ptr), max(self.nbytes, 16)))
        self.address = ptr.value
        self.__array_interface__ = {'shape': tuple(int(x) for x in shape), 'typestr': dtype.str,
                                    'data': (self.address, False), 'version': 3}

    def __del__(self):
        address, self.address = getattr(self, 'address', None), None
        if address and _LIB is not None:
            try:
                _LIB.snf_host_free(C.c_void_p(address))
            except Exception:  # pragma: nocover (interpreter shutdown)
                pass


class _CorpusSlice:
    """Utterances a .. b of a page-locked corpus, as a corpus (Plan.run_pinned of a stretch of it)"""
    def __init__(self, corpus, a, b):
        self.sample_rate = corpus.sample_rate
        s0, s1 = int(corpus.soff[a]), int(corpus.soff[b])
        self.soff = corpus.soff[a:b + 1] - s0
        self.block = corpus.block[s0:s1]
        self.views = corpus.views[a:b]


class PinnedCorpus:
    """The int16 audio of a set of utterances in ONE page-locked block (``Utterances.pin()``): what the
    processors force every signal to before they run (reference processor/base.py:428), loaded once.  A batch
    made of these utterances in this order is uploaded straight from the block - no gather into a staging buffer
    (which is what bounds `process_all` on pageable arrays: the host's memory traffic, not the link) and no
    per-utterance checks: they were made when the block was built."""
    def __init__(self, waves, sample_rate, lengths=None, fill=None):
        """`waves`: one int16 array per utterance - or None with `lengths` and ``fill(block, soff)``, which writes
        the samples itself (the native WAV reader: files go straight into the block)"""
        n = len(waves) if waves is not None else len(lengths)
        self.sample_rate = int(sample_rate)
        self.soff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum([w.shape[0] for w in waves] if waves is not None else lengths, out=self.soff[1:])
        total = int(self.soff[-1])
        self.owner = _PinnedOwner((max(total, 8),), np.int16)
        self.block = np.asarray(self.owner)
        if waves is not None:
            for k, w in enumerate(waves):
                self.block[self.soff[k]:self.soff[k + 1]] = w
        else:
            fill(self.block, self.soff)
        self.views = [self.block[self.soff[k]:self.soff[k + 1]] for k in range(n)]


class StagedCorpus:
    """A PinnedCorpus for ONE call: the int16 samples of a batch in a POOLED page-locked staging buffer (no
    0.25 ms / MB of page-locking per call), filled by the caller (`block`, `soff`), handed to ``Plan.run_pinned``
    and released.  Falls back to plain memory when no pooled buffer can be had (the copy is then synchronous)."""
    def __init__(self, lengths, sample_rate):
        n = len(lengths)
        self.sample_rate = int(sample_rate)
        self.soff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lengths, out=self.soff[1:])
        total = int(self.soff[-1])
        self._array, self._token = STAGING.array((max(total, 8),), np.int16)
        self.block = self._array
        self.views = [self.block[self.soff[k]:self.soff[k + 1]] for k in range(n)]

    def release(self):
        self.views = self.block = self._array = None
        token, self._token = self._token, None
        STAGING.release(token)


RESULT_KINDS = {'page_locked': 0, 'plain': 0}   # batch results by kind of host memory (diagnostics: a corpus run
                                                # whose results land in plain memory downloads at half the rate)


def read_wav_pcm16(paths, first_samples, n_samples, dst, dst_offsets, threads=None):
    """Samples [first, first + n) of every file of `paths` (16-bit mono PCM WAV) into the int16 array `dst` at
    `dst_offsets`, read natively on `threads` threads (snf_wav_read_pcm16; default: the cores this process may
    use, at most 16).  Returns the per-file status (0 read, 1 not 16-bit mono PCM: the caller's general reader
    takes it, 2 I/O error, 3 shorter than its header said)."""
    n = len(paths)
    status = np.zeros(n, dtype=np.int32)
    if n == 0:
        return status
    if threads is None:
        threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))
    names = (C.c_char_p * n)(*[os.fsencode(p) for p in paths])
    first = np.ascontiguousarray(first_samples, dtype=np.int64)
    count = np.ascontiguousarray(n_samples, dtype=np.int64)
    where = np.ascontiguousarray(dst_offsets, dtype=np.int64)
    assert dst.dtype == np.int16 and dst.flags.c_contiguous
    assert int((where + count).max()) <= dst.size
    p64 = C.POINTER(C.c_int64)
    check(lib().snf_wav_read_pcm16(names, n, first.ctypes.data_as(p64), count.ctypes.data_as(p64),
                                   dst.ctypes.data_as(C.POINTER(C.c_int16)), where.ctypes.data_as(p64),
                                   int(threads), status.ctypes.data_as(C.POINTER(C.c_int32))))
    return status


def result_array(shape, dtype):
    """Uninitialised host array for the one device -> host copy of a batch: pooled page-locked memory
    (no page faults, no munmap per batch, twice the link rate of pageable memory).  Page-locking fresh memory
    costs ~0.25 ms/MB - more than the page faults it saves unless the buffer is used again -, so a NEW buffer
    is only made while the result blocks that are alive stay under `_ResultBlock._FRESH` bytes: the steady
    state of a corpus run that writes every batch and drops it reuses one or two blocks for ever, a caller
    that keeps all its results gets plain numpy memory once that much is pinned (pooled blocks that fit are
    still used, up to `_ResultBlock._LIMIT` alive)."""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    with _ResultBlock._lock:
        held = _ResultBlock._held
    if held + nbytes > _ResultBlock._LIMIT or nbytes < _Staging._MIN_BYTES:
        RESULT_KINDS['plain'] += nbytes >= _Staging._MIN_BYTES
        return np.empty(shape, dtype=dtype)
    array, token = STAGING.array(shape, dtype, allocate=held + nbytes <= _ResultBlock._FRESH)
    RESULT_KINDS['plain' if token is None else 'page_locked'] += 1
    if token is None:
        return array
    block = _ResultBlock(shape, dtype, token, token[1])
    del array
    with _ResultBlock._lock:
        _ResultBlock._held += token[0]
    return np.asarray(block)


def stage_rows(mats, dtype):
    """Concatenation of `mats` along axis 0 in a staging buffer -> (array, token)"""
    first = np.asarray(mats[0])
    shape = (int(sum(m.shape[0] for m in mats)),) + tuple(first.shape[1:])
    staged, token = STAGING.array(shape, dtype)
    if len(mats) == 1:
        staged[...] = first
    elif shape[0]:
        np.concatenate([np.asarray(m, dtype=dtype) for m in mats], axis=0, out=staged)
    return staged, token


_COPY_POOL = None
_CLONES = {}   # (options, device) -> private plans of Plan._clones
_MAX_CLONE_SETS = 8   # option sets whose clones (16 plans' tables each) are kept
_MAX_TRACKER_HOURS = float(os.environ.get('SNF_MAX_TRACKER_HOURS', '16'))   # audio per pitch call (~37 GB of scratch)
_LARGE_BATCH_BYTES = 32 << 20   # Plan.run: batches from this many bytes of audio take the overlapped path
_COPY_PIECES = int(os.environ.get('SNF_COPY_PIECES', '4'))    # pieces of a large batch per copy thread
_COPY_THREADS = int(os.environ.get('SNF_COPY_THREADS', '4'))  # (8 / 16 threads measured slower: 6.4 / 5.1 against 4.1 ms per 96 MB)


def upload_rows(mats, dtype, device=None):
    """`mats` concatenated along axis 0 in a new DeviceBuffer.  Large batches are cut in 4 pieces of whole
    rows per copy thread: a thread gathers a piece into its part of one page-locked staging buffer, starts
    the piece's copy on its own stream and gathers the next one meanwhile (numpy's copy loops and the HIP
    calls release the interpreter lock) - a single thread gathers at ~20 GB/s, less than half of what the
    link takes, four reach the host's memory bandwidth (46 GB/s)."""
    global _COPY_POOL
    dtype = np.dtype(dtype)
    mats = [np.asarray(m, dtype=dtype) for m in mats]
    rows = np.zeros(len(mats) + 1, dtype=np.int64)
    np.cumsum([m.shape[0] for m in mats], out=rows[1:])
    row_bytes = dtype.itemsize * int(np.prod(mats[0].shape[1:])) if mats else dtype.itemsize
    total = int(rows[-1]) * row_bytes
    buf = DeviceBuffer(max(total, 16), device)
    if total == 0:
        return buf
    shape = (int(rows[-1]),) + tuple(mats[0].shape[1:])
    staged, token = STAGING.array(shape, dtype)
    try:
        if total < (8 << 20) or len(mats) < 2 * _COPY_THREADS:
            if len(mats) == 1:
                staged[...] = mats[0]
            else:
                np.concatenate(mats, axis=0, out=staged)
            buf.upload(staged)
            return buf
        if _COPY_POOL is None:
            from concurrent.futures import ThreadPoolExecutor
            with _LOCK:
                if _COPY_POOL is None:
                    _COPY_POOL = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix='snf-copy')
        # 4 pieces per thread: a thread gathers a piece, starts its copy on the thread's own stream and gathers
        # the next one meanwhile (measured on 115 MB: 3.7 ms against 4.7 for one gather + one blocking copy per
        # thread; the gather alone takes 2.5 ms with 4 threads - host memory bandwidth -, the copy alone 2.3)
        pieces = 4 * _COPY_THREADS
        cuts = [int(np.searchsorted(rows, rows[-1] * k // pieces)) for k in range(pieces + 1)]
        cuts[0], cuts[-1] = 0, len(mats)
        base = staged.ctypes.data

        def run(w):
            bind_device(buf.device)
            stream = _copy_stream(buf.device)
            try:
                for k in range(w, pieces, _COPY_THREADS):
                    a, b = cuts[k], cuts[k + 1]
                    if b <= a:
                        continue
                    np.concatenate(mats[a:b], axis=0, out=staged[rows[a]:rows[b]])
                    check(lib().snf_memcpy_h2d_async(C.c_void_p(buf.ptr + int(rows[a]) * row_bytes),
                                                     C.c_void_p(base + int(rows[a]) * row_bytes),
                                                     int(rows[b] - rows[a]) * row_bytes, C.c_void_p(stream)))
            finally:
                # whatever happened, nothing of this thread is in flight from `staged` into `buf` afterwards
                lib().snf_stream_synchronize(C.c_void_p(stream))

        # every copy thread has finished (and synchronised its stream) before the staging block and the
        # device buffer can go back to their pools: a failure in one thread must not free them under the others
        from concurrent.futures import wait
        futures = [_COPY_POOL.submit(run, w) for w in range(_COPY_THREADS)]
        wait(futures)
        try:
            for future in futures:
                future.result()
        except BaseException:
            buf.free()
            raise
        return buf
    finally:
        del staged
        STAGING.release(token)


def _check_finite(out):
    """Features.validate's data check for a whole batch at once (NaN propagates through min / max,
    an infinity is the min or the max)"""
    if out.size and not (np.isfinite(out.min()) and np.isfinite(out.max())):
        raise ValueError('data contains non-finite numbers (nan of infinity)')


_STREAMS = threading.local()


def _copy_stream(device, index=0):
    """Non-blocking HIP streams per (thread, device) for asynchronous copies; `index`: one of several"""
    streams = getattr(_STREAMS, 'by_device', None)
    if streams is None:
        streams = _STREAMS.by_device = {}
    key = (device, index)
    if key not in streams:
        handle = C.c_void_p()
        check(lib().snf_stream_create(C.byref(handle)))
        streams[key] = handle.value
    return streams[key]


_SHARED_STREAMS = {}


def _shared_stream(device, index):
    """Process-wide non-blocking streams of a device (HIP streams may be fed from any thread): 0 = uploads of
    large batches, 1 = their kernels and downloads (Plan._run_large).  Two callers at once share them: their
    pieces queue behind each other, which is what the link does with them anyway."""
    key = (device, index)
    with _LOCK:
        if key not in _SHARED_STREAMS:
            handle = C.c_void_p()
            check(lib().snf_stream_create(C.byref(handle)))
            _SHARED_STREAMS[key] = handle.value
        return _SHARED_STREAMS[key]


_SIDE_POOL = None


def side_pool():
    """Two worker threads for launches that do not depend on what the calling thread does next (the pitch
    tracker beside the features -> CMVN -> delta chain of the pipeline): every entry point releases the
    interpreter lock, every plan has its own stream"""
    global _SIDE_POOL
    if _SIDE_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        with _LOCK:
            if _SIDE_POOL is None:
                _SIDE_POOL = ThreadPoolExecutor(max_workers=2, thread_name_prefix='snf-side')
    return _SIDE_POOL


class _DevicePool:
    """Freed device buffers kept for the next batch (hipMalloc / hipFree synchronise the device and cost
    0.1-0.4 ms each and more for large blocks; a pipeline call makes twenty of them).  Bounded by
    SNF_DEVICE_POOL_BYTES (default: a sixth of the device, at most 48 GiB of the 288 GB); a buffer is reused for a request of at least half its
    size.  When a block does not fit under the bound, the blocks that were parked LONGEST AGO are released to make
    room: the pool follows the workload (round 6: after the benchmark's other legs had filled it with their block
    sizes, every buffer of the streamed pipeline and of `process_all` went to hipMalloc and back to hipFree -
    25 against 20 ms per `process_all` call)."""
    def __init__(self):
        # (no bound named: a sixth of the device's memory, between 4 and 48 GiB, read when the first block comes
        # back - 48 GiB of the 288 GB of an MI355X: the 14.4 GB of audio that a 125 h corpus keeps resident
        # between its two passes come back to the pool block by block and used to push each other out through
        # hipFree, 0.13 s per run)
        limit = os.environ.get('SNF_DEVICE_POOL_BYTES')
        self.limit = int(limit) if limit is not None else None
        # tests: a reused buffer is filled with NaN bit patterns before it is handed out, so that a kernel
        # that relies on what a fresh allocation happens to contain shows up
        self.poison = bool(int(os.environ.get('SNF_DEVICE_POOL_POISON', '0')))
        self._free = {}  # device -> list of (capacity, pointer, stamp)
        self._bytes = 0
        self._stamp = 0
        self._lock = threading.Lock()

    def take(self, device, nbytes):
        with self._lock:
            blocks = self._free.get(device)
            if blocks:
                fits = [b for b in blocks if nbytes <= b[0] <= 2 * nbytes + 4096]
                if fits:
                    block = min(fits)
                    blocks.remove(block)
                    self._bytes -= block[0]
                    return block[:2]
        return None

    def give(self, device, block):
        """parks `block` (capacity, pointer); -> False when it alone exceeds the bound (the caller releases it)"""
        evicted = []
        if self.limit is None:
            try:
                self.limit = int(min(max(mem_info(device)[1] // 6, 4 << 30), 48 << 30))
            except Exception:  # pragma: nocover
                self.limit = 16 << 30
        with self._lock:
            if block[0] > self.limit:
                return False
            while self._bytes + block[0] > self.limit:
                oldest = min(((b[2], d, b) for d, bs in self._free.items() for b in bs), default=None)
                if oldest is None:
                    break
                _, d, b = oldest
                self._free[d].remove(b)
                self._bytes -= b[0]
                evicted.append((d, b))
            self._stamp += 1
            self._free.setdefault(device, []).append((block[0], block[1], self._stamp))
            self._bytes += block[0]
        for d, b in evicted:   # (the caller has just waited for the device or says it is idle: see DeviceBuffer.free)
            bind_device(d)
            lib().snf_free(C.c_void_p(b[1]))
        if evicted:
            bind_device(device)
        return True

    def clear(self):
        with self._lock:
            blocks = [(d, b) for d, bs in self._free.items() for b in bs]
            self._free.clear()
            self._bytes = 0
        # (binds the thread to every device that has parked blocks: callers re-bind afterwards -
        # DeviceBuffer.__init__ below; the library's out-of-memory hook, which also runs this in the middle of
        # a plan call, saves and restores the thread's device around it: csrc/capi.hip malloc_with_hook)
        for device, block in blocks:
            bind_device(device)
            lib().snf_free(C.c_void_p(block[1]))


DEVICE_POOL = _DevicePool()


class DeviceBuffer:
    """HBM allocation on the selected GPU (`set_device` / SHENNONG_AMD_DEVICE)"""
    def __init__(self, nbytes, device=None):
        self.device = _DEVICE if device is None else int(device)
        bind_device(self.device)
        want = int(max(nbytes, 16))
        block = DEVICE_POOL.take(self.device, want)
        if block is None:
            ptr = C.c_void_p()
            if lib().snf_malloc(C.byref(ptr), want) != 0:
                DEVICE_POOL.clear()  # (out of memory with buffers parked in the pool: give them back first)
                bind_device(self.device)   # (clear() walks over the devices that had parked blocks)
                check(lib().snf_malloc(C.byref(ptr), want))
            block = (want, ptr.value)
        elif DEVICE_POOL.poison:
            check(lib().snf_memset(C.c_void_p(block[1]), 0xFF, block[0]))  # (complete when it returns)
        self._capacity, self.ptr = block
        self.nbytes = int(nbytes)

    def upload(self, array):
        array = np.ascontiguousarray(array)
        bind_device(self.device)
        check(lib().snf_memcpy_h2d(
            C.c_void_p(self.ptr), array.ctypes.data_as(C.c_void_p),
            array.nbytes))

    def download(self, array):
        bind_device(self.device)
        check(lib().snf_memcpy_d2h(
            array.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
            array.nbytes))
        return array

    def upload_async(self, array):
        """Starts the copy of `array` (contiguous; page-locked for the copy to be asynchronous) into this block
        on the calling thread's copy stream and returns a callable that waits for it (from any thread)"""
        bind_device(self.device)
        stream = _copy_stream(self.device)
        L = lib()
        check(L.snf_memcpy_h2d_async(C.c_void_p(self.ptr), array.ctypes.data_as(C.c_void_p),
                                     array.nbytes, C.c_void_p(stream)))
        done = C.c_void_p()
        check(L.snf_event_create(C.byref(done)))
        check(L.snf_event_record(done, C.c_void_p(stream)))

        def wait():   # (THIS copy, not whatever was enqueued on the stream behind it; a second call is a no-op)
            if done.value:
                try:
                    check(L.snf_event_synchronize(done))
                finally:
                    L.snf_event_destroy(done)
                    done.value = None
        return wait

    def download_async(self, array):
        """Starts the copy into `array` on this thread's download stream (its own: an upload that is on its way
        does not queue behind it, the link's two directions run side by side) and returns a callable that waits
        for THIS copy, from any thread; the caller does its host-side bookkeeping - or the next batch's
        launches - in between (the copy only overlaps when `array` is page-locked, :func:`result_array`; into
        plain memory it is done when this returns)"""
        bind_device(self.device)
        stream = _copy_stream(self.device, 1)
        L = lib()
        check(L.snf_memcpy_d2h_async(array.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                     array.nbytes, C.c_void_p(stream)))
        done = C.c_void_p()
        check(L.snf_event_create(C.byref(done)))
        check(L.snf_event_record(done, C.c_void_p(stream)))

        def wait():   # (a second call is a no-op)
            if done.value:
                try:
                    check(L.snf_event_synchronize(done))
                finally:
                    L.snf_event_destroy(done)
                    done.value = None
        return wait

    def free(self, synced=False):
        """Gives the block back (to the pool, or to the driver when the pool is full).  Like hipFree, which
        it used to be, this waits for the device first: work that was enqueued on any stream and still reads
        or writes the block (``run_device(stream=...)``, ``download_async`` before its wait) is complete
        before another owner - possibly another thread - can get the same memory from the pool.  Callers
        that have just synchronised pass ``synced=True`` and skip the wait (~10 us on an idle device)."""
        if self.ptr:
            ptr, self.ptr = self.ptr, None
            bind_device(self.device)
            if not synced:
                lib().snf_device_synchronize()
            if not DEVICE_POOL.give(self.device, (self._capacity, ptr)):
                lib().snf_free(C.c_void_p(ptr))

    def __del__(self):
        try:
            self.free()
        except Exception:  # pragma: nocover
            pass
