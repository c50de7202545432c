"""Developer measurement for BASELINE config 4 (PlpProcessor 13 + KaldiPitchProcessor (+ post) on
100 000 utterances of 1-6 s sharded over 8 GPUs): ONE GPU's shard of 12 500 ragged utterances,
device resident, kernel times per stage.

    python tools/config4_shard.py [n_utterances]"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _gen(args):
    from shennong_amd import synth
    first, count = args
    rng = np.random.default_rng(20260927 + first)
    return [synth.utterances(first + i, 1, int(rng.integers(16000, 96001)))[0] for i in range(count)]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
    workers = min(16, os.cpu_count() or 1)
    chunk = (n + workers - 1) // workers
    t0 = time.perf_counter()
    with ProcessPoolExecutor(workers) as pool:
        waves = [w for part in pool.map(_gen, [(i, min(chunk, n - i)) for i in range(0, n, chunk)])
                 for w in part]
    from shennong_amd import _backend
    from shennong_amd.processor import KaldiPitchPostProcessor, KaldiPitchProcessor, PlpProcessor
    lengths = np.array([w.shape[0] for w in waves], dtype=np.int64)
    seconds = lengths.sum() / 16000.0
    print(f'{n} utterances, {seconds / 3600:.2f} h of audio (lengths {lengths.min() / 16000:.1f}-'
          f'{lengths.max() / 16000:.1f} s), generated in {time.perf_counter() - t0:.1f} s')
    soff = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(lengths, out=soff[1:])
    wave, token = _backend.stage_rows(waves, np.int16)
    d_wave = _backend.DeviceBuffer(wave.nbytes)
    d_wave.upload(wave)
    del wave
    _backend.STAGING.release(token)
    total_ms = 0.0
    for name, proc in (('plp13', PlpProcessor(dither=0)), ('kaldi pitch', KaldiPitchProcessor())):
        plan = _backend.get_plan(proc._build_options())
        nfr = np.array([plan.num_frames(int(x)) for x in lengths], dtype=np.int64)
        foff = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(nfr, out=foff[1:])
        d_out = _backend.DeviceBuffer(int(foff[-1]) * plan.ndims * 4)
        ms = []
        for _ in range(3):
            plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
            ms.append(plan.last_kernel_ms(0))
        total_ms += min(ms)
        print(f'{name:12s} {int(foff[-1])} frames, kernels {min(ms):8.2f} ms = {foff[-1] / min(ms) * 1e3:.3e} frames/s')
        if name == 'kaldi pitch':
            post = _backend.get_plan(KaldiPitchPostProcessor()._build_options())
            d_post = _backend.DeviceBuffer(int(foff[-1]) * 3 * 4)
            ms = []
            for _ in range(3):
                post.run_post_device(d_out.ptr, 2, foff, d_post.ptr)
                ms.append(post.last_kernel_ms(0))
            total_ms += min(ms)
            print(f'{"pitch post":12s} {int(foff[-1])} frames, kernels {min(ms):8.2f} ms')
            d_post.free()
        d_out.free()
    print(f'shard total: {total_ms:.1f} ms of kernels for {seconds / 3600:.2f} h of audio '
          f'= {seconds / total_ms * 1e3 / 3600:.1f} h of audio per second per GPU '
          f'({n / total_ms * 1e3:.0f} utterances/s)')


if __name__ == '__main__':
    main()
