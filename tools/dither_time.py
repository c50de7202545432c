"""Dev helper (GPU box): kernel time of fbank-40 and MFCC-13 with dither 0 and with the default dither 1.0, same
process, alternated (the ratio DESIGN.md 4.1 quotes).   python tools/dither_time.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shennong_amd import _backend, synth
from shennong_amd.processor import FilterbankProcessor, MfccProcessor
n, ns = 10000, 48000
w = np.tile(synth.utterances(0, 64, ns), (n // 64 + 1, 1))[:n]
d_w = _backend.DeviceBuffer(w.nbytes); d_w.upload(w)
for name, mk in (('fbank40', lambda d: FilterbankProcessor(num_bins=40, dither=d)), ('mfcc13', lambda d: MfccProcessor(dither=d))):
    res = {}
    for rep in range(2):
        for d in (0.0, 1.0):
            plan = _backend.Plan(mk(d)._build_options())
            fpu = plan.num_frames(ns); soff = np.arange(n + 1, dtype=np.int64) * ns; foff = np.arange(n + 1, dtype=np.int64) * fpu
            d_o = _backend.DeviceBuffer(fpu * n * plan.ndims * 4)
            for _ in range(40): plan.run_device(d_w.ptr, soff, foff, d_o.ptr)
            ks = []
            for _ in range(30):
                plan.run_device(d_w.ptr, soff, foff, d_o.ptr); ks.append(plan.last_kernel_ms(0))
            res.setdefault(d, []).append(float(np.median(ks)))
            d_o.free()
    print(name, 'dither 0: %s  dither 1: %s  ratio %.3f' % (res[0.0], res[1.0], min(res[1.0]) / min(res[0.0])))
