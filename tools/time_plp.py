"""Dev helper (GPU box): PLP-13 on 10 000 x 3 s utterances, HIP-event time of the mel kernel and of the tail.
   python tools/time_plp.py"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shennong_amd import _backend, synth
from shennong_amd.processor import PlpProcessor
n, ns = 10000, 48000
w = np.tile(synth.utterances(0, 64, ns), (n // 64 + 1, 1))[:n]
d_w = _backend.DeviceBuffer(w.nbytes); d_w.upload(w)
plan = _backend.Plan(PlpProcessor(dither=0)._build_options())
fpu = plan.num_frames(ns); soff = np.arange(n + 1, dtype=np.int64) * ns; foff = np.arange(n + 1, dtype=np.int64) * fpu
d_o = _backend.DeviceBuffer(fpu * n * plan.ndims * 4)
for _ in range(40): plan.run_device(d_w.ptr, soff, foff, d_o.ptr)
ks, k1, k2 = [], [], []
for _ in range(30):
    plan.run_device(d_w.ptr, soff, foff, d_o.ptr); ks.append(plan.last_kernel_ms(0)); k1.append(plan.last_kernel_ms(1)); k2.append(plan.last_kernel_ms(2))
print('plp13 total %.4f  %s %.4f  %s %.4f' % (np.median(ks), plan.kernel_name(1), np.median(k1), plan.kernel_name(2), np.median(k2)))
