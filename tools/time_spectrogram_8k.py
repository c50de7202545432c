"""Dev helper: the 8 kHz spectrogram (129 bins) on fbank256x2_kernel against the generic kernel
(SNF_DISABLE_DUAL256=1), 4 000 x 3 s utterances"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shennong_amd import _backend, synth
from shennong_amd.processor import SpectrogramProcessor
sr, n_utts = 8000, 4000
ns = 3 * sr
base = synth.utterances(0, 20, ns, sr)
waves = np.ascontiguousarray(np.tile(base, (n_utts // 20, 1)))
d_wave = _backend.DeviceBuffer(waves.nbytes)
d_wave.upload(waves)
proc = SpectrogramProcessor(sample_rate=sr, dither=0)
plan = _backend.get_plan(proc._build_options())
fpu = plan.num_frames(ns)
soff = np.arange(n_utts + 1, dtype=np.int64) * ns
foff = np.arange(n_utts + 1, dtype=np.int64) * fpu
d_out = _backend.DeviceBuffer(fpu * n_utts * plan.ndims * 4)
for _ in range(5):
    plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
ks = []
for _ in range(10):
    plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
    ks.append(plan.last_kernel_ms(0))
print('spectrogram-129 8 kHz %s: %d frames, kernel_ms median %.4f min %.4f (%.2f TB/s of stores)' % (
    plan.kernel_name(1), fpu * n_utts, np.median(ks), np.min(ks), fpu * n_utts * 129 * 4 / np.median(ks) / 1e9))
