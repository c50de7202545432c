"""Dev helper: run the plans of the other sample rates (44.1 / 48 / 32 / 22.05 / 8 kHz: fbank2048_kernel,
fbank1024x2_kernel, fbank256x2_kernel) and the delta plan a few times (for rocprofv3 --kernel-trace / --pmc); prints the
HIP-event kernel times next to the kernel names"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shennong_amd import _backend, synth
from shennong_amd.processor import FilterbankProcessor, MfccProcessor
from shennong_amd.postprocessor import DeltaPostProcessor

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only = int(sys.argv[2]) if len(sys.argv) > 2 else None   # (one sample rate only)
# 596 000 frames at the long rates (2000 x 3 s), 1 192 000 at 8 kHz (4000 x 3 s)
for sr, n_utts in ((44100, 2000), (48000, 2000), (32000, 2000), (22050, 2000), (8000, 4000)):
    if only is not None and sr != only:
        continue
    ns = 3 * sr
    base = synth.utterances(0, 20, ns, sr)
    waves = np.ascontiguousarray(np.tile(base, (n_utts // 20, 1)))
    d_wave = _backend.DeviceBuffer(waves.nbytes)
    d_wave.upload(waves)
    for cls, opts in ((FilterbankProcessor, dict(num_bins=40)), (MfccProcessor, dict())):
        proc = cls(sample_rate=sr, dither=0, **opts)
        plan = _backend.get_plan(proc._build_options())
        fpu = plan.num_frames(ns)
        soff = np.arange(n_utts + 1, dtype=np.int64) * ns
        foff = np.arange(n_utts + 1, dtype=np.int64) * fpu
        d_out = _backend.DeviceBuffer(fpu * n_utts * plan.ndims * 4)
        for _ in range(5):
            plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
        ks = []
        for _ in range(reps):
            plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
            ks.append(plan.last_kernel_ms(0))
        print('%-20s %5d Hz %-20s %8d frames  kernel_ms median %.4f min %.4f' % (
            cls.__name__, sr, plan.kernel_name(1), fpu * n_utts, np.median(ks), np.min(ks)), flush=True)
        d_out.free()
    d_wave.free()
if only is not None:
    sys.exit(0)
# delta 13 -> 39 on 2.98 M frames
n_utts, fpu = 10000, 298
x = np.random.default_rng(0).standard_normal((n_utts * fpu, 13)).astype(np.float32)
d_in = _backend.DeviceBuffer(x.nbytes)
d_in.upload(x)
d_out = _backend.DeviceBuffer(x.nbytes * 3)
plan = _backend.get_plan(DeltaPostProcessor()._build_options())
foff = np.arange(n_utts + 1, dtype=np.int64) * fpu
ks = []
for i in range(5 + reps):
    plan.run_post_device(d_in.ptr, 13, foff, d_out.ptr)
    if i >= 5:
        ks.append(plan.last_kernel_ms(0))
print('%-20s %-29s %8d frames  kernel_ms median %.4f min %.4f' % (
    'DeltaPostProcessor', 'delta_flat_o2w2_kernel', n_utts * fpu, np.median(ks), np.min(ks)), flush=True)
