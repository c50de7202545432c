"""Dev helper: run the pitch / PLP / MFCC+delta plans a few times (for rocprofv3 --kernel-trace)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shennong_amd import _backend, synth
from shennong_amd.processor import KaldiPitchProcessor, PlpProcessor
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
waves = synth.utterances(0, n, 48000)
procs = (KaldiPitchProcessor(), PlpProcessor(dither=0), PlpProcessor(dither=0, rasta=True))
if len(sys.argv) > 2 and sys.argv[2] == 'pitch-only':   # (counter passes: the pitch kernels only)
    procs = procs[:1]
for proc in procs:
    plan = _backend.get_plan(proc._build_options())
    nf = plan.num_frames(48000)
    soff = np.arange(n + 1, dtype=np.int64) * 48000
    foff = np.arange(n + 1, dtype=np.int64) * nf
    d_wave = _backend.DeviceBuffer(waves.nbytes); d_wave.upload(waves)
    d_out = _backend.DeviceBuffer(nf * n * plan.ndims * 4)
    for _ in range(3):
        plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
    print(proc.name, [ (plan.kernel_name(i), round(plan.last_kernel_ms(i), 3)) for i in range(1, 4) if plan.kernel_name(i)], 'total', round(plan.last_kernel_ms(0), 3))
