"""Dev helper: HIP-event times of the pitch kernels for a batch of N x 3 s utterances (min / median of
`reps` calls after a few settle calls).  A/B of a build knob = two processes on the same box:

    python tools/time_pitch.py 1000; SNF_PITCH_TEAM=1 python tools/time_pitch.py 1000
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shennong_amd import _backend, synth
from shennong_amd.processor import KaldiPitchProcessor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
waves = synth.utterances(0, n, 48000)
plan = _backend.get_plan(KaldiPitchProcessor()._build_options())
nf = plan.num_frames(48000)
soff = np.arange(n + 1, dtype=np.int64) * 48000
foff = np.arange(n + 1, dtype=np.int64) * nf
d_wave = _backend.DeviceBuffer(waves.nbytes)
d_wave.upload(waves)
d_out = _backend.DeviceBuffer(nf * n * plan.ndims * 4)
for _ in range(3):
    plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
rows = []
for _ in range(reps):
    plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
    rows.append([plan.last_kernel_ms(i) for i in range(0, 6) if i == 0 or plan.kernel_name(i)])
rows = np.array(rows)
names = ['total'] + [plan.kernel_name(i) for i in range(1, 6) if plan.kernel_name(i)]
out = np.empty((nf * n, plan.ndims), dtype=np.float32)
d_out.download(out)
print('utts %d team %s checksum %.6f' % (n, os.environ.get('SNF_PITCH_TEAM', '-'), float(out.astype(np.float64).sum())))
for k, name in enumerate(names):
    print('  %-28s min %.3f  median %.3f ms' % (name[:28], rows[:, k].min(), np.median(rows[:, k])))
