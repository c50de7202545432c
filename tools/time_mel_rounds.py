"""Dev helper: kernel time against the number of mel bins at one sample rate (python tools/time_mel_rounds.py 32000)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from shennong_amd import _backend, synth
from shennong_amd.processor import FilterbankProcessor, MfccProcessor
sr = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
n_utts = 2000
ns = 3 * sr
base = synth.utterances(0, 20, ns, sr)
waves = np.ascontiguousarray(np.tile(base, (n_utts // 20, 1)))
d_wave = _backend.DeviceBuffer(waves.nbytes); d_wave.upload(waves)
for cls, opts in ((FilterbankProcessor, dict(num_bins=8)), (FilterbankProcessor, dict(num_bins=16)), (FilterbankProcessor, dict(num_bins=23)), (FilterbankProcessor, dict(num_bins=40)), (FilterbankProcessor, dict(num_bins=64)), (FilterbankProcessor, dict(num_bins=80)), (FilterbankProcessor, dict(num_bins=128)), (MfccProcessor, dict()), (MfccProcessor, dict(num_bins=40))):
    proc = cls(sample_rate=sr, dither=0, **opts)
    try:
        plan = _backend.get_plan(proc._build_options())
    except RuntimeError as err:   # (Kaldi refuses banks with an empty bin)
        print(cls.__name__, opts, 'refused:', err)
        continue
    fpu = plan.num_frames(ns)
    soff = np.arange(n_utts + 1, dtype=np.int64) * ns
    foff = np.arange(n_utts + 1, dtype=np.int64) * fpu
    d_out = _backend.DeviceBuffer(fpu * n_utts * plan.ndims * 4)
    for _ in range(5): plan.run_device(d_wave.ptr, soff, foff, d_out.ptr)
    ks = []
    for _ in range(10):
        plan.run_device(d_wave.ptr, soff, foff, d_out.ptr); ks.append(plan.last_kernel_ms(0))
    print(cls.__name__, opts, plan.kernel_name(1), 'median %.4f min %.4f' % (np.median(ks), np.min(ks)), flush=True)
    d_out.free()
