"""Dev helper: Kaldi's "hires" MFCC (40 bins, 40 cepstra) and MFCC over 80 bins at 16 kHz, 10 000 x 3 s utterances:
filterbank kernel + mfcc_dct_kernel against the generic kernel (SNF_DISABLE_MFCC_VIA_FBANK=1)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from shennong_amd import _backend, synth
from shennong_amd.processor import MfccProcessor
n_utts, ns = 10000, 48000
base = synth.utterances(0, 20, ns, 16000)
waves = np.ascontiguousarray(np.tile(base, (n_utts // 20, 1)))
d_wave = _backend.DeviceBuffer(waves.nbytes)
d_wave.upload(waves)
for opts in (dict(num_bins=40, num_ceps=40), dict(num_bins=80, num_ceps=40), dict()):
    proc = MfccProcessor(dither=0, **opts)
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
    names = [(plan.kernel_name(k), round(plan.last_kernel_ms(k), 4)) for k in range(1, 5) if plan.kernel_name(k)]
    print('MFCC', opts, '%d frames: call %.4f ms (min %.4f)' % (fpu * n_utts, np.median(ks), np.min(ks)), names, flush=True)
    d_out.free()
