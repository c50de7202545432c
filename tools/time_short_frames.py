"""Developer timing: fbank-40 / MFCC-13 kernel time on 8 kHz audio (frames pad to 256 samples).
Run twice, with and without SNF_DISABLE_FAST512=1, to compare the zero-extended 512-point fast
path with the generic LDS radix-2 kernel."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shennong_amd import _backend, synth  # noqa: E402
from shennong_amd.processor import FilterbankProcessor, MfccProcessor  # noqa: E402

n_utts = 4000
waves = list(synth.utterances(1, 200, 24000, 8000)) * (n_utts // 200)
for name, proc in (('fbank40_8k', FilterbankProcessor(sample_rate=8000, num_bins=40, dither=0)),
                   ('mfcc13_8k', MfccProcessor(sample_rate=8000, dither=0))):
    plan = _backend.get_plan(proc._build_options())
    out = plan.run(waves)
    ms = []
    for _ in range(5):
        plan.run(waves)
        ms.append(plan.last_kernel_ms(1))
    frames = sum(o.shape[0] for o in out)
    print(f'{name}: {plan.kernel_name(1)} {min(ms):.3f} ms / {frames} frames '
          f'({frames / min(ms) * 1e3:.3e} frames/s)  fast512_disabled={bool(os.environ.get("SNF_DISABLE_FAST512"))}')
