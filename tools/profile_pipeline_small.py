#!/usr/bin/env python
"""Wall clock and cProfile of ONE extract_features call on 1 000 in-memory 3 s utterances (the bench's
pipeline_fbank_pitch_delta_cmvn leg), pageable and pinned index."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from shennong_amd import Audio, Utterances, pipeline, synth  # noqa: E402
from shennong_amd.logger import get_logger  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
waves = synth.utterances(0, n, 48000)
index = Utterances([(f'u{i}', Audio(waves[i], 16000, validate=False), f's{i % 20}') for i in range(n)])
cfg = pipeline.get_default_config('filterbank', with_pitch='kaldi', with_cmvn=True, with_delta=True)
cfg['filterbank']['num_bins'] = 40
cfg['filterbank']['dither'] = 0
cfg['cmvn']['with_vad'] = False
quiet = get_logger('bench', 'error')
for name, idx in (('pageable', index), ('pinned', index.pin())):
    pipeline.extract_features(cfg, idx, log=quiet)
    walls = []
    for _ in range(9):
        t0 = time.perf_counter()
        feats = pipeline.extract_features(cfg, idx, log=quiet)
        walls.append(time.perf_counter() - t0)
        del feats
    print('%-8s median %.2f ms, min %.2f ms' % (name, np.median(walls) * 1e3, min(walls) * 1e3), flush=True)
    prof = cProfile.Profile()
    prof.enable()
    for _ in range(5):
        pipeline.extract_features(cfg, idx, log=quiet)
    prof.disable()
    pstats.Stats(prof).sort_stats('tottime').print_stats(18)
