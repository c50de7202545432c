#!/usr/bin/env python
"""Where the host time of the streamed pipeline goes (BASELINE config 5 at a fraction of one GPU's share):
wall clock of extract_features_streamed over `hours` of 3 s utterances for pageable / pinned indexes and 1-4
batches in flight, then a cProfile of one form, by cumulative and by own time.

    python tools/profile_streamed.py [hours] [profile: pin|page|none] [njobs of the profiled run]
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402,F401
from shennong_amd import Audio, Utterances, pipeline, synth  # noqa: E402
from shennong_amd.logger import get_logger  # noqa: E402

hours = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
what = sys.argv[2] if len(sys.argv) > 2 else 'pin'
pjobs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
print(open('/proc/meminfo').read().split('\n')[0:3])
waves = synth.utterances(0, 2000, 48000)
n = int(hours * 1200)
cfg = pipeline.get_default_config('filterbank', with_pitch='kaldi', with_cmvn=True, with_delta=True)
cfg['filterbank']['num_bins'] = 40
cfg['filterbank']['dither'] = 0
cfg['cmvn']['by_speaker'] = True
audios = [Audio(waves[i], 16000, validate=False) for i in range(len(waves))]
index = Utterances([(f'u{i:06d}', audios[i % len(audios)], f's{i % 1000:04d}') for i in range(n)])
t0 = time.perf_counter()
pinned = index.pin()
print('pin(): %.2f s for %.1f GB' % (time.perf_counter() - t0, n * 96000 / 1e9))
quiet = get_logger('bench', 'error')
seen = [0]


def sink(feats):
    seen[0] += len(feats)
    seen.append(sum(f.nframes for f in feats.values()))


only = os.environ.get('ONLY')
for name, idx in (('pageable', index), ('pinned', pinned)):
    if only and name != only:
        continue
    for njobs in ((1, 2, 3, 4) if not only else (1, 2)):
        best = None
        for rep in range(3):
            st = pipeline.RunStats()
            t0 = time.perf_counter()
            pipeline.extract_features_streamed(cfg, idx, sink, log=quiet, njobs=njobs, stats=st)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        d = st.as_dict()
        print('%-8s njobs %d: %.3f s = %6.1f h/s (%.2f us per utterance); up %.2f GB wait %.3f s, down %.2f GB '
              'wait %.3f s, gpu %.3f s' % (name, njobs, best, hours / best, best / n * 1e6, d['bytes_up'] / 1e9,
                                           d['upload_wait_s'], d['bytes_down'] / 1e9, d['download_wait_s'],
                                           d['gpu_ms'] / 1e3), flush=True)
        from shennong_amd import _backend
        print('   result blocks:', _backend.RESULT_KINDS, 'held', _backend._ResultBlock._held, 'pool', [b[0] >> 20 for b in _backend.STAGING._free], flush=True)
if what != 'none':
    idx = pinned if what == 'pin' else index
    prof = cProfile.Profile()
    prof.enable()
    pipeline.extract_features_streamed(cfg, idx, sink, log=quiet, njobs=pjobs)
    prof.disable()
    st = pstats.Stats(prof)
    st.sort_stats('cumulative').print_stats(40)
    st.sort_stats('tottime').print_stats(30)
