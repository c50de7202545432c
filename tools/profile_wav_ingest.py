#!/usr/bin/env python
"""Where the time of a corpus of WAV FILES goes (SURVEY.md 8f rank 4, int16 wav ingest): N files of 3 s written to a
directory, then the index (one header scan per file), Utterances.pin() and the features of the whole corpus from
the files and from the pinned index.

    python tools/profile_wav_ingest.py [n_files] [directory]
"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import scipy.io.wavfile  # noqa: E402
from shennong_amd import Utterances, pipeline, synth  # noqa: E402
from shennong_amd.logger import get_logger  # noqa: E402
from shennong_amd.processor import FilterbankProcessor  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
where = tempfile.mkdtemp(dir=sys.argv[2] if len(sys.argv) > 2 else None)
waves = synth.utterances(0, min(n, 500), 48000)
t0 = time.perf_counter()
for i in range(n):
    scipy.io.wavfile.write(os.path.join(where, f'u{i:06d}.wav'), 16000, waves[i % len(waves)])
print('wrote %d files in %.2f s' % (n, time.perf_counter() - t0))


def timed(what, fn, reps=3):
    best, out = None, None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print('%-44s %.3f s = %6.1f us per file = %5.1f h of audio per second' % (
        what, best, best / n * 1e6, n * 3.0 / 3600.0 / best), flush=True)
    return out


items = [(f'u{i:06d}', os.path.join(where, f'u{i:06d}.wav'), f's{i % 50:02d}') for i in range(n)]
index = timed('Utterances(...) (header scans)', lambda: Utterances(items))
quiet = get_logger('ingest', 'error')
fbank = FilterbankProcessor(num_bins=40, dither=0)
cfg = pipeline.get_default_config('filterbank', with_cmvn=True, with_delta=True)
cfg['filterbank']['num_bins'] = 40
cfg['filterbank']['dither'] = 0
fbank.process_all(index)
timed('process_all from files', lambda: fbank.process_all(index))
for njobs in (1, 4, 8):
    timed('extract_features from files, njobs %d' % njobs,
          lambda: pipeline.extract_features(cfg, index, njobs=njobs, log=quiet))
pinned = timed('Utterances.pin()', lambda: index.pin(), reps=2)
timed('process_all from the pinned index', lambda: fbank.process_all(pinned))
timed('extract_features from the pinned index', lambda: pipeline.extract_features(cfg, pinned, log=quiet))
# the streamed pipeline (by-speaker CMVN: two passes, the audio of the first kept in HBM) straight from the files
cfg5 = pipeline.get_default_config('filterbank', with_pitch='kaldi', with_cmvn=True, with_delta=True)
cfg5['filterbank']['num_bins'] = 40
cfg5['filterbank']['dither'] = 0
cfg5['cmvn']['by_speaker'] = True
for njobs in (1, 2):
    timed('extract_features_streamed from files, njobs %d' % njobs,
          lambda: pipeline.extract_features_streamed(cfg5, index, lambda f: None, njobs=njobs, log=quiet), reps=2)
timed('extract_features_streamed from the pinned index',
      lambda: pipeline.extract_features_streamed(cfg5, pinned, lambda f: None, log=quiet), reps=2)
# files in, Kaldi archive out: the whole corpus run
from shennong_amd.serializers import KaldiStreamWriter  # noqa: E402
for double in (False, True):
    def run():
        out = tempfile.mkdtemp(dir=where)
        with KaldiStreamWriter(os.path.join(out, 'corpus.ark'), double=double) as writer:
            pipeline.extract_features_streamed(cfg5, index, writer.write, njobs=2, log=quiet)
        size = os.path.getsize(os.path.join(out, 'corpus.ark')) / 1e9
        shutil.rmtree(out)
        return size
    size = timed('WAV files -> pipeline -> %s archive' % ('double' if double else 'float'), run, reps=2)
    print('   (%.2f GB of matrices)' % size)
shutil.rmtree(where)
