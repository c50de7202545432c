"""Developer measurement for BASELINE config 5: the full pipeline (fbank-40 + Kaldi pitch + delta +
CMVN by speaker) streamed over a synthetic corpus in bounded batches, features written to a Kaldi
archive.  Prints wall time, hours of audio per second of wall time, and where the time goes.

    python tools/stream_corpus.py [n_utterances] [batch_seconds] [out_dir]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shennong_amd import Audio, Utterances, pipeline, synth  # noqa: E402
from shennong_amd.logger import get_logger  # noqa: E402
from shennong_amd.serializers import KaldiStreamWriter  # noqa: E402

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
batch_seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1800.0
out_dir = sys.argv[3] if len(sys.argv) > 3 else tempfile.mkdtemp()

waves = synth.utterances(1, min(n_utts, 500), 48000)
index = Utterances([(f'utt{i:06d}', Audio(waves[i % len(waves)], 16000), f'spk{i % 50:02d}')
                    for i in range(n_utts)])
config = pipeline.get_default_config('filterbank', with_cmvn=True, with_delta=True, with_pitch='kaldi')
config['filterbank']['num_bins'] = 40
config['cmvn']['by_speaker'] = True
config['cmvn']['with_vad'] = False
log = get_logger('stream', 'error')
hours = n_utts * 3.0 / 3600.0

for label, make_sink in (('null sink', lambda: (None, lambda f: None)),
                         ('kaldi ark', lambda: (KaldiStreamWriter(os.path.join(out_dir, 'corpus.ark')),
                                                None)),
                         ('kaldi ark, float matrices',
                          lambda: (KaldiStreamWriter(os.path.join(out_dir, 'corpus32.ark'), double=False),
                                   None))):
    writer, sink = make_sink()
    if writer is not None:
        sink = writer.write
    pipeline.extract_features_streamed(config, Utterances(list(index)[:50]), lambda f: None, log=log)  # warm
    t0 = time.perf_counter()
    n = pipeline.extract_features_streamed(config, index, sink, max_batch_duration=batch_seconds, log=log)
    if writer is not None:
        writer.close()
    dt = time.perf_counter() - t0
    size = ''
    if writer is not None:
        size = ' %.2f GB written' % (os.path.getsize(writer._root + '.ark') / 1e9)
    print(f'{label}: {n} utterances ({hours:.2f} h of audio) in {dt:.2f} s wall = '
          f'{hours / dt:.3f} h of audio per second ({hours * 3600 / dt:.0f} x real time), '
          f'batches of {batch_seconds:.0f} s{size}')
