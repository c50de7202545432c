#!/usr/bin/env python
"""Throughput of serializers.KaldiStreamWriter on one pipeline-sized batch (4 800 utterances x 298 x 123 float32,
704 MB of rows) float and double matrices, next to the round-5 form of the
writer (one buffered write per matrix through `tobytes`, a deep copy of the properties per item, ONE indented JSON
dump at close) on the same box and directory.

    python tools/profile_ark_writer.py [directory] [utterances]
"""
import copy
import json
import os
import shutil
import struct
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from shennong_amd import Features, FeaturesCollection, serializers  # noqa: E402

where = sys.argv[1] if len(sys.argv) > 1 else None
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4800
block = np.random.default_rng(0).standard_normal((n * 298, 123)).astype(np.float32)
times = np.stack([np.arange(298) * 0.01, np.arange(298) * 0.01 + 0.025], axis=1)
props = {'filterbank': {'num_bins': 40, 'sample_rate': 16000.0, 'dither': 0.0, 'window_type': 'povey'},
         'cmvn': {'stats': np.random.rand(2, 41)}, 'delta': {'order': 2, 'window': 2},
         'pitch': {'min_f0': 50.0, 'max_f0': 400.0}, 'pitch_post': {'pitch_scale': 2.0},
         'pipeline': [{'name': 'filterbank', 'columns': [0, 39]}, {'name': 'cmvn', 'columns': [0, 39]},
                      {'name': 'delta', 'columns': [0, 119]}, {'name': 'pitch', 'columns': [120, 122]}]}


def collection():
    return FeaturesCollection(
        (f'u{i:06d}', Features._of_batch(block[298 * i:298 * (i + 1)], times, props,
                                         {'audio': {'file': None, 'sample_rate': 16000, 'duration': 3.0},
                                          'speaker': 's%04d' % (i % 1000)})) for i in range(n))


def round5_writer(features, root, double):
    """the writer as it was (kept here for the comparison only)"""
    kept = {}
    with open(root + '.ark', 'wb') as data, open(root + '.times.ark', 'wb') as tfile:
        for key, feat in features.items():
            for stream, mat, dbl in ((data, feat.data, double), (tfile, np.atleast_2d(feat.times), True)):
                mat = np.ascontiguousarray(mat, dtype=np.float64 if dbl else np.float32)
                stream.write(key.encode('utf-8') + b' ' + (b'\0BDM ' if dbl else b'\0BFM '))
                stream.write(b'\4' + struct.pack('<i', mat.shape[0]) + b'\4' + struct.pack('<i', mat.shape[1]))
                stream.write(mat.tobytes())
            entry = copy.deepcopy(feat.properties)
            entry['__dtype_data__'], entry['__dtype_times__'] = str(feat.dtype), str(feat.times.dtype)
            kept[key] = entry
    with open(root + '.properties.json', 'wt', encoding='utf-8') as stream:
        stream.write(json.dumps(kept, indent=4, cls=serializers._ArrayEncoder, ensure_ascii=False))


for double in (False, True):
    nbytes = block.nbytes * (2 if double else 1)
    d = tempfile.mkdtemp(dir=where)
    coll = collection()
    t0 = time.perf_counter()
    round5_writer(coll, os.path.join(d, 'old'), double)
    dt = time.perf_counter() - t0
    print('%-6s round 5 writer      : %.3f s = %.2f GB/s of matrices, %.0f us per utterance' % (
        'double' if double else 'float', dt, nbytes / dt / 1e9, dt / n * 1e6), flush=True)
    shutil.rmtree(d)
    for threads in (1,):
        d = tempfile.mkdtemp(dir=where)
        coll = collection()
        t0 = time.perf_counter()
        with serializers.KaldiStreamWriter(os.path.join(d, 'new.ark'), double=double) as writer:
            writer.write(coll)
        dt = time.perf_counter() - t0
        print('%-6s round 6 writer      : %.3f s = %.2f GB/s of matrices, %.0f us per utterance' % (
            'double' if double else 'float', dt, nbytes / dt / 1e9, dt / n * 1e6), flush=True)
        if not double:
            back = FeaturesCollection.load(os.path.join(d, 'new.ark'))
            assert list(back) == list(coll) and np.array_equal(back['u000007'].data, coll['u000007'].data)
            assert back['u000007'].properties['speaker'] == 's0007'
        shutil.rmtree(d)
