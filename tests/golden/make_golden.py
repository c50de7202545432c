#!/usr/bin/env python
"""Generates tests/golden/reference_numpy.npz from the parts of the reference that run here.

Run IN THE BUILD CONTAINER ONLY (needs /root/reference; never on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

`import shennong` fails here (pykaldi, sox, pydub, h5features... are absent), so the
missing third-party modules are stubbed with MagicMock; only the reference's pure numpy/scipy code is
executed: RastaFilter.filter, _lpc2cepstrum (shennong/processor/plp.py:64-168), Audio.astype
(shennong/audio.py:469-518), Features.concatenate / validate (shennong/features.py:298-437) and
dict_equal.  The fixture holds inputs and the reference's outputs only (no reference source).
"""
import os
import sys
import types
from unittest import mock

import numpy as np

sys.dont_write_bytecode = True
for name in ['sox', 'pydub', 'kaldi', 'kaldi.base', 'kaldi.base.math', 'kaldi.feat',
             'kaldi.feat.window', 'kaldi.feat.mel', 'kaldi.feat.fbank', 'kaldi.feat.mfcc',
             'kaldi.feat.plp', 'kaldi.feat.spectrogram', 'kaldi.feat.pitch',
             'kaldi.feat.functions', 'kaldi.matrix', 'kaldi.matrix.common',
             'kaldi.matrix.functions', 'kaldi.transform', 'kaldi.transform.cmvn',
             'kaldi.ivector', 'kaldi.gmm', 'kaldi.util', 'kaldi.util.table', 'kaldi.util.io',
             'kaldi.transform.lvtln', 'kaldi.transform.mllr', 'kaldi.gmm.am', 'kaldi.gmm.full',
             'h5features', 'json_tricks', 'tensorflow', 'tensorflow.keras',
             'tensorflow.keras.layers', 'tensorflow.keras.models', 'hmmlearn', 'hmmlearn.hmm',
             'joblib', 'pkg_resources', 'yaml', 'kaldi.ivector.plda', 'kaldi.gmm.diag']:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:  # noqa
            sys.modules[name] = mock.MagicMock()
sys.path.insert(0, '/root/reference')

from shennong.processor.plp import RastaFilter, _lpc2cepstrum  # noqa
from shennong.audio import Audio  # noqa
from shennong.features import Features  # noqa

out = {}
rng = np.random.default_rng(20260927)

# ---- RASTA (do_log=True on positive float32 mel energies; do_log=False like test_plp.py:108-124)
mel = np.exp(rng.normal(8.0, 2.0, size=(40, 23))).astype(np.float32)
flt = RastaFilter(23)
out['rasta_in'] = mel
out['rasta_out'] = np.array([flt.filter(m.copy(), do_log=True) for m in mel])
short = mel[:3]
flt = RastaFilter(23)
out['rasta_short_out'] = np.array([flt.filter(m.copy(), do_log=True) for m in short])
sin = np.sin(2 * np.pi * np.arange(16000 * 0.05) * 200 / 16000).astype(np.float32)[5:]
pulse = np.zeros((sin.shape[0],))
pulse[0] = 1
data = np.dstack((sin, rng.random(sin.shape), pulse)).squeeze().astype(np.float32)
flt = RastaFilter(3)
out['rasta_nolog_in'] = data
out['rasta_nolog_out'] = np.array([flt.filter(s, do_log=False) for s in data])

# ---- _lpc2cepstrum on seeded LPC vectors.  The reference calls it on pykaldi Vectors: indexing
# returns a Python float (double arithmetic), assignment rounds to float32 storage.
class KaldiLikeVector:
    def __init__(self, data):
        self.a = np.asarray(data, dtype=np.float32).copy()

    def __getitem__(self, i):
        return float(self.a[i])

    def __setitem__(self, i, v):
        self.a[i] = np.float32(v)


lpcs = (rng.normal(0, 0.5, size=(8, 12))).astype(np.float32)
ceps = np.zeros_like(lpcs)
for i in range(lpcs.shape[0]):
    cep = KaldiLikeVector(np.zeros(12))
    _lpc2cepstrum(12, KaldiLikeVector(lpcs[i]), cep)
    ceps[i] = cep.a
out['lpc_in'] = lpcs
out['lpc_out'] = ceps

# ---- Audio.astype
f32 = rng.uniform(-1, 1, size=200).astype(np.float32)
f32[:4] = [1.0, -1.0, 0.99996948, 0.5]
f64 = f32.astype(np.float64)
i32 = (rng.integers(-2**30, 2**30, size=200)).astype(np.int32)
i16 = (rng.integers(-2**15, 2**15, size=200)).astype(np.int16)
out['astype_f32'] = f32
out['astype_f64'] = f64
out['astype_i32'] = i32
out['astype_i16'] = i16
with np.errstate(all='ignore'):
    out['astype_f32_to_i16'] = Audio(f32, 16000, validate=False).astype(np.int16).data
    out['astype_f64_to_i16'] = Audio(f64, 16000, validate=False).astype(np.int16).data
    out['astype_i32_to_i16'] = Audio(i32, 16000, validate=False).astype(np.int16).data
    out['astype_i16_to_f32'] = Audio(i16, 16000, validate=False).astype(np.float32).data
    # int16 -> int32 (audio.py:499-500: `data * 2**15`) raises OverflowError under numpy >= 2
    # (NEP 50); under the reference's numpy 1.x it is data.astype(int32) * 32768: not captured here

# ---- Features.concatenate with tolerance + properties merging
d1 = rng.normal(size=(10, 3)).astype(np.float32)
d2 = rng.normal(size=(12, 2)).astype(np.float32)
t1 = np.vstack((np.arange(10) * 0.01, np.arange(10) * 0.01 + 0.025)).T
t2 = np.vstack((np.arange(12) * 0.01, np.arange(12) * 0.01 + 0.025)).T
fa = Features(d1, t1, properties={'pipeline': [{'name': 'mfcc', 'columns': [0, 2]}], 'mfcc': {'a': 1}})
fb = Features(d2, t2, properties={'pipeline': [{'name': 'pitch', 'columns': [0, 1]}], 'pitch': {'b': 2}})
fc = fa.concatenate(fb, tolerance=2)
out['concat_d1'], out['concat_d2'], out['concat_t1'], out['concat_t2'] = d1, d2, t1, t2
out['concat_data'], out['concat_times'] = fc.data, fc.times
out['concat_pipeline'] = np.array(repr(fc.properties['pipeline']))
try:
    fa.concatenate(fb, tolerance=1)
except ValueError as err:
    out['concat_err_tol'] = np.array(str(err))
try:
    fa.concatenate(fb)
except ValueError as err:
    out['concat_err_notol'] = np.array(str(err))

# ---- times of the mel processors: float64 multiples of the float32 shift (processor/base.py:264-268)
shift, length = np.float32(10.0 / 1000.0), np.float32(25.0 / 1000.0)
out['times_140'] = np.vstack((np.arange(140) * shift, np.arange(140) * shift + length)).T

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_numpy.npz')
np.savez_compressed(dst, **out)
print('wrote', dst, {k: getattr(v, 'shape', None) for k, v in out.items()})
