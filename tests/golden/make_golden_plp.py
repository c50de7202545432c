#!/usr/bin/env python
"""Generates tests/golden/reference_plp_glue.npz: the outputs of the reference's OWN PLP control flow
(shennong/processor/plp.py `PlpProcessor.process` -> `_compute` -> `_extract_window` / `_process_window` /
`_compute_frame`, RASTA included) on tests/golden/test.wav, with the pykaldi primitives it calls replaced by
the numpy stand-ins of kaldi_shim.py.  GLUE PINNED, PRIMITIVES ARE STAND-INS (see kaldi_shim.py).

Run IN THE BUILD CONTAINER ONLY (needs /root/reference; never on the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_plp.py

The fixture holds the option sets and the reference's outputs only (no reference source).
"""
import os
import sys
from unittest import mock

import numpy as np
import scipy.io.wavfile

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import kaldi_shim  # noqa: E402

kaldi_shim.install()
for name in ['sox', 'pydub', 'kaldi.feat.fbank', 'kaldi.feat.mfcc', 'kaldi.feat.spectrogram', 'kaldi.feat.pitch',
             'kaldi.transform', 'kaldi.transform.cmvn', 'kaldi.ivector', 'kaldi.gmm', 'kaldi.util',
             'kaldi.util.table', 'kaldi.util.io', 'kaldi.transform.lvtln', 'kaldi.transform.mllr',
             'kaldi.gmm.am', 'kaldi.gmm.full', 'h5features', 'json_tricks', 'tensorflow', 'tensorflow.keras',
             'tensorflow.keras.layers', 'tensorflow.keras.models', 'hmmlearn', 'hmmlearn.hmm', 'joblib',
             'pkg_resources', 'yaml', 'kaldi.ivector.plda', 'kaldi.gmm.diag']:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:  # noqa
            sys.modules[name] = mock.MagicMock()
sys.path.insert(0, '/root/reference')

from shennong.audio import Audio  # noqa: E402
from shennong.processor.plp import PlpProcessor  # noqa: E402

rate, wave = scipy.io.wavfile.read(os.path.join(HERE, 'test.wav'))
audio = Audio(wave, rate, validate=False)
CASES = [
    ('default', dict(), 1.0),
    ('no_energy', dict(use_energy=False), 1.0),
    ('post_energy', dict(raw_energy=False), 1.0),
    ('rasta', dict(rasta=True), 1.0),
    ('htk', dict(htk_compat=True), 1.0),
    ('htk_no_energy', dict(htk_compat=True, use_energy=False), 1.0),
    ('floor', dict(energy_floor=1.0e9), 1.0),
    ('ceps5', dict(num_ceps=5), 1.0),
    ('lifter0_scale2', dict(cepstral_lifter=0, cepstral_scale=2.0), 1.0),
    ('centred', dict(snip_edges=False), 1.0),
    ('hamming_nodc', dict(window_type='hamming', remove_dc_offset=False), 1.0),
    ('warp_0.9', dict(), 0.9),
    ('warp_1.15_rasta', dict(rasta=True), 1.15),
]
out = {}
for name, opts, warp in CASES:
    proc = PlpProcessor(dither=0, **opts)
    feats = proc.process(audio, vtln_warp=warp)
    out['case_' + name] = np.asarray(feats.data, dtype=np.float32)
    out['opts_' + name] = np.array(repr((opts, warp)))
dst = os.path.join(HERE, 'reference_plp_glue.npz')
np.savez_compressed(dst, **out)
print('wrote', dst, {k: v.shape for k, v in out.items() if k.startswith('case_')})
