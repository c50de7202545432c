#!/usr/bin/env python
"""Writes tests/golden/reference_kaldi.npz: the outputs of the REFERENCE (bootphon/shennong on pykaldi) for
tests/golden/test.wav and test.8k.wav - the pin the C oracle has lacked since round 1 ("parity unpinned":
the reference's own tests hold no coefficient values and Kaldi cannot run in the build container).

Run it WHERE THE REFERENCE IS INSTALLED (its conda environment: shennong + shennong-pykaldi) - never in the
build container, never on the GPU box:

    python tests/golden/make_golden_kaldi.py            # from a checkout of this repository

It imports the reference, runs every deterministic processor on the two wav files (dither 0, delta-pitch
noise 0) and stores the float32 matrices together with the constructor arguments of every case, so that
tests/test_kaldi_pin.py can rebuild the same options on this side and compare oracle/kaldi_oracle.c (and,
under `-m gpu`, the HIP path) with them at the north_star's 1e-4.  Nothing of the reference but its
OUTPUTS goes into the file.  Reference call sites: shennong/processor/base.py:408-436 (mel processors),
spectrogram.py:90-143, plp.py:510-676, pitch_kaldi.py:260-302 and :497-540, energy.py:148-186,
postprocessor/delta.py:113-136, cmvn.py:180-282 and :399-470, vad.py:163-188.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# name -> (processor, constructor arguments, input: a wav file or the name of an earlier case, call arguments)
CASES = [
    ('spectrogram', 'spectrogram', {'dither': 0}, 'test.wav', {}),
    ('spectrogram_hamming_noraw', 'spectrogram', {'dither': 0, 'window_type': 'hamming', 'raw_energy': False},
     'test.wav', {}),
    ('fbank23', 'filterbank', {'dither': 0}, 'test.wav', {}),
    ('fbank40', 'filterbank', {'dither': 0, 'num_bins': 40}, 'test.wav', {}),
    ('fbank40_energy', 'filterbank', {'dither': 0, 'num_bins': 40, 'use_energy': True}, 'test.wav', {}),
    ('fbank40_linear_htk', 'filterbank', {'dither': 0, 'num_bins': 40, 'use_energy': True, 'use_log_fbank': False, 'htk_compat': True, 'use_power': False}, 'test.wav', {}),
    ('fbank40_centred', 'filterbank', {'dither': 0, 'num_bins': 40, 'snip_edges': False}, 'test.wav', {}),
    ('fbank40_warp1.1', 'filterbank', {'dither': 0, 'num_bins': 40}, 'test.wav', {'vtln_warp': 1.1}),
    ('fbank40_8k', 'filterbank', {'dither': 0, 'num_bins': 40, 'sample_rate': 8000}, 'test.8k.wav', {}),
    ('mfcc', 'mfcc', {'dither': 0}, 'test.wav', {}),
    ('mfcc_htk_noenergy', 'mfcc', {'dither': 0, 'htk_compat': True, 'use_energy': False}, 'test.wav', {}),
    ('mfcc_warp0.9_centred', 'mfcc', {'dither': 0, 'snip_edges': False}, 'test.wav', {'vtln_warp': 0.9}),
    ('plp', 'plp', {'dither': 0}, 'test.wav', {}),
    ('plp_rasta', 'plp', {'dither': 0, 'rasta': True}, 'test.wav', {}),
    ('plp_htk', 'plp', {'dither': 0, 'htk_compat': True}, 'test.wav', {}),
    ('energy', 'energy', {'dither': 0}, 'test.wav', {}),
    ('energy_sqrt_noraw', 'energy', {'dither': 0, 'raw_energy': False, 'compression': 'sqrt'}, 'test.wav', {}),
    ('pitch', 'kaldi_pitch', {}, 'test.wav', {}),
    ('pitch_8k', 'kaldi_pitch', {'sample_rate': 8000}, 'test.8k.wav', {}),
    ('pitch_post', 'kaldi_pitch_post', {'delta_pitch_noise_stddev': 0}, 'pitch', {}),
    ('pitch_post_raw', 'kaldi_pitch_post', {'delta_pitch_noise_stddev': 0, 'add_raw_log_pitch': True, 'add_pov_feature': False}, 'pitch', {}),
    ('delta_mfcc', 'delta', {}, 'mfcc', {}),
    ('delta_o3w3_fbank40', 'delta', {'order': 3, 'window': 3}, 'fbank40', {}),
    ('vad', 'vad', {}, 'energy', {}),
    ('sliding_cmvn_mfcc', 'sliding_window_cmvn', {}, 'mfcc', {}),
    ('sliding_cmvn_var', 'sliding_window_cmvn', {'cmn_window': 50, 'min_window': 20, 'normalize_variance': True, 'center': False}, 'mfcc', {}),
]


def main():
    import shennong
    from shennong import Audio
    from shennong.postprocessor.cmvn import CmvnPostProcessor, SlidingWindowCmvnPostProcessor
    from shennong.postprocessor.delta import DeltaPostProcessor
    from shennong.postprocessor.vad import VadPostProcessor
    from shennong.processor.energy import EnergyProcessor
    from shennong.processor.filterbank import FilterbankProcessor
    from shennong.processor.mfcc import MfccProcessor
    from shennong.processor.pitch_kaldi import KaldiPitchPostProcessor, KaldiPitchProcessor
    from shennong.processor.plp import PlpProcessor
    from shennong.processor.spectrogram import SpectrogramProcessor
    classes = {
        'spectrogram': SpectrogramProcessor, 'filterbank': FilterbankProcessor, 'mfcc': MfccProcessor,
        'plp': PlpProcessor, 'energy': EnergyProcessor, 'kaldi_pitch': KaldiPitchProcessor,
        'kaldi_pitch_post': KaldiPitchPostProcessor, 'delta': DeltaPostProcessor, 'vad': VadPostProcessor,
        'sliding_window_cmvn': SlidingWindowCmvnPostProcessor}
    audios = {name: Audio.load(os.path.join(HERE, name)) for name in ('test.wav', 'test.8k.wav')}
    out, feats, meta = {}, {}, {}
    for name, kind, params, source, call in CASES:
        proc = classes[kind](**params)
        result = proc.process(audios[source] if source in audios else feats[source], **call)
        feats[name] = result
        out[name] = np.ascontiguousarray(result.data)
        out[name + '__times'] = np.ascontiguousarray(result.times)
        meta[name] = dict(processor=kind, params=params, source=source, call=call,
                          dtype=str(result.data.dtype), shape=list(result.data.shape))
    # CMVN: statistics accumulated over the MFCC of the utterance (with and without VAD weights), applied
    for tag, weights in (('cmvn_mfcc', None), ('cmvn_mfcc_vad', feats['vad'].data[:, 0])):
        cmvn = CmvnPostProcessor(feats['mfcc'].ndims)
        cmvn.accumulate(feats['mfcc'], weights=weights)
        out[tag + '__stats'] = np.asarray(cmvn.stats, dtype=np.float64)
        for norm_vars in (True, False):
            key = tag + ('' if norm_vars else '_meanonly')
            out[key] = np.ascontiguousarray(cmvn.process(feats['mfcc'], norm_vars=norm_vars).data)
            meta[key] = dict(processor='cmvn', params=dict(dim=int(feats['mfcc'].ndims)), source='mfcc',
                             call=dict(norm_vars=norm_vars), stats=tag + '__stats',
                             weights='vad' if weights is not None else None,
                             dtype=str(out[key].dtype), shape=list(out[key].shape))
    out['__cases__'] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8)
    out['__versions__'] = np.frombuffer(json.dumps({
        'shennong': getattr(shennong, '__version__', 'unknown'),
        'numpy': np.__version__, 'python': sys.version.split()[0]}).encode(), dtype=np.uint8)
    target = os.path.join(HERE, 'reference_kaldi.npz')
    np.savez_compressed(target, **out)
    print('wrote %s: %d cases' % (target, len(meta)))


if __name__ == '__main__':
    main()
