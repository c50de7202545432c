#!/usr/bin/env python
"""profiles/r04_f64_report.txt: the C oracle against its second statements on CPU (no GPU needed).

    python tools/f64_report.py > profiles/r04_f64_report.txt

Per family: max |oracle - float64 restatement| (oracle/spec_f64.py, no shared code), the absolute tolerance that
would be needed on top of the north_star's 1e-4 relative one, and the fraction of elements inside the pure
relative tolerance; then the oracle against the reference's own PLP glue (tests/golden/reference_plp_glue.npz).
"""
import ast
import os
import sys

import numpy as np
import scipy.io.wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc, spec_f64  # noqa: E402
from shennong_amd import _abi, synth  # noqa: E402
from shennong_amd.processor import (  # noqa: E402
    FilterbankProcessor, KaldiPitchPostProcessor, MfccProcessor, PlpProcessor, SpectrogramProcessor)


def row(name, got, want, rtol=1e-4):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want)
    inside = float((err <= rtol * np.abs(want)).mean())
    print('%-44s max abs %.3g   needed atol at rtol %.0e: %.3g   inside pure rtol: %.4f   (%d values)' % (
        name, err.max(), rtol, max((err - rtol * np.abs(want)).max(), 0.0), inside, got.size))


def main():
    wave = scipy.io.wavfile.read(os.path.join(ROOT, 'tests', 'golden', 'test.wav'))[1]
    waves = [wave] + synth.ragged_utterances(900, 6, min_s=0.5, max_s=1.5)
    print('C oracle (oracle/kaldi_oracle.c, float32 with Kaldi\'s summation orders) against the float64 numpy')
    print('restatements of oracle/spec_f64.py on test.wav + 6 synthetic utterances; dither 0\n')

    def over(proc, fn):
        got = np.concatenate([orc.compute(proc._build_options(), w) for w in waves])
        return got, np.concatenate([fn(w) for w in waves])
    row('fbank-40', *over(FilterbankProcessor(dither=0, num_bins=40), lambda w: spec_f64.features(w, kind='fbank', num_bins=40)))
    row('MFCC-13', *over(MfccProcessor(dither=0), lambda w: spec_f64.features(w, kind='mfcc')))
    row('spectrogram-257', *over(SpectrogramProcessor(dither=0), lambda w: spec_f64.features(w, kind='spectrogram')))
    row('PLP-13', *over(PlpProcessor(dither=0), spec_f64.plp))
    row('PLP-13 + RASTA', *over(PlpProcessor(dither=0, rasta=True), lambda w: spec_f64.plp(w, use_rasta=True)))
    for warp in (0.85, 1.2):
        got = np.concatenate([orc.compute(PlpProcessor(dither=0)._build_options(), w, warp) for w in waves])
        row('PLP-13, VTLN warp %.2f' % warp, got, np.concatenate([spec_f64.plp(w, warp=warp) for w in waves]))
        got = np.concatenate([orc.compute(FilterbankProcessor(dither=0, num_bins=40)._build_options(), w, warp) for w in waves])
        w40 = spec_f64.mel_banks_vtln(40, 16000.0, 512, warp=warp)[0]
        want = []
        for w in waves:
            shift, length, padded = spec_f64.frame_geometry(16000, 0.01, 0.025)
            x = spec_f64.extract_frames(w, shift, length)
            x = x - x.mean(axis=1, keepdims=True)
            y = x.copy()
            y[:, 1:] = x[:, 1:] - 0.97 * x[:, :-1]
            y[:, 0] = x[:, 0] - 0.97 * x[:, 0]
            spec = np.fft.rfft(y * spec_f64.window_function(length)[None, :], n=padded, axis=1)
            want.append(np.log(np.maximum((spec.real ** 2 + spec.imag ** 2)[:, :256] @ w40.T, spec_f64.EPS32)))
        row('fbank-40, VTLN warp %.2f' % warp, got, np.concatenate(want))
    mfcc = orc.compute(MfccProcessor(dither=0)._build_options(), wave)
    row('delta order 2 window 2 (13 -> 39)', orc.deltas(mfcc, 2, 2), spec_f64.delta(mfcc, 2, 2), 1e-5)
    stats = np.zeros((2, 14))
    orc.cmvn_accumulate(mfcc, stats=stats)
    row('CMVN statistics', stats, spec_f64.cmvn_stats(mfcc), 1e-7)
    row('CMVN apply (mean + variance)', orc.cmvn_apply(mfcc, stats), spec_f64.cmvn_apply(mfcc, stats), 1e-5)
    row('sliding CMVN (600 / 100, centred)', orc.sliding_cmn(mfcc), spec_f64.sliding_cmvn(mfcc), 1e-5)
    row('sliding CMVN (30 / 10, causal, variance)',
        orc.sliding_cmn(mfcc, center=False, cmn_window=30, min_window=10, normalize_variance=True),
        spec_f64.sliding_cmvn(mfcc, center=False, cmn_window=30, min_window=10, normalize_variance=True), 1e-5)
    raw = orc.pitch(_abi.default_pitch_options(), wave)
    post = KaldiPitchPostProcessor(delta_pitch_noise_stddev=0, add_raw_log_pitch=True)
    row('pitch post-processing (4 columns, noise 0)', orc.process_pitch(post._options, raw),
        spec_f64.process_pitch(raw, add_raw_log_pitch=True), 1e-5)

    print('\nC oracle against the reference\'s OWN PLP control flow (shennong/processor/plp.py:171-260, :510-626,')
    print('run in the build container over numpy stand-ins of the pykaldi primitives: glue pinned, primitives')
    print('are stand-ins; tests/golden/make_golden_plp.py) on test.wav\n')
    fixture = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_plp_glue.npz'))
    for key in fixture.files:
        if key.startswith('case_'):
            opts, warp = ast.literal_eval(str(fixture['opts_' + key[5:]]))
            got = orc.compute(PlpProcessor(dither=0, **opts)._build_options(), wave, warp)
            row('%s %s' % (opts or 'defaults', '' if warp == 1.0 else 'warp %g' % warp), got, fixture[key])


if __name__ == '__main__':
    main()
