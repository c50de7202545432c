"""The C oracle against a second, independent statement of every family (CPU only).

* oracle/spec_f64.py: float64 numpy restatements written from the algorithm descriptions, sharing no code
  with oracle/kaldi_oracle.c (round 3: fbank / MFCC / spectrogram / pitch, tests/test_oracle_pins.py; round
  4: PLP + RASTA, VTLN-warped mel banks, delta, CMVN, sliding CMVN, pitch post-processing).  A transcription
  error in either shows up as a difference far above float32 round-off.
* tests/golden/reference_plp_glue.npz: outputs of the REFERENCE'S OWN PLP control flow
  (shennong/processor/plp.py:171-260, :510-626, RASTA included) run in the build container with numpy
  stand-ins for the pykaldi primitives (tests/golden/make_golden_plp.py + kaldi_shim.py): the glue - order
  of operations, floors, which energy, slicing, HTK reorder - is pinned; the primitives are stand-ins.
The bounds are what float32 round-off of the oracle explains (printed by tools/f64_report.py into
profiles/r04_f64_report.txt); coefficient values against real Kaldi stay unpinned (no Kaldi offline).
"""
import ast
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as orc, spec_f64
from shennong_amd import _abi, synth
from shennong_amd.processor import (
    FilterbankProcessor, KaldiPitchPostProcessor, MfccProcessor, PlpProcessor)


def _excess(got, want, rtol=1e-4):
    """largest |got - want| - rtol |want|: what an absolute tolerance has to cover at the north_star's rtol"""
    err = np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64))
    return float((err - rtol * np.abs(want)).max()), float(err.max())


@pytest.mark.parametrize('opts, f64', [
    (dict(), dict()),
    (dict(use_energy=False), dict(use_energy=False)),
    (dict(raw_energy=False), dict(raw_energy=False)),
    (dict(rasta=True), dict(use_rasta=True)),
    (dict(htk_compat=True), dict(htk_compat=True)),
    (dict(num_ceps=5, lpc_order=8), dict(num_ceps=5, lpc_order=8)),
    (dict(cepstral_lifter=0, cepstral_scale=2.0), dict(cepstral_lifter=0, cepstral_scale=2.0)),
    (dict(snip_edges=False), dict(snip_edges=False)),
    (dict(num_bins=30, low_freq=100, high_freq=-300), dict(num_bins=30, low_freq=100.0, high_freq=-300.0)),
    (dict(window_type='hamming', remove_dc_offset=False, preemph_coeff=0.0),
     dict(window='hamming', remove_dc=False, preemph=0.0)),
])
def test_plp_against_float64(wave, opts, f64):
    """reference plp.py:510-626"""
    got = orc.compute(PlpProcessor(dither=0, **opts)._build_options(), wave)
    want = spec_f64.plp(wave, **f64)
    assert got.shape == want.shape
    excess, worst = _excess(got, want)
    assert excess < 2e-5 and worst < 1e-4, (opts, excess, worst)


@pytest.mark.parametrize('warp', [0.8, 0.9, 1.1, 1.25])
@pytest.mark.parametrize('kind', ['plp', 'fbank'])
def test_vtln_against_float64(wave, warp, kind):
    """reference processor/base.py:376-406 (MelBanks with a VTLN warp)"""
    if kind == 'plp':
        got = orc.compute(PlpProcessor(dither=0)._build_options(), wave, warp)
        want = spec_f64.plp(wave, warp=warp)
        excess, worst = _excess(got, want)
        assert excess < 5e-5 and worst < 1e-4, (warp, excess, worst)
        return
    _, _, got, centers = orc.mel_banks(
        _abi.MelOptions(num_bins=40, low_freq=20, high_freq=0, vtln_low=100, vtln_high=-500),
        _abi.default_frame_options(), warp)
    want, want_centers = spec_f64.mel_banks_vtln(40, 16000.0, 512, warp=warp)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 5e-5, np.abs(got - want).max()
    assert np.allclose(centers, want_centers, rtol=2e-6)
    # the warped banks move: the test would pass trivially if the warp were ignored on both sides
    plain, _ = spec_f64.mel_banks_vtln(40, 16000.0, 512)
    assert np.abs(want - plain).max() > 0.05


def test_rasta_against_float64():
    """reference plp.py:64-146, including an utterance shorter than the 4-frame warm-up"""
    mel = np.exp(np.random.default_rng(3).normal(8.0, 2.0, size=(60, 23))).astype(np.float32)
    for rows in (60, 5, 4, 3, 1):
        got = orc.rasta(mel[:rows].copy())
        want = spec_f64.rasta(mel[:rows])
        assert np.allclose(got, want, rtol=2e-6, atol=0), rows


@pytest.mark.parametrize('order, window', [(1, 1), (2, 2), (1, 3), (3, 1), (5, 2)])
def test_delta_against_float64(wave, order, window):
    """reference postprocessor/delta.py:113-136; rows shorter than the filter included"""
    feats = orc.compute(MfccProcessor(dither=0)._build_options(), wave)
    for rows in (feats.shape[0], 3, 1):
        got = orc.deltas(feats[:rows], order, window)
        want = spec_f64.delta(feats[:rows], order, window)
        assert got.shape == want.shape
        assert _excess(got, want, 1e-5)[0] < 2.5e-6
    for i, s in enumerate(spec_f64.delta_scales(order, window)):
        assert np.allclose(orc.delta_scales(order, window)[i], s, rtol=1e-6, atol=1e-9)


def test_cmvn_against_float64(wave):
    """reference postprocessor/cmvn.py:180-282"""
    feats = orc.compute(MfccProcessor(dither=0)._build_options(), wave)
    weights = np.random.default_rng(1).random(feats.shape[0]).astype(np.float32)
    for w in (None, weights):
        stats = np.zeros((2, feats.shape[1] + 1))
        orc.cmvn_accumulate(feats, weights=w, stats=stats)
        want = spec_f64.cmvn_stats(feats, w)
        assert np.allclose(stats, want, rtol=1e-7, atol=0)
        for norm_vars in (True, False):
            for reverse in (False, True):
                got = orc.cmvn_apply(feats, stats, norm_vars=norm_vars, reverse=reverse)
                ref = spec_f64.cmvn_apply(feats, stats, norm_vars=norm_vars, reverse=reverse)
                assert _excess(got, ref, 1e-5)[0] < 5e-6, (norm_vars, reverse)


@pytest.mark.parametrize('kw', [
    dict(), dict(center=False), dict(cmn_window=40, min_window=10),
    dict(center=False, cmn_window=30, min_window=10, normalize_variance=True),
    dict(normalize_variance=True, cmn_window=7, min_window=50), dict(cmn_window=1000, min_window=100),
])
def test_sliding_cmvn_against_float64(wave, kw):
    """reference postprocessor/cmvn.py:285-392 (SlidingWindowCmvnPostProcessor)"""
    feats = orc.compute(MfccProcessor(dither=0)._build_options(), wave)
    got = orc.sliding_cmn(feats, **kw)
    want = spec_f64.sliding_cmvn(feats, **kw)
    assert _excess(got, want, 1e-5)[0] < 5e-6, kw


@pytest.mark.parametrize('kw, f64', [
    (dict(), dict()),
    (dict(add_raw_log_pitch=True), dict(add_raw_log_pitch=True)),
    (dict(add_pov_feature=False, add_normalized_log_pitch=False, add_raw_log_pitch=True),
     dict(add_pov_feature=False, add_normalized_log_pitch=False, add_raw_log_pitch=True)),
    (dict(normalization_left_context=10, normalization_right_context=3, delta_window=4, pitch_scale=1.5,
          pov_scale=0.7, pov_offset=0.2, delta_pitch_scale=3.0),
     dict(left_context=10, right_context=3, delta_window=4, pitch_scale=1.5, pov_scale=0.7, pov_offset=0.2,
          delta_pitch_scale=3.0)),
])
def test_pitch_post_against_float64(wave, kw, f64):
    """reference pitch_kaldi.py:497-540, noise term at 0"""
    raw = orc.pitch(_abi.default_pitch_options(), wave)
    got = orc.process_pitch(KaldiPitchPostProcessor(delta_pitch_noise_stddev=0, **kw)._options, raw)
    want = spec_f64.process_pitch(raw, **f64)
    assert got.shape == want.shape
    assert _excess(got, want, 1e-5)[0] < 7e-6, kw


def test_plp_against_the_references_own_glue(wave):
    """tests/golden/reference_plp_glue.npz: PlpProcessor.process of the reference itself over numpy stand-ins
    of the pykaldi primitives (13 option sets incl. RASTA, VTLN warps, centred frames)"""
    fixture = np.load(os.path.join(GOLDEN, 'reference_plp_glue.npz'))
    names = [k[5:] for k in fixture.files if k.startswith('case_')]
    assert len(names) == 13
    for name in names:
        opts, warp = ast.literal_eval(str(fixture['opts_' + name]))
        got = orc.compute(PlpProcessor(dither=0, **opts)._build_options(), wave, warp)
        want = fixture['case_' + name]
        assert got.shape == want.shape, name
        excess, worst = _excess(got, want)
        assert excess < 5e-5 and worst < 1e-4, (name, excess, worst)


def test_fraction_within_pure_rtol():
    """how much of a family sits inside the north_star's flat 1e-4 relative tolerance
    when no absolute term helps (oracle against float64; zero crossings of signed cepstra are what is left)"""
    waves = synth.ragged_utterances(900, 4, min_s=0.5, max_s=1.0)
    inside = {}
    for kind, proc, f64 in (
            ('fbank', FilterbankProcessor(dither=0, num_bins=40), dict(kind='fbank', num_bins=40)),
            ('mfcc', MfccProcessor(dither=0), dict(kind='mfcc')),
            ('plp', PlpProcessor(dither=0), None)):
        ok = total = 0
        for w in waves:
            got = orc.compute(proc._build_options(), w)
            want = spec_f64.plp(w) if f64 is None else spec_f64.features(w, **f64)
            ok += int((np.abs(got - want) <= 1e-4 * np.abs(want)).sum())
            total += got.size
        inside[kind] = ok / total
    assert inside['fbank'] == 1.0, inside
    assert inside['mfcc'] > 0.98 and inside['plp'] > 0.98, inside   # (99.2 % / 98.6 %: signed cepstra near zero)


def test_random_option_sets(monkeypatch):
    """tests/tools/fuzz_oracle_f64.py: random option sets of the four spectral families, the C oracle against the
    float64 restatement (1 650 cases in profiles/r04_f64_report.txt; a short run here)"""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'fuzz_oracle_f64.py')
    spec = importlib.util.spec_from_file_location('fuzz_oracle_f64', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr('sys.argv', ['fuzz_oracle_f64.py', '40', '11'])
    assert mod.main() == 0


def test_random_post_processing_cases(monkeypatch):
    """tests/tools/fuzz_oracle_f64_post.py: random delta / CMVN / sliding CMVN / pitch post-processing cases, the C
    oracle against the float64 restatement (2 300 cases in profiles/r04_f64_report.txt; a short run here)"""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tools', 'fuzz_oracle_f64_post.py')
    spec = importlib.util.spec_from_file_location('fuzz_oracle_f64_post', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr('sys.argv', ['fuzz_oracle_f64_post.py', '60', '12'])
    assert mod.main() == 0
