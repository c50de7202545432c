"""Randomised comparison of the C oracle (oracle/kaldi_oracle.c, float32) with the independent float64 restatement
(oracle/spec_f64.py) over random option sets - CPU only, no GPU needed.  A transcription error in either statement
shows up far above float32 round-off; what is compared is chosen so that round-off stays small (log-domain values
of frames with signal; elements near zero crossings and spectral nulls get an absolute term).

    python tests/tools/fuzz_oracle_f64.py [n_cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc, spec_f64  # noqa: E402
from shennong_amd import synth  # noqa: E402
from shennong_amd.processor import (  # noqa: E402
    EnergyProcessor, FilterbankProcessor, MfccProcessor, PlpProcessor, SpectrogramProcessor)


def random_case(rng):
    kind = str(rng.choice(['fbank', 'mfcc', 'plp', 'spectrogram', 'energy']))
    sr = int(rng.choice([8000, 16000, 16000, 22050, 32000, 44100]))
    frame_length = float(rng.choice([0.01, 0.02, 0.025, 0.025, 0.03, 0.05]))
    frame_shift = float(rng.choice([0.005, 0.01, 0.01, 0.015]))
    window = str(rng.choice(['povey', 'hamming', 'hanning', 'rectangular', 'blackman']))
    bc = float(rng.choice([0.42, 0.42, 0.3]))
    pow2 = bool(rng.integers(6) > 0)
    if not pow2 and int(sr * frame_length) % 2:   # (an odd transform size is an option error on both sides)
        pow2 = True
    common = dict(sample_rate=sr, frame_shift=frame_shift, frame_length=frame_length, window_type=window,
                  snip_edges=bool(rng.integers(2)), remove_dc_offset=bool(rng.integers(2)),
                  preemph_coeff=float(rng.choice([0.0, 0.9, 0.97])), dither=0, blackman_coeff=bc,
                  round_to_power_of_two=pow2)
    f64 = dict(sample_rate=sr, frame_shift=frame_shift, frame_length=frame_length, window=window,
               snip_edges=common['snip_edges'], remove_dc=common['remove_dc_offset'], preemph=common['preemph_coeff'])
    floor = float(rng.choice([0.0, 0.0, 1.0, 1.0e4]))
    if kind == 'energy':
        raw, comp = bool(rng.integers(2)), str(rng.choice(['log', 'sqrt', 'off']))
        return kind, EnergyProcessor(raw_energy=raw, compression=comp, **common), \
            dict(raw_energy=raw, compression=comp, blackman_coeff=bc, **f64), 1.0
    if kind == 'spectrogram':
        raw = bool(rng.integers(2))
        return kind, SpectrogramProcessor(raw_energy=raw, energy_floor=floor, **common), \
            dict(kind='spectrogram', raw_energy=raw, energy_floor=floor, blackman_coeff=bc, round_pow2=pow2, **f64), 1.0
    nb = int(rng.choice([20, 23, 26, 40]))
    low = float(rng.choice([0.0, 20.0, 100.0]))
    high = float(rng.choice([0.0, -200.0, -400.0]))
    mel = dict(num_bins=nb, low_freq=low, high_freq=high)
    raw = bool(rng.integers(2))
    if kind == 'fbank':
        use_energy = bool(rng.integers(2))
        use_log = bool(rng.integers(2))
        use_power = bool(rng.integers(2))
        htk = bool(rng.integers(2))
        proc = FilterbankProcessor(use_energy=use_energy, raw_energy=raw, use_log_fbank=use_log, energy_floor=floor,
                                   htk_compat=htk, use_power=use_power, **mel, **common)
        return kind, proc, dict(kind='fbank', use_energy=use_energy, raw_energy=raw, use_log_fbank=use_log,
                                use_power=use_power, energy_floor=floor, htk_compat=htk, blackman_coeff=bc,
                                round_pow2=pow2, **mel, **f64), 1.0
    if kind == 'mfcc':
        nc = int(rng.choice([5, 13, nb]))
        lift = float(rng.choice([0.0, 22.0]))
        use_energy = bool(rng.integers(2))
        htk = bool(rng.integers(2))
        proc = MfccProcessor(num_ceps=nc, cepstral_lifter=lift, use_energy=use_energy, raw_energy=raw,
                             energy_floor=floor, htk_compat=htk, **mel, **common)
        return kind, proc, dict(kind='mfcc', num_ceps=nc, cepstral_lifter=lift, use_energy=use_energy,
                                raw_energy=raw, energy_floor=floor, htk_compat=htk, blackman_coeff=bc,
                                round_pow2=pow2, **mel, **f64), 1.0
    order = int(rng.choice([8, 12, 16]))
    nc = int(rng.choice([5, order + 1, min(13, order + 1)]))
    warp = float(rng.choice([1.0, 1.0, 0.85, 0.93, 1.1, 1.2]))
    opts = dict(lpc_order=order, num_ceps=nc, cepstral_lifter=float(rng.choice([0.0, 22.0])),
                cepstral_scale=float(rng.choice([1.0, 2.0])), use_energy=bool(rng.integers(2)), raw_energy=raw,
                htk_compat=bool(rng.integers(2)), compress_factor=float(rng.choice([1.0 / 3.0, 1.0 / 3.0, 0.5])),
                energy_floor=float(rng.choice([0.0, 0.0, 1.0, 1.0e4])))
    rasta = bool(rng.integers(2))
    proc = PlpProcessor(rasta=rasta, **opts, **mel, **common)
    return kind, proc, dict(use_rasta=rasta, warp=warp, blackman_coeff=bc, round_pow2=pow2, **opts, **mel, **f64), warp


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    worst = {}
    skipped = nulls = 0
    for case in range(n_cases):
        kind, proc, f64, warp = random_case(rng)
        sr = int(proc.sample_rate)
        wave = synth.utterances(100 * seed + case, 1, int(rng.integers(int(0.2 * sr), int(0.8 * sr))), sr)[0]
        what = f'case {case} (seed {seed}): {kind} {proc.get_params()} warp {warp} samples {len(wave)}'
        try:
            got = orc.compute(proc._build_options(), wave, warp)
        except RuntimeError as err:
            # an option error of the oracle (empty mel bin, VTLN limits): the restatement has no error model
            skipped += 1
            continue
        want = spec_f64.plp(wave, **f64) if kind == 'plp' else (
            spec_f64.energy(wave, **f64) if kind == 'energy' else spec_f64.features(wave, **f64))
        if got.shape != want.shape:
            print('FAIL shape', got.shape, want.shape, what)
            return 1
        if not got.size:
            continue
        err = np.abs(got.astype(np.float64) - want)
        # float32 round-off of a log-domain value: relative to the value, plus an absolute term for values near
        # zero (cepstra cross zero; a log of a sum of float32 products carries ~1e-6 of absolute error; the PLP
        # recursion amplifies the round-off of 23 compressed energies)
        atol = {'fbank': 2e-4, 'mfcc': 2e-4, 'spectrogram': 2e-3, 'plp': 5e-4, 'energy': 1e-5}[kind]
        if kind == 'energy':
            bound = 2e-5 * np.abs(want) + 1e-5
        elif kind == 'fbank' and not proc.use_log_fbank:
            bound = 2e-4 * np.abs(want) + 1e-3 * np.abs(want).max() * 1e-3
        else:
            bound = 1e-4 * np.abs(want) + atol
        bad = err > bound
        # Spectral nulls: a bin (or a narrow mel bin) 80 dB below the frame's peak carries the float32 round-off of the
        # whole transform, and its logarithm - and every cepstrum it feeds - moves by 1e-3 ... 1e-2.  A transcription
        # error moves most elements, not a handful: up to 0.5 % of the elements may sit outside the tight bound as long
        # as none is off by more than 0.05 in the log domain (the GPU tests bound such bins in the linear domain).
        if bad.any() and (bad.mean() > 0.005 or err.max() > 0.05 or (kind == 'fbank' and not proc.use_log_fbank)):
            i = np.unravel_index(np.argmax(err - bound), err.shape)
            print('FAIL', what, 'at', i, 'got', got[i], 'want', want[i], 'err', err[i], 'bad', int(bad.sum()), 'of', err.size)
            return 1
        nulls += int(bad.sum())
        worst[kind] = max(worst.get(kind, 0.0), float(err[~bad].max()) if (~bad).any() else 0.0)
    print(f'{n_cases} random option sets (seed {seed}): the C oracle agrees with the float64 restatement; '
          f'worst absolute difference per family {dict((k, float("%.2e" % v)) for k, v in sorted(worst.items()))}; '
          f'{skipped} option errors skipped, {nulls} elements at spectral nulls outside the tight bound')
    return 0


if __name__ == '__main__':
    sys.exit(main())
