"""Random option sets of the Kaldi pitch tracker: the C oracle (float32, the summation orders the GPU reproduces bit for
bit) against the float64 restatement with a FULL-search Viterbi (oracle/spec_f64.pitch) - CPU only.  Reported per
case: the largest difference of the resampled NCCF, the frames whose Viterbi state differs (paths that tie within float32
round-off may part for a few frames) and the largest distance between the two states.

    python tests/tools/fuzz_oracle_f64_pitch.py [n_cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc, spec_f64  # noqa: E402
from shennong_amd import _abi, synth  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    worst, differ, total, step = 0.0, 0, 0, 0
    for case in range(n_cases):
        kw = dict(frame_shift_ms=float(rng.choice([10.0, 10.0, 20.0])), frame_length_ms=float(rng.choice([25.0, 25.0, 40.0])),
                  min_f0=float(rng.choice([50.0, 50.0, 70.0])), max_f0=float(rng.choice([400.0, 400.0, 300.0])),
                  soft_min_f0=float(rng.choice([10.0, 10.0, 30.0])), penalty_factor=float(rng.choice([0.1, 0.1, 0.3])),
                  delta_pitch=float(rng.choice([0.005, 0.005, 0.01])), nccf_ballast=float(rng.choice([7000.0, 7000.0, 1000.0])),
                  upsample_filter_width=int(rng.choice([5, 5, 7])), lowpass_cutoff=float(rng.choice([1000.0, 1000.0, 800.0, 900.0])),
                  samp_freq=int(rng.choice([16000, 16000, 8000, 22050])), resample_freq=int(rng.choice([4000, 4000, 3200])),
                  lowpass_filter_width=int(rng.choice([1, 1, 2])))
        if 2 * kw['lowpass_cutoff'] > kw['resample_freq']:
            kw['lowpass_cutoff'] = kw['resample_freq'] / 4.0
        po = _abi.default_pitch_options()
        for k, v in kw.items():
            setattr(po, k, v)
        n = int(rng.integers(8000, 40000))
        wave = synth.utterances(5000 + 100 * seed + case, 1, n * kw['samp_freq'] // 16000, kw['samp_freq'])[0]
        ref = spec_f64.pitch(wave, **kw)
        out, _, res, _, states = orc.pitch_debug(po, wave)
        what = f'case {case} (seed {seed}): {kw} samples {n}'
        if res.shape != ref['nccf'].shape or states.shape != ref['states'].shape:
            print('FAIL shape', res.shape, ref['nccf'].shape, what)
            return 1
        d = float(np.abs(res - ref['nccf']).max()) if res.size else 0.0
        diff = int((states != ref['states']).sum())
        st = int(np.abs(states - ref['states']).max()) if states.size else 0
        same = states == ref['states']
        ok = d < 2e-6 and diff <= max(2, states.shape[0] // 20) and st <= 3
        if ok and same.any():
            ok = bool(np.allclose(out[same, 1], ref['out'][same, 1], rtol=1e-6) and
                      np.allclose(out[same, 0], ref['out'][same, 0], atol=5e-5))
        if not ok:
            print('FAIL', what, 'nccf', d, 'states differ', diff, 'of', states.shape[0], 'step', st)
            return 1
        worst, differ, total, step = max(worst, d), differ + diff, total + states.shape[0], max(step, st)
    print(f'{n_cases} random pitch option sets (seed {seed}): largest |NCCF oracle - float64| {worst:.2e}; '
          f'{differ} of {total} frames on another Viterbi state, at most {step} states apart; pitch and POV equal '
          f'(1e-6 relative / 5e-5 absolute) wherever the states agree')
    return 0


if __name__ == '__main__':
    sys.exit(main())
