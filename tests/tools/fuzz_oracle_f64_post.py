"""Randomised comparison of the C oracle with the float64 restatement for the post-processing families: delta,
energy VAD, CMVN (statistics, apply, reverse, weights), sliding-window CMVN and the pitch post-processing (noise term at 0) - CPU
only.  Inputs are random float32 matrices shaped like features (and synthetic (NCCF, pitch) tracks), so every
difference is float32 round-off of the oracle or a transcription error in one of the two statements.

    python tests/tools/fuzz_oracle_f64_post.py [n_cases] [seed]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc, spec_f64  # noqa: E402
from shennong_amd.processor import KaldiPitchPostProcessor  # noqa: E402


def close(got, want, rtol, atol):
    err = np.abs(np.asarray(got, np.float64) - want)
    return bool((err <= rtol * np.abs(want) + atol).all()), float(err.max()) if err.size else 0.0


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    worst, count = {}, {}
    for case in range(n_cases):
        family = str(rng.choice(['delta', 'cmvn', 'sliding', 'pitch_post', 'vad']))
        n, d = int(rng.integers(1, 400)), int(rng.integers(1, 45))
        scale = float(rng.choice([0.1, 1.0, 20.0]))
        x = (rng.standard_normal((n, d)) * scale + rng.standard_normal(d) * scale).astype(np.float32)
        what = f'case {case} (seed {seed}): {family} [{n}, {d}]'
        if family == 'delta':
            order, window = int(rng.integers(1, 4)), int(rng.integers(1, 5))
            got, want = orc.deltas(x, order, window), spec_f64.delta(x, order, window)
            ok, w = close(got, want, 1e-5, 1e-5 * scale)
            what += f' order {order} window {window}'
        elif family == 'cmvn':
            weights = rng.random(n).astype(np.float32) if rng.integers(2) else None
            norm_vars, reverse = bool(rng.integers(2)), bool(rng.integers(2))
            stats = orc.cmvn_accumulate(x, weights)
            want_stats = spec_f64.cmvn_stats(x, weights)
            ok, w = close(stats, want_stats, 1e-6, 1e-6 * scale * scale * n)
            if ok and stats[0, -1] >= 1.0 and n > 1:   # (Kaldi refuses statistics of less than one frame)
                got = orc.cmvn_apply(x, stats, norm_vars, reverse)
                want = spec_f64.cmvn_apply(x, want_stats, norm_vars, reverse)
                ok, w = close(got, want, 1e-4, 1e-4 * max(scale, 1.0))
            what += f' weights {weights is not None} norm_vars {norm_vars} reverse {reverse}'
        elif family == 'sliding':
            center, nv = bool(rng.integers(2)), bool(rng.integers(2))
            cmn_window, min_window = int(rng.choice([5, 30, 100, 600])), int(rng.choice([1, 10, 100]))
            min_window = min(min_window, cmn_window)
            got = orc.sliding_cmn(x, center, cmn_window, min_window, nv)
            want = spec_f64.sliding_cmvn(x, center, cmn_window, min_window, nv)
            ok, w = close(got, want, 1e-4, 2e-4 * max(scale, 1.0))
            what += f' center {center} window {cmn_window} min {min_window} norm_vars {nv}'
        elif family == 'vad':
            kw = dict(energy_threshold=float(rng.choice([5.0, 0.0, -1.0, 2.0])) * scale,
                      energy_mean_scale=float(rng.choice([0.5, 0.0, 1.0])), frames_context=int(rng.choice([0, 1, 2, 5])),
                      proportion_threshold=float(rng.choice([0.6, 0.5, 0.25, 1.0])))
            got = orc.vad_energy(x, **kw)
            want, margin = spec_f64.vad_energy(x, **kw)
            # (a frame whose energy sits within float32 round-off of the threshold may fall on either side)
            w = float(np.abs(got - want).max()) if margin > 1e-5 * max(scale, 1.0) else 0.0
            ok = w == 0.0
            what += f' {kw}'
        else:
            nccf = np.clip(rng.standard_normal(n) * 0.5, -1.0, 1.0)
            pitch = np.exp(rng.uniform(np.log(50.0), np.log(400.0), n))
            raw = np.stack([nccf, pitch], axis=1).astype(np.float32)
            flags = dict(add_pov_feature=bool(rng.integers(2)), add_normalized_log_pitch=bool(rng.integers(2)),
                         add_delta_pitch=bool(rng.integers(2)), add_raw_log_pitch=bool(rng.integers(2)))
            if not any(flags.values()):
                flags['add_raw_log_pitch'] = True
            kw = dict(pitch_scale=float(rng.choice([1.0, 2.0])), pov_scale=float(rng.choice([0.5, 2.0])),
                      pov_offset=float(rng.choice([0.0, 0.3])), delta_pitch_scale=float(rng.choice([3.0, 10.0])),
                      normalization_left_context=int(rng.choice([3, 75])),
                      normalization_right_context=int(rng.choice([0, 75])), delta_window=int(rng.integers(1, 5)))
            got = orc.process_pitch(KaldiPitchPostProcessor(delta_pitch_noise_stddev=0, **kw, **flags)._options, raw)
            want = spec_f64.process_pitch(
                raw, pitch_scale=kw['pitch_scale'], pov_scale=kw['pov_scale'], pov_offset=kw['pov_offset'],
                delta_pitch_scale=kw['delta_pitch_scale'], left_context=kw['normalization_left_context'],
                right_context=kw['normalization_right_context'], delta_window=kw['delta_window'], **flags)
            if got.shape != want.shape:
                print('FAIL shape', got.shape, want.shape, what, kw, flags)
                return 1
            ok, w = close(got, want, 1e-4, 2e-5)
            what += f' {kw} {flags}'
        if not ok:
            print('FAIL', what, 'worst', w)
            return 1
        worst[family] = max(worst.get(family, 0.0), w / (scale if family not in ('pitch_post', 'vad') else 1.0))
        count[family] = count.get(family, 0) + 1
    print(f'{n_cases} random cases (seed {seed}): the C oracle agrees with the float64 restatement; cases {count}; '
          f'worst difference per family, relative to the scale of the data '
          f'{dict((k, float("%.2e" % v)) for k, v in sorted(worst.items()))}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
