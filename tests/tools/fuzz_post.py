"""Randomised differential test of the pitch tracker and the post-processors (pitch post-processing,
delta, VAD, CMVN, sliding-window CMVN) against the CPU oracle; companion of tools/fuzz_parity.py.

    python tools/fuzz_post.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc  # noqa: E402  (test infrastructure: the checker)
from shennong_amd import Audio, Features, _backend, synth  # noqa: E402
from shennong_amd.postprocessor import (  # noqa: E402
    CmvnPostProcessor, DeltaPostProcessor, SlidingWindowCmvnPostProcessor, VadPostProcessor)
from shennong_amd.processor import KaldiPitchPostProcessor, KaldiPitchProcessor  # noqa: E402


def feats_of(rng, cols, lo=1, hi=400):
    n = int(rng.integers(lo, hi))
    data = (rng.standard_normal((n, cols)) * rng.uniform(0.5, 20) + rng.uniform(-30, 30)).astype(np.float32)
    return Features(data, np.arange(n, dtype=np.float64) * 0.01)


def fail(*what):
    print('FAIL', *what)
    return 1


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    counts = {}
    for case in range(n_cases):
        kind = str(rng.choice(['pitch', 'pitch_post', 'delta', 'vad', 'cmvn', 'sliding']))
        counts[kind] = counts.get(kind, 0) + 1
        _backend.check(_backend.lib().snf_debug_fill_lds(0xFFFFFFFF))
        tag = f'case {case} (seed {seed}) {kind}'
        if kind == 'pitch':
            sr = int(rng.choice([8000, 16000]))
            proc = KaldiPitchProcessor(
                sample_rate=sr, frame_shift=float(rng.choice([0.01, 0.0125, 0.02])),
                frame_length=float(rng.choice([0.02, 0.025, 0.04])),
                min_f0=float(rng.choice([40, 50, 80])), max_f0=float(rng.choice([300, 400, 500])),
                soft_min_f0=float(rng.choice([5, 10, 20])), penalty_factor=float(rng.choice([0.05, 0.1, 0.3])),
                lowpass_cutoff=float(rng.choice([800, 1000, 1500])),
                resample_freq=float(rng.choice([4000, 4000, 3600])),
                delta_pitch=float(rng.choice([0.005, 0.01])), nccf_ballast=float(rng.choice([0, 7000, 20000])),
                lowpass_filter_width=int(rng.choice([1, 2])), upsample_filter_width=int(rng.choice([3, 5])))
            n = int(rng.integers(1, 8))
            waves = [synth.utterances(100 * case + i, 1, int(rng.integers(int(0.01 * sr), int(2.5 * sr))), sr)[0]
                     for i in range(n)]
            try:
                outs = proc._process_batch([Audio(w, sr) for w in waves])
            except RuntimeError as err:
                try:
                    for w in waves:
                        orc.pitch(proc._options, w)
                except RuntimeError:
                    continue
                return fail(tag, proc.get_params(), 'device raised, oracle did not:', err)
            same = total = 0
            for w, o in zip(waves, outs):
                want = orc.pitch(proc._options, w)
                if o.data.shape != want.shape and (o.data.size or want.size):
                    return fail(tag, proc.get_params(), 'shape', o.data.shape, want.shape, len(w))
                if not want.size:
                    continue
                eq = o.data[:, 1] == want[:, 1]
                same += int(eq.sum())
                total += eq.size
                if not np.allclose(o.data[eq, 0], want[eq, 0], rtol=2e-3, atol=5e-4):
                    return fail(tag, proc.get_params(), 'nccf', float(np.abs(o.data[eq, 0] - want[eq, 0]).max()))
                if np.max(np.abs(o.data[:, 1] / want[:, 1] - 1)) > 0.3:
                    return fail(tag, proc.get_params(), 'pitch far off')
            if total > 200 and same < 0.95 * total:
                return fail(tag, proc.get_params(), f'only {same}/{total} frames on the same Viterbi path')
        elif kind == 'pitch_post':
            flags = [bool(rng.integers(2)) for _ in range(4)]
            if not any(flags):
                flags[int(rng.integers(4))] = True
            proc = KaldiPitchPostProcessor(
                pitch_scale=float(rng.uniform(0.5, 3)), pov_scale=float(rng.uniform(0.5, 3)),
                pov_offset=float(rng.uniform(-1, 1)), delta_pitch_scale=float(rng.uniform(1, 20)),
                delta_pitch_noise_stddev=0, normalization_left_context=int(rng.integers(0, 120)),
                normalization_right_context=int(rng.integers(0, 120)), delta_window=int(rng.integers(1, 5)),
                add_pov_feature=flags[0], add_normalized_log_pitch=flags[1], add_delta_pitch=flags[2],
                add_raw_log_pitch=flags[3])
            raws = []
            for _ in range(int(rng.integers(1, 6))):
                n = int(rng.integers(1, 500))
                raw = np.stack([rng.uniform(-1, 1, n), rng.uniform(50, 400, n)], axis=1).astype(np.float32)
                raws.append(Features(raw, np.arange(n) * 0.01, properties={'pitch': {}, 'pipeline': [{'name': 'pitch', 'columns': [0, 1]}]}))
            for raw, got in zip(raws, proc._process_batch(raws)):
                want = orc.process_pitch(proc._options, raw.data)
                if not np.allclose(got.data, want, rtol=2e-4, atol=2e-5):
                    return fail(tag, proc.get_params(), float(np.abs(got.data - want).max()))
        elif kind == 'delta':
            proc = DeltaPostProcessor(order=int(rng.integers(0, 4)), window=int(rng.integers(1, 5)))
            cols = int(rng.integers(1, 80))
            mats = [feats_of(rng, cols) for _ in range(int(rng.integers(1, 7)))]
            for m, got in zip(mats, proc._process_batch(mats)):
                want = orc.deltas(m.data, proc.order, proc.window)
                if not np.allclose(got.data, want, rtol=1e-5, atol=1e-4):
                    return fail(tag, proc.get_params(), cols, float(np.abs(got.data - want).max()))
        elif kind == 'vad':
            proc = VadPostProcessor(
                energy_threshold=float(rng.uniform(-5, 10)), energy_mean_scale=float(rng.uniform(0, 1.5)),
                frames_context=int(rng.integers(0, 6)), proportion_threshold=float(rng.uniform(0.05, 0.95)))
            mats = [feats_of(rng, int(rng.integers(1, 14))) for _ in range(int(rng.integers(1, 7)))]
            mats = [Features(m.data[:, :mats[0].ndims] if m.ndims >= mats[0].ndims else
                             np.resize(m.data, (m.nframes, mats[0].ndims)), m.times) for m in mats]
            for m, got in zip(mats, proc._process_batch(mats)):
                want = orc.vad_energy(m.data, proc.energy_threshold, proc.energy_mean_scale,
                                      proc.frames_context, proc.proportion_threshold)
                if not np.array_equal(got.data.reshape(-1).astype(np.float32), want):
                    return fail(tag, proc.get_params(), int((got.data.reshape(-1) != want).sum()), 'decisions differ')
        elif kind == 'cmvn':
            cols = int(rng.integers(1, 300))
            mats = [feats_of(rng, cols, lo=2) for _ in range(int(rng.integers(1, 6)))]
            proc = CmvnPostProcessor(cols)
            for m in mats:
                proc.accumulate(m)
            stats = np.zeros((2, cols + 1))
            for m in mats:
                orc.cmvn_accumulate(m.data, stats=stats)
            if not np.allclose(proc.stats, stats, rtol=1e-12, atol=1e-9):
                return fail(tag, cols, 'statistics', float(np.abs(proc.stats - stats).max()))
            norm_vars, reverse = bool(rng.integers(2)), bool(rng.integers(4) == 0)
            got = proc.process(mats[0], norm_vars=norm_vars, reverse=reverse)
            want = orc.cmvn_apply(mats[0].data, stats, norm_vars=norm_vars, reverse=reverse)
            if not np.allclose(got.data, want, rtol=1e-5, atol=1e-5):
                return fail(tag, cols, norm_vars, reverse, float(np.abs(got.data - want).max()))
        else:
            proc = SlidingWindowCmvnPostProcessor(
                center=bool(rng.integers(2)), cmn_window=int(rng.integers(2, 300)),
                min_window=int(rng.integers(1, 100)), normalize_variance=bool(rng.integers(2)))
            proc.min_window = min(proc.min_window, proc.cmn_window)
            cols = int(rng.integers(1, 60))
            mats = [feats_of(rng, cols) for _ in range(int(rng.integers(1, 6)))]
            for m, got in zip(mats, proc._process_batch(mats)):
                want = orc.sliding_cmn(m.data, center=proc.center, cmn_window=proc.cmn_window,
                                       min_window=proc.min_window, normalize_variance=proc.normalize_variance)
                if not np.allclose(got.data, want, rtol=2e-4, atol=2e-4):
                    return fail(tag, proc.get_params(), cols, m.nframes, float(np.abs(got.data - want).max()))
    print(f'{n_cases} random cases agree with the oracle (seed {seed}): {counts}')
    return 0


if __name__ == '__main__':
    sys.exit(main())
