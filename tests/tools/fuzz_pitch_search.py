"""Randomised differential test of the two Viterbi searches of the pitch tracker (csrc/kernels_pitch.hip:
section 4, a lane per state, and section 4c, a lane per candidate - the default for large batches since round
5): random tracker options (hence random numbers of states, penalties and lag tables) and ragged utterances,
one wave per utterance forced, and the two forms must return the same BITS; one utterance per case is also
compared with the CPU oracle.  Needs a GPU.

    python tests/tools/fuzz_pitch_search.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as orc  # noqa: E402  (test infrastructure: the checker)
from shennong_amd import Audio, synth  # noqa: E402
from shennong_amd.processor import KaldiPitchProcessor  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    os.environ['SNF_PITCH_TEAM'] = '1'
    states, eligible, oracle_same, oracle_total = {}, 0, 0, 0
    for case in range(n_cases):
        sr = int(rng.choice([8000, 16000]))
        proc = KaldiPitchProcessor(
            sample_rate=sr, frame_shift=float(rng.choice([0.01, 0.0125, 0.02])),
            frame_length=float(rng.choice([0.02, 0.025, 0.04])),
            min_f0=float(rng.choice([40, 50, 65, 80])), max_f0=float(rng.choice([250, 300, 400, 500])),
            soft_min_f0=float(rng.choice([5, 10, 20])),
            penalty_factor=float(rng.choice([0.01, 0.05, 0.1, 0.3, 1.0])),
            lowpass_cutoff=float(rng.choice([800, 1000, 1500])), resample_freq=float(rng.choice([4000, 3600])),
            delta_pitch=float(rng.choice([0.004, 0.005, 0.007, 0.01, 0.02])),
            nccf_ballast=float(rng.choice([0, 7000, 20000])),
            lowpass_filter_width=int(rng.choice([1, 2])), upsample_filter_width=int(rng.choice([3, 5])))
        n_states = 1 + int(np.floor(np.log(proc.max_f0 / proc.min_f0) / np.log(1.0 + proc.delta_pitch)))
        states[n_states] = states.get(n_states, 0) + 1
        eligible += 128 < n_states <= 448
        n = int(rng.integers(1, 10))
        waves = [synth.utterances(100 * case + i, 1, int(rng.integers(int(0.05 * sr), int(4 * sr))), sr)[0]
                 for i in range(n)]
        if rng.integers(4) == 0:
            waves[0] = np.zeros_like(waves[0])          # digital silence: every cost ties
        if rng.integers(4) == 0:
            waves[-1] = (rng.standard_normal(len(waves[-1])) * 300).astype(np.int16)
        audios = [Audio(w, sr, validate=False) for w in waves]
        os.environ['SNF_PITCH_FLAT'] = '0'
        try:
            by_state = [f.data.copy() for f in proc._process_batch(audios)]
        except RuntimeError:
            continue        # (options the tracker refuses: the other form refuses them too, test_parity_gpu.py)
        os.environ['SNF_PITCH_FLAT'] = '1'
        by_candidate = [f.data.copy() for f in proc._process_batch(audios)]
        for i, (a, b) in enumerate(zip(by_state, by_candidate)):
            if not np.array_equal(a, b):
                print('FAIL case', case, 'seed', seed, proc.get_params(), 'states', n_states, 'utterance', i,
                      int((a != b).sum()), 'values differ')
                return 1
        k = int(rng.integers(n))
        want = orc.pitch(proc._options, waves[k])
        if want.size:
            eq = by_candidate[k][:, 1] == want[:, 1]
            oracle_same += int(eq.sum())
            oracle_total += eq.size
    print(f'{n_cases} random option sets: lane-per-candidate == lane-per-state, bit for bit (seed {seed}); '
          f'{eligible} of them run the candidate form (128 < states <= 448); states seen '
          f'{min(states)} ... {max(states)}; {oracle_same} of {oracle_total} frames on the oracle\'s path')
    return 0


if __name__ == '__main__':
    sys.exit(main())
