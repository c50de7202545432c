"""Randomised test of the pipeline's three execution paths against each other (GPU box): the
device-resident pipeline, the stage-by-stage pipeline through the host-pointer entry points, and the
streamed pipeline in small batches must return the same FeaturesCollection for random configurations
and random utterance indexes (in-memory audio at mixed sample rates, speakers, warps).

    python tools/fuzz_pipeline.py [n_cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shennong_amd import Audio, FeaturesCollection, Utterances, _backend, pipeline, synth  # noqa: E402
from shennong_amd.logger import get_logger  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    log = get_logger('fuzz', 'error')
    pinned_cases = [0]
    for case in range(n_cases):
        features = str(rng.choice(['mfcc', 'filterbank', 'plp', 'spectrogram']))
        with_cmvn, with_delta = bool(rng.integers(2)), bool(rng.integers(2))
        with_pitch = 'kaldi' if rng.integers(2) else False
        config = pipeline.get_default_config(features, with_cmvn=with_cmvn, with_delta=with_delta,
                                             with_pitch=with_pitch)
        config[features]['dither'] = 0
        if with_pitch:
            config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
        by_speaker = bool(rng.integers(2))
        if with_cmvn:
            config['cmvn']['by_speaker'] = by_speaker
            config['cmvn']['with_vad'] = False  # (the VAD energy is dithered per batch position)
        rates = [16000] if features == 'spectrogram' or rng.integers(2) else [16000, 8000]
        n = int(rng.integers(1, 9))
        items = []
        for i in range(n):
            sr = int(rng.choice(rates))
            wave = synth.utterances(50 * case + i, 1, int(rng.integers(int(0.2 * sr), int(1.6 * sr))), sr)[0]
            items.append((f'u{i}', Audio(wave, sr), f's{int(rng.integers(3))}'))
        index = Utterances(items)
        warps = None
        if features != 'spectrogram' and rng.integers(2):
            warps = {f's{k}': float(rng.choice([0.9, 1.0, 1.15])) for k in range(3)
                     if any(it[2] == f's{k}' for it in items)}
        _backend.check(_backend.lib().snf_debug_fill_lds(0xFFFFFFFF))
        tag = (f'case {case} (seed {seed}): {features} cmvn={with_cmvn} by_speaker={by_speaker} '
               f'delta={with_delta} pitch={with_pitch} n={n} rates={rates} warps={warps}')
        cfg = pipeline._init_config(config, log=log)
        per_utt = pipeline._init_warps(warps, cfg, index, log) if warps else None
        try:
            a = pipeline._extract_features(cfg, index, per_utt, log)
            b = pipeline._extract_features_by_stage(cfg, index, per_utt, log)
        except Exception:
            print('FAIL (exception)', tag, [(it[0], it[1].nsamples, it[1].sample_rate, it[2]) for it in items])
            raise
        if list(a.keys()) != list(b.keys()) or any(not a[k] == b[k] for k in a):
            bad = [k for k in a if not a[k] == b[k]]
            print('FAIL resident != by stage', tag, bad,
                  [float(np.abs(a[k].data - b[k].data).max()) for k in bad if a[k].shape == b[k].shape])
            return 1
        got = FeaturesCollection()
        batch_s = float(rng.uniform(0.3, 3.0))
        try:
            pipeline.extract_features_streamed(config, index, got.update, warps=warps,
                                               max_batch_duration=batch_s, log=log)
        except Exception:
            print('FAIL (exception, streamed)', tag, batch_s,
                  [(it[0], it[1].nsamples, it[1].sample_rate, it[2]) for it in items])
            raise
        if list(got.keys()) != list(a.keys()) or any(not got[k] == a[k] for k in a):
            print('FAIL streamed != one shot', tag, [k for k in a if not got[k] == a[k]])
            return 1
        if len(rates) == 1 or len(set(it[1].sample_rate for it in items)) == 1:
            # round 6: the same index from ONE page-locked block (Utterances.pin()): one shot, and streamed with
            # 1 - 3 batches in flight (the audio of a batch sent ahead of it)
            pinned = index.pin()
            njobs = int(rng.integers(1, 4))
            try:
                c = pipeline.extract_features(config, pinned, warps=warps, log=log)
                d = FeaturesCollection()
                pipeline.extract_features_streamed(config, pinned, d.update, warps=warps, njobs=njobs,
                                                   max_batch_duration=batch_s, log=log)
            except Exception:
                print('FAIL (exception, pinned)', tag, batch_s, njobs)
                raise
            for name, other in (('pinned one shot', c), ('pinned streamed njobs %d' % njobs, d)):
                same = list(other.keys()) == list(a.keys()) and all(
                    np.array_equal(other[k].data, a[k].data) and np.array_equal(other[k].times, a[k].times)
                    and other[k].properties['pipeline'] == a[k].properties['pipeline'] for k in a)
                if not same:
                    print('FAIL %s != one shot' % name, tag)
                    return 1
            pinned_cases[0] += 1
    print(f'{n_cases} random pipelines: resident == by stage == streamed (seed {seed}); '
          f'{pinned_cases[0]} of them also from a pinned index, one shot and streamed with 1-3 batches in flight')
    return 0


if __name__ == '__main__':
    sys.exit(main())
