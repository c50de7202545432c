"""BASELINE.json configurations 4 and 5 at the scale one GPU of the node sees.

Config 4: PlpProcessor 13 + KaldiPitchProcessor (+ post-processing) on 100 000 utterances of 1-6 s
sharded over 8 GPUs -> ONE rank's shard of 12 500 ragged utterances (12 h of audio) goes through the
batched launches; a seeded sample of them is compared with the CPU oracle (pitch: every frame
bit-identical; PLP: the parity tolerance) and with the same utterance processed alone (bit-identical:
results do not depend on what else is in the batch).

Config 5: fbank-40 + pitch + delta + CMVN by speaker (VAD-weighted statistics: the reference's default,
pipeline.py:584-596) streamed over a corpus that does not fit in one batch -> (a) 10 h of unique synthetic
audio through pipeline.extract_features_streamed in 30-minute batches, (b) ONE GPU's share of the 1 000 h
corpus: 125 h = 150 000 utterances of 3 s (400 seeded waves reused round robin so that host synthesis is
not the cost; names, speakers and statistics are per utterance), 1 000 speakers, default batches.  Every
utterance arrives exactly once and finite, a sample equals the one-shot pipeline bit for bit.
"""

import os
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import pytest

from conftest import assert_close
from oracle import oracle as orc
from shennong_amd import Audio, Utterances, pipeline, synth
from shennong_amd.logger import get_logger
from shennong_amd.processor import KaldiPitchPostProcessor, KaldiPitchProcessor, PlpProcessor

pytestmark = pytest.mark.gpu


def _ragged(args):
    first, count = args
    rng = np.random.default_rng(20260927 + first)
    return [synth.utterances(first + i, 1, int(rng.integers(16000, 96001)))[0] for i in range(count)]


def _uniform(args):
    return synth.utterances(*args)


def _pool_map(fn, jobs):
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    if workers == 1:
        return [fn(j) for j in jobs]
    with ProcessPoolExecutor(workers) as pool:
        return list(pool.map(fn, jobs))


def _oracle_pipeline_check(config, index, kept, wave_of, what):
    """the composed pipeline against the ORACLE, not against itself.  For the utterances in
    `kept` ({name: Features [frames, 123]}, all the utterances of their speakers) the expected columns are built
    from oracle calls only (reference shennong/pipeline.py:570-644, postprocessor/cmvn.py:180-282):

        fbank-40 (orc.compute) -> VAD weights (orc.vad_energy of the oracle's log-energy) -> statistics per
        speaker (orc.cmvn_accumulate over ALL the speaker's utterances) -> orc.cmvn_apply -> orc.deltas ->
        | orc.pitch -> orc.process_pitch | joined with the paste-feats tolerance of 2 frames

    The reference dithers the energy the VAD looks at (EnergyProcessor's default dither = 1, drawn from a
    generator the oracle cannot replay), so a handful of frames next to the threshold may be weighted the other
    way: (a) the statistics the pipeline recorded in the properties agree with the oracle's to that extent -
    frame counts within 0.2 %, sums within 2e-3 relative -, (b) the final columns agree with the all-oracle chain
    at 5e-3 absolute (normalised units), and (c) with the oracle chain run on the pipeline's OWN recorded
    statistics - which removes the dither from the comparison - at the parity tolerance, every column family."""
    from shennong_amd.processor import EnergyProcessor, FilterbankProcessor
    fopts = FilterbankProcessor(**config['filterbank'])._build_options()
    eopts = EnergyProcessor(dither=0)._build_options()
    pproc = KaldiPitchProcessor(**{k: v for k, v in config['pitch'].items()
                                   if k not in ('processor', 'postprocessing')})
    post = KaldiPitchPostProcessor(**config['pitch']['postprocessing'])
    per_wave = {}

    def oracle_of(wave_id):
        if wave_id not in per_wave:
            wave = wave_of(wave_id)
            fbank = orc.compute(fopts, wave)
            weights = orc.vad_energy(orc.compute(eopts, wave), **config['cmvn']['vad'])
            pitch = orc.process_pitch(post._options, orc.pitch(pproc._options, wave))
            per_wave[wave_id] = (fbank, weights, pitch)
        return per_wave[wave_id]
    by_speaker = {}
    for name in kept:
        by_speaker.setdefault(index.by_name()[name].speaker, []).append(name)
    checked = 0
    for speaker, names in by_speaker.items():
        assert len(names) == sum(1 for u in index if u.speaker == speaker)   # (all of the speaker's utterances)
        want_stats = np.zeros((2, 41))
        for name in sorted(names):
            fbank, weights, _ = oracle_of(index.by_name()[name]._wave_id)
            orc.cmvn_accumulate(fbank, weights=weights, stats=want_stats)
        got_stats = kept[names[0]].properties['cmvn']['stats']
        assert got_stats.shape == (2, 41)
        assert abs(got_stats[0, -1] - want_stats[0, -1]) <= 2e-3 * want_stats[0, -1], (what, speaker)
        np.testing.assert_allclose(got_stats, want_stats, rtol=2e-3, err_msg=f'{what}: statistics of {speaker}')
        for name in names:
            fbank, _, pitch = oracle_of(index.by_name()[name]._wave_id)
            got = kept[name].data
            rows = min(fbank.shape[0], pitch.shape[0])
            assert abs(fbank.shape[0] - pitch.shape[0]) <= 2 and got.shape == (rows, 123)
            for stats, atol in ((want_stats, 5e-3), (got_stats, None)):
                normed = orc.cmvn_apply(fbank, stats)
                want = np.hstack((orc.deltas(normed, 2, 2)[:rows], pitch[:rows]))
                assert want.shape == got.shape and want.dtype == np.float32
                if atol is not None:   # (b) all-oracle: within the VAD's dither
                    np.testing.assert_allclose(got[:, :120], want[:, :120], rtol=1e-4, atol=atol,
                                               err_msg=f'{what}: {name} vs the all-oracle chain')
                else:                  # (c) the oracle chain on the recorded statistics: the parity tolerance
                    assert_close(got[:, :40], want[:, :40], rtol=1e-4, atol=1e-4, what=f'{what} cmvn(fbank) {name}')
                    assert_close(got[:, 40:120], want[:, 40:120], rtol=1e-4, atol=1e-4,
                                 what=f'{what} delta columns {name}')
                    assert_close(got[:, 120:], want[:, 120:], rtol=1e-4, family='pitch_post',
                                 what=f'{what} pitch columns {name}')
            checked += 1
    return checked


def test_config4_one_shard(gpu):
    n = 12500
    waves = [w for part in _pool_map(_ragged, [(i, min(500, n - i)) for i in range(0, n, 500)]) for w in part]
    seconds = sum(w.shape[0] for w in waves) / 16000.0
    assert 11.0 * 3600 < seconds < 13.5 * 3600
    audios = [Audio(w, 16000, validate=False) for w in waves]
    plp, pitch, post = PlpProcessor(dither=0), KaldiPitchProcessor(), KaldiPitchPostProcessor(
        delta_pitch_noise_stddev=0)
    f_plp = plp._process_batch(audios)
    f_pitch = pitch._process_batch(audios)
    f_post = post._process_batch(f_pitch)
    assert len(f_plp) == len(f_pitch) == len(f_post) == n
    frames = sum(f.nframes for f in f_pitch)
    assert frames > 4.0e6
    for feats, dims in ((f_plp, 13), (f_pitch, 2), (f_post, 3)):
        assert all(f.ndims == dims for f in feats)
        assert all(np.isfinite(f.data).all() for f in feats[::97])
    sample = [int(i) for i in np.random.default_rng(4).choice(n, size=16, replace=False)]
    sample += [int(np.argmax([w.shape[0] for w in waves])), int(np.argmin([w.shape[0] for w in waves]))]
    for i in sample:
        want = orc.pitch(pitch._options, waves[i])
        np.testing.assert_array_equal(f_pitch[i].data, want, err_msg=f'pitch of utterance {i}')
        assert_close(f_post[i].data, orc.process_pitch(post._options, want), rtol=1e-4, family='pitch_post',
                     what=f'pitch post {i}')
        assert_close(f_plp[i].data, orc.compute(plp._build_options(), waves[i]), rtol=1e-4,
                     what=f'plp {i}')
        alone = plp.process(audios[i])
        np.testing.assert_array_equal(f_plp[i].data, alone.data)
        np.testing.assert_array_equal(f_pitch[i].data, pitch.process(audios[i]).data)


def test_config5_ten_hours_streamed(gpu):
    n, nsamples, speakers = 12000, 48000, 200      # 10 h of 3 s utterances
    waves = np.concatenate(_pool_map(_uniform, [(i, 500, nsamples) for i in range(0, n, 500)]))
    index = Utterances([(f'u{i:05d}', Audio(waves[i], 16000, validate=False), f's{i % speakers:03d}')
                        for i in range(n)])
    config = pipeline.get_default_config('filterbank', with_pitch='kaldi', with_cmvn=True, with_delta=True)
    config['filterbank']['num_bins'] = 40
    config['filterbank']['dither'] = 0
    config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
    config['cmvn']['by_speaker'] = True
    config['cmvn']['with_vad'] = True
    quiet = get_logger('test', 'error')
    seen, kept = [], {}
    keep = {f'u{i:05d}' for i in range(0, n, 997)}

    def sink(feats):
        for name, f in feats.items():
            seen.append(name)
            assert f.shape == (298, 123) and f.dtype == np.float32   # 40 x 3 + pitch 3
            if name in keep:
                kept[name] = f
    written = pipeline.extract_features_streamed(config, index, sink, max_batch_duration=1800.0, log=quiet)
    assert written == n and sorted(seen) == sorted(u.name for u in index) and len(set(seen)) == n
    assert all(np.isfinite(f.data).all() for f in kept.values())
    # the speakers of the kept utterances, extracted in one shot: the statistics of a speaker come from
    # the same utterances, so the features must be the same bits
    wanted = {index.by_name()[k].speaker for k in list(keep)[:3]}
    subset = Utterances([(u.name, u.load_audio(), u.speaker) for u in index if u.speaker in wanted])
    whole = pipeline.extract_features(config, subset, log=quiet)
    checked = 0
    for name, f in kept.items():
        if name in whole:
            assert f == whole[name], name
            checked += 1
    assert checked >= 3


def test_config5_per_gpu_share_streamed(gpu):
    """125 h (one GPU's eighth of BASELINE config 5) with the reference's default CMVN: by speaker,
    VAD-weighted (reference shennong/pipeline.py:525-603)"""
    import resource
    from shennong_amd import _backend
    n, unique, speakers, nsamples = 150000, 400, 1000, 48000
    waves = np.concatenate(_pool_map(_uniform, [(i, 100, nsamples) for i in range(0, unique, 100)]))
    index = Utterances([(f'u{i:06d}', Audio(waves[i % unique], 16000, validate=False), f's{i % speakers:04d}')
                        for i in range(n)])
    for i, utt in enumerate(index):
        utt._wave_id = i % unique
    assert abs(sum(u.duration for u in index) / 3600.0 - 125.0) < 1e-6
    config = pipeline.get_default_config('filterbank', with_pitch='kaldi', with_cmvn=True, with_delta=True)
    config['filterbank']['num_bins'] = 40
    config['filterbank']['dither'] = 0
    config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
    config['cmvn']['by_speaker'] = True
    assert config['cmvn']['with_vad'] is True   # the reference's default
    quiet = get_logger('test', 'error')
    wanted = ('s0000', 's0417', 's0999')
    count = {'utts': 0, 'batches': 0, 'peak_device': 0}
    seen = np.zeros(n, dtype=np.int32)
    kept = {}

    def sink(feats):
        count['batches'] += 1
        blocks = {}
        for name, f in feats.items():
            seen[int(name[1:])] += 1
            assert f.shape == (298, 123) and f.dtype == np.float32
            base = f.data.base if f.data.base is not None else f.data
            blocks[id(base)] = base
            if f.properties['speaker'] in wanted:
                kept[name] = f.copy()
        for block in blocks.values():   # (the utterances of a batch are views of one downloaded array)
            block = np.asarray(block)
            assert np.isfinite(block.min()) and np.isfinite(block.max())
        count['utts'] += len(feats)
        free, total = _backend.mem_info()
        count['peak_device'] = max(count['peak_device'], total - free)
    written = pipeline.extract_features_streamed(config, index, sink, log=quiet)
    assert written == n == count['utts'] and (seen == 1).all()
    assert count['batches'] >= 125.0 * 3600 / pipeline.default_batch_duration(1) - 1
    assert count['peak_device'] < 200 << 30
    assert resource.getrusage(resource.RUSAGE_SELF).ru_maxrss < 64 << 20   # (KiB: 64 GiB)
    # the wanted speakers extracted in one shot: same utterances per speaker, same order -> same bits
    assert len(kept) == 3 * n // speakers
    subset = Utterances([(u.name, u.load_audio(), u.speaker) for u in index if u.speaker in wanted])
    whole = pipeline.extract_features(config, subset, log=quiet)
    assert sorted(whole.keys()) == sorted(kept)
    for name, f in kept.items():
        assert f == whole[name], name
    # ... and against the oracle: the 450 utterances of the three speakers, every column family
    assert _oracle_pipeline_check(config, index, kept, lambda k: waves[k], 'config 5, 125 h streamed') == 450
    # the same corpus from ONE page-locked block (Utterances.pin(), two batches in flight): the same bits
    pinned_kept = {}

    def pinned_sink(feats):
        for name, f in feats.items():
            if name in kept:
                pinned_kept[name] = f.copy()
    assert pipeline.extract_features_streamed(config, index.pin(), pinned_sink, njobs=2, log=quiet) == n
    assert sorted(pinned_kept) == sorted(kept)
    for name, f in kept.items():
        assert np.array_equal(pinned_kept[name].data, f.data), name
