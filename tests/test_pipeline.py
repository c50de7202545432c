"""Pipeline orchestration: the reference's test/test_pipeline.py re-expressed
against shennong_amd.pipeline.  Configuration handling runs on CPU; extraction needs the GPU."""

import os

import numpy as np
import pytest
import yaml

from conftest import GOLDEN
from shennong_amd import Audio, Utterances, pipeline
from shennong_amd.logger import get_logger

WAV = os.path.join(GOLDEN, 'test.wav')
WAV_8K = os.path.join(GOLDEN, 'test.8k.wav')


@pytest.fixture(scope='module')
def utterances():
    return Utterances([('utt1', WAV, 'speaker1'), ('utt2', WAV, 'speaker2')])


def equal_dict(d1, d2):
    assert 'htk_compat' not in d1.keys()
    assert 'sample_rate' not in d1.keys()
    if not d1.keys() == d2.keys():
        return False
    for k, v in d1.items():
        if isinstance(v, str):
            if not v == d2[k]:
                return False
        elif isinstance(v, dict):
            if not equal_dict(v, d2[k]):
                return False
        else:
            if v != pytest.approx(d2[k]):
                return False
    return True


@pytest.mark.parametrize('features, with_pitch', [
    (f, p) for f in ('mfcc', 'plp', 'filterbank', 'spectrogram') for p in (False, 'kaldi')])
def test_config_good(features, with_pitch):
    """reference test_pipeline.py:48-70"""
    c1 = pipeline.get_default_config(features, to_yaml=False, with_pitch=with_pitch,
                                     with_cmvn=True, with_delta=True)
    c2 = pipeline.get_default_config(features, to_yaml=True, yaml_commented=False,
                                     with_pitch=with_pitch, with_cmvn=True, with_delta=True)
    c3 = pipeline.get_default_config(features, to_yaml=True, yaml_commented=True,
                                     with_pitch=with_pitch, with_cmvn=True, with_delta=True)
    assert features in c1.keys()
    assert '#' not in c2
    assert '#' in c3
    assert equal_dict(c1, yaml.load(c2, Loader=yaml.FullLoader))
    assert equal_dict(c1, yaml.load(c3, Loader=yaml.FullLoader))


def test_config_keys():
    """reference pipeline.py:18-30 doctest"""
    config = pipeline.get_default_config('mfcc', with_pitch='kaldi', with_cmvn=True, with_delta=True)
    assert list(config.keys()) == ['mfcc', 'pitch', 'cmvn', 'delta']
    assert list(pipeline.get_default_config('mfcc', with_pitch='kaldi').keys()) == ['mfcc', 'pitch']
    assert config['cmvn'] == {'by_speaker': True, 'with_vad': True, 'vad': {
        'energy_threshold': 5.0, 'energy_mean_scale': 0.5, 'frames_context': 0,
        'proportion_threshold': pytest.approx(0.6)}}
    assert config['delta'] == {'order': 2, 'window': 2}
    assert config['pitch']['processor'] == 'kaldi'
    assert 'frame_shift' not in config['pitch'] and 'sample_rate' not in config['pitch']
    assert len(config['pitch']['postprocessing']) == 13
    assert 'sample_rate' not in config['mfcc'] and 'htk_compat' not in config['mfcc']
    assert pipeline.valid_features() == ['spectrogram', 'filterbank', 'mfcc', 'plp']


@pytest.mark.parametrize('kind', ['dict', 'file', 'str'])
def test_config_format(capsys, tmpdir, kind):
    """reference test_pipeline.py:73-98"""
    config = pipeline.get_default_config(
        'mfcc', with_pitch='kaldi', with_cmvn=True, with_delta=True, to_yaml=kind != 'dict')
    if kind == 'file':
        tempfile = str(tmpdir.join('foo'))
        open(tempfile, 'w').write(config)
        config = tempfile
    if kind == 'str':
        with pytest.raises(ValueError) as err:
            pipeline._init_config('a:\nb\n')
        assert 'error in configuration' in str(err.value)
    parsed = pipeline._init_config(config, log=get_logger('pipeline', level='info'))
    output = capsys.readouterr().err
    for word in ('mfcc', 'pitch', 'cmvn', 'delta'):
        assert word in output
        assert word in parsed


def test_config_bad(utterances):
    """reference test_pipeline.py:101-160 (the backend-specific refusals are documented in
    shennong_amd/pipeline.py)"""
    with pytest.raises(ValueError) as err:
        pipeline.get_default_config('bad')
    assert 'invalid features "bad"' in str(err.value)

    config = pipeline.get_default_config('mfcc')
    del config['mfcc']
    with pytest.raises(ValueError) as err:
        pipeline.extract_features(config, utterances)
    assert 'the configuration does not define any features' in str(err.value)

    config = pipeline.get_default_config('mfcc')
    config['plp'] = config['mfcc']
    with pytest.raises(ValueError) as err:
        pipeline.extract_features(config, utterances)
    assert 'more than one features extraction processor' in str(err.value)

    config = pipeline.get_default_config('mfcc')
    config['invalid'] = config['mfcc']
    with pytest.raises(ValueError) as err:
        pipeline.extract_features(config, utterances)
    assert 'invalid keys in configuration' in str(err.value)

    with pytest.raises(ValueError) as err:
        pipeline.get_default_config('mfcc', with_vtln=True)
    assert 'must be False, "simple" or "full" but is "True"' in str(err.value)
    with pytest.raises(ValueError) as err:
        pipeline.get_default_config('mfcc', with_pitch='bad')
    assert 'with_pitch argument must be' in str(err.value)
    for kwargs in (dict(with_vtln='simple'), dict(with_vtln='full'), dict(with_pitch='crepe')):
        with pytest.raises(ValueError) as err:
            pipeline.get_default_config('mfcc', **kwargs)
        assert 'not available in this backend' in str(err.value)
    with pytest.raises(ValueError):
        pipeline.get_default_config('bottleneck')

    config = pipeline.get_default_config('mfcc', with_cmvn=True)
    del config['cmvn']['with_vad']
    parsed = pipeline._init_config(config)
    assert parsed['cmvn']['with_vad']
    config = pipeline.get_default_config('mfcc', with_cmvn=True)
    del config['cmvn']['by_speaker']
    assert not pipeline._init_config(config)['cmvn']['by_speaker']
    config = pipeline.get_default_config('mfcc', with_pitch='kaldi')
    del config['pitch']['postprocessing']
    assert pipeline._init_config(config)['pitch']['postprocessing'] == {}

    config = pipeline.get_default_config('spectrogram')
    config['vtln'] = {}
    with pytest.raises(ValueError) as err:
        pipeline.extract_features(config, utterances)
    assert 'do not support VTLN' in str(err.value)

    config = pipeline.get_default_config('mfcc', with_cmvn=True)
    with pytest.raises(ValueError) as err:
        pipeline.extract_features(config, Utterances([('toto', WAV)]))
    assert 'no speaker information provided' in str(err.value)


def test_init_warps(utterances, capsys):
    """reference test_pipeline.py:163-205"""
    log = get_logger('test', 'info')
    with pytest.raises(ValueError) as err:
        pipeline._init_warps({}, pipeline.get_default_config('spectrogram'), utterances, log)
    assert 'features do not support VTLN' in str(err.value)
    for warps in ({}, {'a': 0}, {'utt1': 0}):
        with pytest.raises(ValueError) as err:
            pipeline._init_warps(warps, pipeline.get_default_config('mfcc'), utterances, log)
        assert 'warps do not match utterances' in str(err.value)
    capsys.readouterr()
    w = pipeline._init_warps({'utt1': 1.0, 'utt2': 0.0}, pipeline.get_default_config('mfcc'),
                             utterances, log)
    assert 'warps are defined by utterance' in capsys.readouterr().err
    assert w == {'utt1': 1.0, 'utt2': 0.0}
    w = pipeline._init_warps({'speaker1': 1.0, 'speaker2': 0.0},
                             pipeline.get_default_config('mfcc'), utterances, log)
    assert 'warps are defined by speaker' in capsys.readouterr().err
    assert w == {'utt1': 1.0, 'utt2': 0.0}
    with pytest.raises(ValueError) as err:
        pipeline._init_warps({'speaker1': 'a', 'speaker2': 0.0},
                             pipeline.get_default_config('mfcc'), utterances, log)
    assert 'could not convert string to float' in str(err.value)


def test_utterances_index():
    """reference utterances.py semantics used by the pipeline"""
    utts = Utterances([('u1', WAV, 's1', 0, 1), ('u2', WAV, 's2', 1, 1.2)])
    assert utts.has_speakers() and utts.format() == 4
    assert utts.format(type=str) == '<utterance-id> <audio-file> <speaker-id> <tstart> <tstop>'
    assert sorted(utts.by_speaker().keys()) == ['s1', 's2']
    assert utts.duration() == pytest.approx(1.2)
    with pytest.warns(UserWarning):
        u3 = Utterances([('u3', WAV_8K, 1, 3)])
    assert u3['u3'].duration < 0.5 and not u3.has_speakers()
    with pytest.raises(ValueError) as err:
        Utterances([('1', WAV, 1, 0)])
    assert 'we must have 0 <= tstart < tstop' in str(err.value)
    with pytest.raises(ValueError):
        Utterances([('u1', WAV), ('u1', WAV)])
    meta = Audio.scan(WAV)
    assert (meta.nchannels, meta.sample_rate, meta.nsamples) == (1, 16000, 22713)


# ---- extraction (GPU) ----------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('features', pipeline.valid_features())
def test_extract_features(gpu, utterances, features):
    """reference test_pipeline.py:279-315"""
    config = pipeline.get_default_config(features, with_delta=True)
    feat1 = pipeline.extract_features(config, utterances)['utt1']
    assert feat1.is_valid() and feat1.shape[0] == 140 and feat1.dtype == np.float32

    config = pipeline.get_default_config(features, with_delta=True, with_pitch='kaldi')
    feat2 = pipeline.extract_features(config, utterances)['utt1']
    assert feat2.is_valid() and feat2.shape == (140, feat1.shape[1] + 3)

    config = pipeline.get_default_config(features, with_delta=True)
    feat3 = pipeline.extract_features(config, Utterances([('utt1', WAV, 0, 1)]))['utt1']
    assert feat3.is_valid() and feat3.shape == (98, feat1.shape[1])


@pytest.mark.gpu
@pytest.mark.parametrize('by_speaker, with_vad',
                         [(s, v) for s in (True, False) for v in (True, False)])
def test_pipeline_cmvn(gpu, by_speaker, with_vad):
    """reference test_pipeline.py:318-329 + equality with the step-by-step processors and the
    oracle's VAD / CMVN arithmetic"""
    from oracle import oracle as orc
    from shennong_amd.processor import EnergyProcessor, MfccProcessor
    from shennong_amd.postprocessor import VadPostProcessor
    utts = Utterances([('utt1', WAV, 'spk1', 0, 1), ('utt2', WAV, 'spk1', 0.5, 1.4),
                       ('utt3', WAV, 'spk2', 0, 1.4)])
    config = pipeline.get_default_config('mfcc', with_cmvn=True)
    config['mfcc']['dither'] = 0
    config['cmvn']['by_speaker'] = by_speaker
    config['cmvn']['with_vad'] = with_vad
    feats = pipeline.extract_features(config, utts)
    assert list(feats.keys()) == ['utt1', 'utt2', 'utt3']
    mfcc = {u.name: MfccProcessor(dither=0).process(u.load_audio()) for u in utts}
    vads = {}
    for u in utts:
        w = None
        if with_vad:
            w = VadPostProcessor().process(EnergyProcessor().process(u.load_audio())).data[:, 0]
            # (the energy used for the VAD is dithered like in the reference: compare loosely)
            assert w.shape[0] == mfcc[u.name].nframes
        vads[u.name] = w
    groups = {'utt1': 'a', 'utt2': 'a', 'utt3': 'b'} if by_speaker else {k: k for k in mfcc}
    for name, f in feats.items():
        assert f.is_valid() and f.shape == mfcc[name].shape and f.dtype == np.float32
        assert f.properties['pipeline'] == [
            {'name': 'mfcc', 'columns': [0, 12]}, {'name': 'cmvn', 'columns': [0, 12]}]
        assert f.properties['speaker'] == utts[name].speaker
        assert f.properties['audio']['file'] == WAV
        stats = f.properties['cmvn']['stats']
        if not with_vad:
            want = np.zeros((2, 14))
            for other in mfcc:
                if groups[other] == groups[name]:
                    orc.cmvn_accumulate(mfcc[other].data, stats=want)
            np.testing.assert_allclose(stats, want, rtol=1e-12)
        assert np.array_equal(f.data, orc.cmvn_apply(mfcc[name].data, stats))


@pytest.mark.gpu
def test_pipeline_full(gpu, tmp_path, capsys):
    """reference test_pipeline.py:355-410: different sampling rates, speakers and segments; the
    result must equal the chain of the individual processors"""
    import scipy.io.wavfile
    from shennong_amd.processor import (
        KaldiPitchPostProcessor, KaldiPitchProcessor, MfccProcessor)
    from shennong_amd.postprocessor import CmvnPostProcessor, DeltaPostProcessor
    audio = Audio.load(WAV)
    wav_f32 = str(tmp_path / 'test.float32.wav')
    scipy.io.wavfile.write(wav_f32, 16000, (audio.data / 2 ** 15).astype(np.float32))
    with pytest.warns(UserWarning):
        index = Utterances([('u1', WAV, 's1', 0, 1), ('u2', wav_f32, 's2', 1, 1.2),
                            ('u3', WAV_8K, 's1', 1, 3)])
    config = pipeline.get_default_config('mfcc', with_cmvn=True, with_delta=True, with_pitch='kaldi')
    config['cmvn']['with_vad'] = False
    config['mfcc']['dither'] = 0
    config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
    feats = pipeline.extract_features(config, index, njobs=2, log=get_logger('test', 'info'))
    messages = capsys.readouterr().err
    assert 'INFO - test - get 3 utterances from 2 speakers in 3 audio files' in messages
    assert 'WARNING - test - several sample rates found in audio files' in messages
    p1, p2, p3 = (feats[u].properties for u in ('u1', 'u2', 'u3'))
    assert p1['audio']['file'] == WAV and p1['audio']['duration'] == 1.0
    assert p2['audio']['duration'] == pytest.approx(0.2)
    assert p3['audio']['duration'] < 0.5
    assert p1['mfcc'] == p2['mfcc'] and p1['mfcc']['sample_rate'] != p3['mfcc']['sample_rate']
    assert p1.keys() == {'audio', 'mfcc', 'cmvn', 'pitch', 'delta', 'speaker', 'pipeline'}
    assert p1.keys() == p2.keys() == p3.keys()
    assert p1['pipeline'] == p2['pipeline'] == p3['pipeline']
    assert feats['u1'].shape == (98, 42) and feats['u2'].shape == (18, 42)
    assert all(f.dtype == np.float32 and f.is_valid() for f in feats.values())

    # the collection goes through pickle (the .pkl serializer, joblib transport) before anything
    # has read the lazily made properties
    from shennong_amd import FeaturesCollection
    fresh = pipeline.extract_features(config, index, njobs=2, log=get_logger('test', 'error'))
    fresh.save(str(tmp_path / 'feats.pkl'))
    assert FeaturesCollection.load(str(tmp_path / 'feats.pkl')) == feats

    # the same thing, one processor at a time
    mfcc, pitch = {}, {}
    for u in index:
        a = u.load_audio()
        mfcc[u.name] = MfccProcessor(sample_rate=a.sample_rate, dither=0).process(a)
        raw = KaldiPitchProcessor(sample_rate=a.sample_rate).process(a)
        pitch[u.name] = KaldiPitchPostProcessor(delta_pitch_noise_stddev=0).process(raw)
    cmvn = {'s1': CmvnPostProcessor(13), 's2': CmvnPostProcessor(13)}
    for u in index:
        cmvn[u.speaker].accumulate(mfcc[u.name])
    for u in index:
        want = DeltaPostProcessor().process(cmvn[u.speaker].process(mfcc[u.name]))
        want = want.concatenate(pitch[u.name], tolerance=2)
        assert np.array_equal(feats[u.name].data, want.data)
        assert np.array_equal(feats[u.name].times, want.times)


@pytest.mark.gpu
def test_pipeline_warps(gpu, utterances):
    """reference test_pipeline.py:346-352"""
    config = pipeline.get_default_config('mfcc')
    feats = pipeline.extract_features(config, utterances, warps={'speaker1': 1.2, 'speaker2': 0.85})
    assert feats['utt1'].properties['mfcc']['vtln_warp'] == 1.2
    assert feats['utt2'].properties['mfcc']['vtln_warp'] == 0.85


@pytest.mark.gpu
@pytest.mark.parametrize('features', ['mfcc', 'filterbank', 'plp'])
def test_extract_features_warp(gpu, utterances, features):
    """reference pipeline.py:650-696 (what the VTLN trainer calls between its iterations; the reference has no
    test of its own for it): every utterance warped by ONE factor, deltas when configured, nothing else - the
    result is the chain delta.process(features.process(audio, vtln_warp=warp)), properties included"""
    from shennong_amd import processor
    from shennong_amd.postprocessor import DeltaPostProcessor
    config = pipeline.get_default_config(features, with_pitch='kaldi', with_cmvn=True, with_delta=True)
    config[features]['dither'] = 0
    got = pipeline.extract_features_warp(config, utterances, 1.1, get_logger('test', 'error'), njobs=2)
    cls = {'mfcc': processor.MfccProcessor, 'filterbank': processor.FilterbankProcessor,
           'plp': processor.PlpProcessor}[features]
    proc = cls(**config[features])
    assert list(got.keys()) == ['utt1', 'utt2']
    for utt in utterances:
        want = DeltaPostProcessor(**config['delta']).process(proc.process(utt.load_audio(), vtln_warp=1.1))
        assert got[utt.name] == want           # data, times and properties (no 'audio' / 'speaker' / pitch / cmvn)
        assert got[utt.name].properties[features]['vtln_warp'] == 1.1
    del config['delta']
    plain = pipeline.extract_features_warp(config, utterances, 0.9, get_logger('test', 'error'))
    assert plain['utt1'] == proc.process(utterances['utt1'].load_audio(), vtln_warp=0.9)
    with pytest.raises(ValueError, match='do not support VTLN'):
        pipeline.extract_features_warp(pipeline.get_default_config('spectrogram'), utterances, 1.1)
    with pytest.raises(ValueError, match='must be strictly positive'):
        pipeline.extract_features_warp(config, utterances, 1.1, njobs=0)


@pytest.mark.gpu
def test_pipeline_pinned_index(gpu):
    """a pinned index (Utterances.pin(): the audio in ONE page-locked block) goes through the pipeline from where
    it lies - one-shot and streamed in several batches, 1 and 3 batches in flight: the same features, bit for
    bit, as the same utterances given as separate arrays"""
    from shennong_amd import synth
    waves = synth.utterances(7, 40, 16000)
    rng = np.random.RandomState(3)
    cuts = rng.randint(6000, 16000, size=40)
    index = Utterances([(f'u{i:02d}', Audio(waves[i, :cuts[i]].copy(), 16000, validate=False), f's{i % 3}')
                        for i in range(40)])
    config = pipeline.get_default_config('filterbank', with_pitch='kaldi', with_cmvn=True, with_delta=True)
    config['filterbank']['dither'] = 0
    config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
    want = pipeline.extract_features(config, index)
    pinned = index.pin()
    got = pipeline.extract_features(config, pinned)
    assert list(got) == list(want)
    for name in want:
        assert np.array_equal(got[name].data, want[name].data) and np.array_equal(got[name].times, want[name].times)
        assert got[name].properties['speaker'] == want[name].properties['speaker']
        assert got[name].properties['pipeline'] == want[name].properties['pipeline']
    for njobs in (1, 3):
        out, stats = {}, pipeline.RunStats()
        count = pipeline.extract_features_streamed(config, pinned, out.update, max_batch_duration=6.0, njobs=njobs,
                                                   stats=stats)
        assert count == 40 and list(out) == list(want)
        assert all(np.array_equal(out[name].data, want[name].data) for name in want)
        seen = stats.as_dict()
        assert seen['utterances'] == 40 and seen['batches'] > 3 and seen['gpu_ms'] > 0
        assert seen['bytes_up'] == 2 * int(cuts.sum())     # (every sample crossed the link once: the second
        assert seen['bytes_down'] == sum(f.data.nbytes for f in want.values())   # pass reads the resident audio)
    # no room (or not enough) for the audio between the passes: the second pass sends its batches ahead again
    for budget in (0, 150000):
        out, stats = {}, pipeline.RunStats()
        pipeline.extract_features_streamed(config, pinned, out.update, max_batch_duration=6.0, njobs=2,
                                           resident_bytes=budget, stats=stats)
        assert list(out) == list(want) and all(np.array_equal(out[name].data, want[name].data) for name in want)
        up = stats.as_dict()['bytes_up']
        assert (up == 4 * int(cuts.sum())) if budget == 0 else (2 * int(cuts.sum()) < up < 4 * int(cuts.sum()))


@pytest.mark.gpu
def test_extract_features_falls_back_to_batches_when_the_corpus_does_not_fit(gpu, monkeypatch):
    """a corpus whose single batch would not fit the free HBM (2.3 GB of tracker scratch per hour of audio: 125 h
    would ask for more than the device has) is extracted in bounded batches behind the same call: same keys, same
    order, same bits"""
    from shennong_amd import synth
    waves = synth.utterances(11, 24, 16000)
    index = Utterances([(f'u{i:02d}', Audio(waves[i, :9000 + 250 * i].copy(), 16000, validate=False), f's{i % 4}')
                        for i in range(24)])
    config = pipeline.get_default_config('filterbank', with_pitch='kaldi', with_cmvn=True, with_delta=True)
    config['filterbank']['dither'] = 0
    config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
    want = pipeline.extract_features(config, index)
    assert not pipeline._too_large_for_one_batch(index)
    # (an hour of audio "needs" 2^50 bytes: nothing fits one batch; the streamed default is ten minutes then)
    monkeypatch.setattr(pipeline, '_BATCH_BYTES_PER_HOUR', 1 << 50)
    monkeypatch.setattr(pipeline, 'default_batch_duration', lambda depth=1: 5.0)
    assert pipeline._too_large_for_one_batch(index)
    got = pipeline.extract_features(config, index, njobs=2)
    assert list(got) == list(want)
    for name in want:
        assert got[name] == want[name], name


@pytest.mark.gpu
def test_pipeline_from_wav_files_natively(gpu, tmp_path):
    """a corpus of WAV files goes through the native reader (16-bit mono PCM side by side into page-locked memory,
    other sample types through the Python reader): the pipeline, process_all and Utterances.pin() see the same
    samples as utterances whose audio was loaded one by one"""
    import scipy.io.wavfile
    from shennong_amd.processor import MfccProcessor
    audio = Audio.load(WAV)
    wav_f32 = str(tmp_path / 'f32.wav')
    scipy.io.wavfile.write(wav_f32, 16000, (audio.data / 2 ** 15).astype(np.float32))
    items = [('a', WAV, 's1', 0, 1.0), ('b', WAV, 's2', 0.3, 1.3), ('c', wav_f32, 's1', 0.1, 0.9),
             ('d', WAV, 's2', 0.0, 1.4195), ('e', WAV, 's1', 1.0, 1.4)]
    with pytest.warns(UserWarning):
        items.append(('f', WAV, 's2', 1.0, 3.0))      # (runs past the end of the file: cut there)
        index = Utterances(items)
    loaded = Utterances([(u.name, u.load_audio(), u.speaker) for u in index])
    config = pipeline.get_default_config('mfcc', with_pitch='kaldi', with_cmvn=True, with_delta=True)
    config['mfcc']['dither'] = 0
    config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
    want = pipeline.extract_features(config, loaded)
    for name, got in (('files', pipeline.extract_features(config, index)),
                      ('pinned files', pipeline.extract_features(config, index.pin()))):
        assert sorted(got) == sorted(want), name
        for key in want:
            assert np.array_equal(got[key].data, want[key].data), (name, key)
    assert pipeline.extract_features(config, index)['b'].properties['audio']['file'] == WAV
    from shennong_amd.processor import (
        EnergyProcessor, KaldiPitchProcessor, PlpProcessor, SpectrogramProcessor)
    for proc in (MfccProcessor(dither=0), KaldiPitchProcessor(), SpectrogramProcessor(dither=0),
                 EnergyProcessor(dither=0), PlpProcessor(dither=0, rasta=True)):
        a, b, c = proc.process_all(index), proc.process_all(loaded), proc.process_all(index.pin())
        assert list(a) == list(b) == list(c), proc.name
        assert all(a[k] == b[k] and np.array_equal(c[k].data, b[k].data) for k in b), proc.name
    with pytest.raises(ValueError, match='mismatch in sample rates'):
        KaldiPitchProcessor(sample_rate=8000).process_all(index)


@pytest.mark.gpu
@pytest.mark.parametrize('features', ['mfcc', 'filterbank', 'plp', 'spectrogram'])
def test_pipeline_resident_equals_by_stage(gpu, tmp_path, features):
    """the device-resident pipeline (one upload, one download) returns exactly what the chain of
    host-pointer processor calls returns: data, times and properties"""
    # (spectrograms of 8 kHz and 16 kHz audio have different widths and cannot share CMVN statistics)
    third = WAV if features == 'spectrogram' else WAV_8K
    with pytest.warns(UserWarning):
        index = Utterances([('u1', WAV, 's1', 0, 1), ('u2', WAV, 's2', 0.3, 1.4),
                            ('u3', third, 's1', 1, 3), ('u4', WAV, 's2', 0.9, 1.3)])
    config = pipeline.get_default_config(features, with_cmvn=True, with_delta=True,
                                         with_pitch='kaldi')
    config[features]['dither'] = 0
    config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
    log = get_logger('test', 'error')
    for with_vad, by_speaker in ((False, True), (True, False)):
        config['cmvn']['with_vad'] = with_vad
        config['cmvn']['by_speaker'] = by_speaker
        cfg = pipeline._init_config(config, log=log)
        warps = None if features == 'spectrogram' else {'u1': 1.1, 'u2': 0.9, 'u3': 1.0, 'u4': 1.2}
        a = pipeline._extract_features(cfg, index, warps, log)
        b = pipeline._extract_features_by_stage(cfg, index, warps, log)
        assert list(a.keys()) == list(b.keys())
        for k in a:
            if with_vad:
                # (the VAD weights come from a dithered energy in both paths: compare the layout only)
                assert a[k].shape == b[k].shape and a[k].properties.keys() == b[k].properties.keys()
            else:
                assert a[k] == b[k], k


def test_utterances_file_and_duration(tmp_path):
    """reference test/test_utterances.py: index files, ordering, fit_to_duration"""
    utts = Utterances([('b', WAV, 's1', 0, 1), ('a', WAV, 's1', 0.5, 1.4), ('c', WAV_8K, 's2', 0, 1)])
    assert [u.name for u in utts] == ['c', 'a', 'b']   # sorted by (audio file, name)
    index = str(tmp_path / 'index.txt')
    utts.save(index)
    assert Utterances.load(index) == utts
    with pytest.raises(ValueError) as err:
        Utterances.load(str(tmp_path / 'nothere'))
    assert 'not found' in str(err.value)
    sub = utts.fit_to_duration(0.8)
    assert sub['a'].duration == pytest.approx(0.8) and sub['c'].duration == pytest.approx(0.8)
    assert 'b' not in sub.by_name()
    with pytest.raises(ValueError) as err:
        utts.fit_to_duration(5)
    assert 'of audio available but 5s requested' in str(err.value)
    with pytest.warns(UserWarning):
        assert len(utts.fit_to_duration(5, truncate=True)) == 3
    with pytest.raises(ValueError):
        utts.fit_to_duration(0)
    with pytest.raises(ValueError):
        Utterances([('x', WAV)]).fit_to_duration(1)


# ---- streamed extraction (BASELINE config 5: a corpus that does not sit in memory at once) ------------
def _segments_index():
    return Utterances([
        ('u1', WAV, 's1', 0, 0.9), ('u2', WAV, 's2', 0.2, 1.3), ('u3', WAV, 's1', 0.5, 1.4),
        ('u4', WAV, 's2', 0.1, 0.6), ('u5', WAV, 's3', 0.3, 1.2), ('u6', WAV, 's1', 0.7, 1.4)])


def test_batches():
    index = _segments_index()
    utts = list(index)
    durations = [u.duration for u in utts]
    got = list(pipeline._batches(utts, 2.0))
    assert [u.name for b in got for u in b] == [u.name for u in utts]  # order kept, nothing lost
    for b in got:
        assert sum(u.duration for u in b) <= 2.0 or len(b) == 1
    assert len(got) > 1
    assert list(pipeline._batches(utts, sum(durations) + 1)) == [utts]
    assert [len(b) for b in pipeline._batches(utts, 1e-3)] == [1] * len(utts)
    with pytest.raises(ValueError, match='max_batch_duration'):
        pipeline.extract_features_streamed(
            pipeline.get_default_config('mfcc'), index, lambda f: None, max_batch_duration=0)
    with pytest.raises(ValueError, match='no speaker information'):
        pipeline.extract_features_streamed(
            pipeline.get_default_config('mfcc', with_cmvn=True),
            Utterances([('a', WAV), ('b', WAV)]), lambda f: None)


def test_streamed_driver_logic(monkeypatch):
    """host logic of the two passes with the device pipeline replaced by a stand-in: every speaker's
    statistics are the sum over ALL batches (in utterance order) before any batch is normalised, and
    a cross-process reduction is applied exactly once, after the first pass"""
    index = _segments_index()
    calls = []

    def fake(config, utterances, warps, log, tolerance=2, stats_hook=None, stats_only=False, **_resident):
        utts = list(utterances)
        per_utt = np.stack([np.full((2, 3), float(int(u.name[1]))) for u in utts])
        if stats_only:
            calls.append(('stats', [u.name for u in utts]))
            return [u.speaker for u in utts], per_utt
        names = list(dict.fromkeys(u.speaker for u in utts))
        partial = np.stack([sum(per_utt[i] for i, u in enumerate(utts) if u.speaker == s)
                            for s in names])
        stats = stats_hook(names, partial) if stats_hook else partial
        calls.append(('apply', [u.name for u in utts]))
        out = {u.name: stats[names.index(u.speaker)][0, 0] for u in utts}
        return (out, lambda: None) if _resident.get('defer') else out

    monkeypatch.setattr(pipeline, '_extract_features', fake)
    config = pipeline.get_default_config('mfcc', with_cmvn=True)
    out = {}
    n = pipeline.extract_features_streamed(config, index, out.update, max_batch_duration=2.0)
    assert n == 6
    # s1 = u1 + u3 + u6, s2 = u2 + u4, s3 = u5 whatever the batch boundaries are
    assert out == {'u1': 10.0, 'u3': 10.0, 'u6': 10.0, 'u2': 6.0, 'u4': 6.0, 'u5': 5.0}
    kinds = [k for k, _ in calls]
    assert kinds == ['stats'] * (len(kinds) // 2) + ['apply'] * (len(kinds) // 2) and len(kinds) > 2
    assert [x for k, b in calls if k == 'stats' for x in b] == ['u1', 'u2', 'u3', 'u4', 'u5', 'u6']

    reduced = []

    def reduce(names, stats):
        reduced.append(list(names))
        return stats * 2

    out = {}
    pipeline.extract_features_streamed(config, index, out.update, max_batch_duration=2.0,
                                       stats_reduce=reduce)
    assert reduced == [['s1', 's2', 's3']]
    assert out['u1'] == 20.0 and out['u5'] == 10.0

    # CMVN by utterance (or no CMVN): a single pass, no statistics carried between the batches
    calls.clear()
    config['cmvn']['by_speaker'] = False
    pipeline.extract_features_streamed(config, index, out.update, max_batch_duration=2.0)
    assert all(k == 'apply' for k, _ in calls)


def test_resident_waves_budget():
    """the audio kept in HBM between the two streamed passes: a byte budget, every buffer handed back once"""
    class Buffer:
        def __init__(self, nbytes):
            self.nbytes, self.freed = nbytes, 0

        def free(self, synced=False):
            self.freed += 1

    keep = pipeline._ResidentWaves(100)
    a, b, c = Buffer(60), Buffer(50), Buffer(40)
    assert keep.offer((0, 16000), a, 'soff a') and keep.held == 60
    assert not keep.offer((1, 16000), b, 'soff b') and keep.held == 60     # over the budget: the caller frees
    assert keep.offer((2, 16000), c, 'soff c') and keep.held == 100
    assert keep.take((1, 16000)) is None
    assert keep.take((0, 16000)) == (a, 'soff a') and keep.held == 40 and keep.take((0, 16000)) is None
    keep.clear()
    assert (a.freed, b.freed, c.freed) == (0, 0, 1) and keep.held == 0 and keep.take((2, 16000)) is None
    assert not pipeline._ResidentWaves(0).offer('k', Buffer(1), None)


def test_timed_launch_statistics(tmp_path):
    """tools/timed_launch_stats.py keeps the last K dispatches of every kernel of a rocprofv3 kernel trace"""
    import csv
    import subprocess
    import sys
    trace = tmp_path / 'x_kernel_trace.csv'
    with open(trace, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['Kind', 'Kernel_Name', 'Start_Timestamp', 'End_Timestamp'])
        t = 0
        for i in range(13):                      # 10 settle + 3 warm-up: slow
            w.writerow(['KERNEL_DISPATCH', 'k<1>(int)', t, t + 2000])
            t += 3000
        for i in range(5):                       # 5 timed
            w.writerow(['KERNEL_DISPATCH', 'k<1>(int)', t, t + 1000 + i])
            t += 3000
        w.writerow(['KERNEL_DISPATCH', 'other()', t, t + 50])
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools',
                        'timed_launch_stats.py')
    out = subprocess.run([sys.executable, tool, str(trace), '5'], capture_output=True, text=True, check=True)
    rows = list(csv.DictReader(out.stdout.splitlines()))
    assert [r['Name'] for r in rows] == ['k<1>(int)', 'other()']
    assert rows[0]['Calls'] == '18' and rows[0]['TimedCalls'] == '5'
    assert float(rows[0]['AverageNs']) == 1002.0 and rows[0]['MinNs'] == '1000' and rows[0]['MaxNs'] == '1004'
    assert abs(float(rows[0]['AllCallsAverageNs']) - (13 * 2000 + 5010) / 18) < 0.1
    assert rows[1]['TimedCalls'] == '1'


@pytest.mark.parametrize('depth', [1, 2, 3, 8])
def test_batches_in_flight(depth):
    """results in order whatever the completion order, never more than `depth` batches started and not
    handed over, the first failure reaches the caller"""
    import threading
    import time
    lock = threading.Lock()
    state = {'running': 0, 'peak': 0, 'started': 0}

    def work(b, batch):
        with lock:
            state['running'] += 1
            state['started'] += 1
            state['peak'] = max(state['peak'], state['running'])
        time.sleep(0.002 * ((7 * b) % 5))
        with lock:
            state['running'] -= 1
        if batch == 'bad':
            raise ValueError('batch %d' % b)
        return b, batch

    batches = ['b%d' % i for i in range(11)]
    handed = []
    for b, batch in pipeline._in_flight(iter(batches), work, depth):
        assert state['started'] - len(handed) <= depth
        handed.append((b, batch))
    assert handed == list(enumerate(batches))
    assert state['peak'] <= depth and (depth == 1 or state['peak'] > 1)
    assert list(pipeline._in_flight(iter([]), work, depth)) == []
    with pytest.raises(ValueError, match='batch 2'):
        list(pipeline._in_flight(iter(['a', 'b', 'bad', 'c', 'bad']), work, depth))


@pytest.mark.parametrize('depth', [1, 2, 3])
def test_batches_in_flight_deferred(depth):
    """deferred results (the copy of a batch still on its way when its work returns): handed over in order, every
    `finish` called exactly once and before its result is handed over; with one batch at a time the finish of
    batch k comes AFTER the work of batch k + 1 (that is the overlap); a consumer that stops early, or a batch
    that fails, still has every started copy waited for"""
    events = []

    def work(b, batch):
        events.append(('work', b))
        if batch == 'bad':
            raise ValueError('batch %d' % b)
        return (b, batch), (lambda b=b: events.append(('finish', b)))

    batches = ['b%d' % i for i in range(7)]
    handed = []
    for b, batch in pipeline._in_flight(iter(batches), work, depth, deferred=True):
        assert ('finish', b) in events
        handed.append((b, batch))
    assert handed == list(enumerate(batches))
    assert sorted(e for e in events if e[0] == 'finish') == [('finish', b) for b in range(7)]
    if depth == 1:
        for b in range(6):
            assert events.index(('work', b + 1)) < events.index(('finish', b))
    assert list(pipeline._in_flight(iter([]), work, depth, deferred=True)) == []
    # early close: the copies of what ran are waited for
    del events[:]
    gen = pipeline._in_flight(iter(batches), work, depth, deferred=True)
    assert next(gen) == (0, 'b0')
    gen.close()
    started = sorted(b for kind, b in events if kind == 'work')
    assert sorted(b for kind, b in events if kind == 'finish') == started
    # a failing batch: the earlier results were either handed over or finished
    del events[:]
    with pytest.raises(ValueError, match='batch 2'):
        list(pipeline._in_flight(iter(['a', 'b', 'bad', 'c']), work, depth, deferred=True))
    finished = sorted(b for kind, b in events if kind == 'finish')
    assert finished[:2] == [0, 1] and 2 not in finished


def test_batches_in_flight_closed_early():
    """a consumer that stops early (its sink raised): what has not started is cancelled, what is running has
    FINISHED when close() returns - the audio buffers behind the batches may then be released"""
    import threading
    import time
    lock = threading.Lock()
    state = {'running': 0, 'done': []}

    def work(b, batch):
        with lock:
            state['running'] += 1
        time.sleep(0.05)
        with lock:
            state['running'] -= 1
            state['done'].append(b)
        return b

    gen = pipeline._in_flight(iter(range(12)), work, 3)
    assert next(gen) == 0
    gen.close()
    with lock:
        assert state['running'] == 0, state
        assert len(state['done']) < 12   # (the tail of the corpus never ran)
    done = list(state['done'])
    time.sleep(0.12)
    assert state['done'] == done        # (and nothing starts afterwards)


@pytest.mark.gpu
@pytest.mark.parametrize('by_speaker', [True, False, None])
def test_pipeline_streamed(gpu, tmp_path, by_speaker):
    """the streamed pipeline (several batches, speaker statistics from a first pass, features
    recomputed in the second) returns exactly the one-shot pipeline's features, and the incremental
    Kaldi writer stores them like FeaturesCollection.save does"""
    from shennong_amd import FeaturesCollection
    from shennong_amd.serializers import KaldiStreamWriter
    index = _segments_index()
    config = pipeline.get_default_config(
        'mfcc', with_cmvn=by_speaker is not None, with_delta=True, with_pitch='kaldi')
    config['mfcc']['dither'] = 0
    config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
    if by_speaker is not None:
        config['cmvn']['by_speaker'] = by_speaker
        config['cmvn']['with_vad'] = False  # (the VAD energy is dithered per batch position)
    warps = {'s1': 1.1, 's2': 0.9, 's3': 1.0}
    whole = pipeline.extract_features(config, index, warps=warps)
    got = FeaturesCollection()
    name = str(tmp_path / 'corpus.ark')
    with KaldiStreamWriter(name, scp=True) as writer:
        def sink(feats):
            got.update(feats)
            writer.write(feats)
        n = pipeline.extract_features_streamed(config, index, sink, warps=warps,
                                               max_batch_duration=1.5)
    assert n == 6 and list(got.keys()) == list(whole.keys())
    for k in whole:
        assert got[k] == whole[k], k
    assert FeaturesCollection.load(name) == whole
    whole.save(str(tmp_path / 'oneshot.ark'), scp=True)
    for suffix in ('.ark', '.times.ark'):
        assert open(str(tmp_path / 'corpus') + suffix, 'rb').read() == \
            open(str(tmp_path / 'oneshot') + suffix, 'rb').read()


@pytest.mark.gpu
@pytest.mark.parametrize('njobs', [1, 3])
@pytest.mark.parametrize('resident_bytes', [0, 40000, 16 << 30])
def test_pipeline_streamed_resident_waves(gpu, resident_bytes, njobs):
    """the audio uploaded by the statistics pass stays in HBM for the second pass (all of it, the first
    batches only, none of it): same features as the one-shot pipeline, nothing left allocated"""
    index = _segments_index()
    config = pipeline.get_default_config('mfcc', with_cmvn=True, with_delta=True, with_pitch='kaldi')
    config['mfcc']['dither'] = 0
    config['pitch']['postprocessing']['delta_pitch_noise_stddev'] = 0
    config['cmvn']['with_vad'] = False
    whole = pipeline.extract_features(config, index)
    kept = []
    offer = pipeline._ResidentWaves.offer

    def spy(self, key, d_wave, soff):
        ok = offer(self, key, d_wave, soff)
        kept.append((key, ok, self.held))
        return ok

    got = {}
    pipeline._ResidentWaves.offer = spy
    try:
        n = pipeline.extract_features_streamed(config, index, got.update, max_batch_duration=1.5,
                                               resident_bytes=resident_bytes, njobs=njobs)
    finally:
        pipeline._ResidentWaves.offer = offer
    assert n == 6 and list(got) == list(whole.keys())
    for k in whole:
        assert got[k] == whole[k], k
    if resident_bytes == 0:
        assert not kept
    else:
        assert len(kept) >= 3 and any(ok for _, ok, _ in kept)
        assert all(held <= resident_bytes for _, _, held in kept)
        assert all(ok for _, ok, _ in kept) == (resident_bytes > 40000)


@pytest.mark.gpu
@pytest.mark.parametrize('features', ['mfcc', 'filterbank'])
def test_pipeline_random_terms_are_a_function_of_the_utterance(gpu, features):
    """the reference's DEFAULT configuration dithers (processor/base.py:122), its VAD energy too, and the
    pitch post-processing adds noise to the delta-pitch column: the pipeline draws all of them from one named
    noise call keyed per utterance, so the statistics pass and the apply pass of the streamed pipeline see
    the same features - streamed == one-shot bit for bit with every random term ON, whatever the batch
    split, and two runs agree"""
    index = _segments_index()
    config = pipeline.get_default_config(features, with_cmvn=True, with_delta=True, with_pitch='kaldi')
    assert config[features]['dither'] == 1.0 and config['cmvn']['with_vad'] is True
    assert config['pitch']['postprocessing']['delta_pitch_noise_stddev'] > 0
    config['cmvn']['by_speaker'] = True
    whole = pipeline.extract_features(config, index)
    again = pipeline.extract_features(config, index)
    quiet = pipeline.get_default_config(features, with_cmvn=True, with_delta=True, with_pitch='kaldi')
    quiet[features]['dither'] = 0
    quiet['cmvn']['by_speaker'] = True
    silent = pipeline.extract_features(quiet, index)
    for k in whole:
        assert whole[k] == again[k], k
        assert not np.array_equal(whole[k].data, silent[k].data), k   # (the noise is there)
    for batch_s in (0.7, 1.5, 100.0):
        got = {}
        pipeline.extract_features_streamed(config, index, got.update, max_batch_duration=batch_s)
        assert list(got) == list(whole.keys())
        for k in whole:
            assert got[k] == whole[k], (k, batch_s)
    # a processor called directly keeps the reference's behaviour: a new stream per call
    from shennong_amd.processor import MfccProcessor
    audio = index['u1'].load_audio()
    proc = MfccProcessor(sample_rate=audio.sample_rate)
    assert not np.array_equal(proc.process(audio).data, proc.process(audio).data)


@pytest.mark.gpu
def test_result_blocks_and_pooled_buffers(gpu):
    """the host array of a batch is a pooled page-locked block: views keep it alive, the last one gives it
    back, a new block is only page-locked while few result bytes are alive; device buffers are pooled"""
    import gc
    from shennong_amd import _backend
    block_cls = _backend._ResultBlock
    gc.collect()
    held0 = block_cls._held
    a = _backend.result_array((300000, 4), np.float32)         # 4.8 MB: page-locked
    assert isinstance(a.base, block_cls) and block_cls._held > held0
    a[...] = 7.0
    rows = a[10:20]
    address = a.ctypes.data
    del a
    gc.collect()
    assert block_cls._held > held0 and float(rows.sum()) == 7.0 * 40   # a view keeps the block
    del rows
    gc.collect()
    assert block_cls._held == held0
    pooled = {blk[1] for blk in _backend.STAGING._free}
    assert address in pooled                                    # the block came back to the pool
    b = _backend.result_array((300000, 4), np.float32)          # ... and a pooled block serves the next batch
    assert isinstance(b.base, block_cls) and b.ctypes.data in pooled
    small = _backend.result_array((10, 4), np.float32)          # not worth a pinned buffer
    assert not isinstance(small.base, block_cls)
    fresh, block_cls._FRESH = block_cls._FRESH, 1 << 20
    try:
        # larger than every pooled block, and none may be made: plain memory
        rows_c = (max([blk[0] for blk in _backend.STAGING._free] + [0]) >> 4) + 400000
        c = _backend.result_array((rows_c, 4), np.float32)
        assert not isinstance(c.base, block_cls)
        del c
    finally:
        block_cls._FRESH = fresh
    del b
    # device buffers: a freed buffer is handed out again for a request of similar size (the pool is bounded - 8 GiB -
    # and earlier tests of the session may have filled it: start from an empty one)
    _backend.DEVICE_POOL.clear()
    d = _backend.DeviceBuffer(3 << 20)
    ptr = d.ptr
    d.free()
    parked = {blk[1] for blocks in _backend.DEVICE_POOL._free.values() for blk in blocks}
    assert ptr in parked
    e = _backend.DeviceBuffer((3 << 20) - 4096)
    assert e.ptr in parked and e.nbytes == (3 << 20) - 4096 and e._capacity >= e.nbytes
    e.free()
    rows = [np.full(50000 + 7 * k, k, dtype=np.int16) for k in range(120)]   # 12 MB: the threaded path
    buf = _backend.upload_rows(rows, np.int16)
    back = np.empty(sum(r.shape[0] for r in rows), dtype=np.int16)
    buf.download(back)
    buf.free()
    assert np.array_equal(back, np.concatenate(rows))


@pytest.mark.gpu
def test_non_finite_features_are_refused_on_the_device(gpu):
    """Features.validate's data check runs on the batch while it is still in HBM"""
    from shennong_amd import _backend
    x = np.zeros(100003, dtype=np.float32)
    buf = _backend.DeviceBuffer(x.nbytes)
    try:
        buf.upload(x)
        _backend.check_finite_device(buf.ptr, x.size)
        for where, value in ((0, np.nan), (57, np.inf), (100002, -np.inf), (100000, np.nan)):
            y = x.copy()
            y[where] = value
            buf.upload(y)
            with pytest.raises(ValueError, match='non-finite'):
                _backend.check_finite_device(buf.ptr, x.size)
            if where < 64:  # (a clean part of the same buffer)
                _backend.check_finite_device(buf.ptr + 4 * 64, 1000)
        y = x.copy()
        y[::7] = np.float32(3.4e38)
        y[1::7] = np.float32(-1e-45)
        buf.upload(y)
        _backend.check_finite_device(buf.ptr, x.size)
    finally:
        buf.free()


def test_copy_properties():
    """the pipeline's properties copy is a deep copy: equal, nothing mutable shared"""
    import copy
    props = {'pipeline': [{'name': 'mfcc', 'columns': [0, 12]}, {'name': 'cmvn', 'columns': [0, 12]}],
             'mfcc': {'dither': np.float32(0.0), 'window_type': 'povey', 'snip_edges': True,
                      'vtln_warp': 1.1, 'num_ceps': 13, 'none': None},
             'cmvn': {'stats': np.arange(28.0).reshape(2, 14)}, 'other': (1, [2])}
    got = pipeline.copy_properties(props)
    want = copy.deepcopy(props)
    assert got.keys() == want.keys() and got['mfcc'] == want['mfcc'] and got['pipeline'] == want['pipeline']
    assert np.array_equal(got['cmvn']['stats'], props['cmvn']['stats'])
    assert got['cmvn']['stats'] is not props['cmvn']['stats']
    assert got['pipeline'] is not props['pipeline'] and got['pipeline'][0] is not props['pipeline'][0]
    assert got['pipeline'][0]['columns'] is not props['pipeline'][0]['columns']
    assert type(got['mfcc']['dither']) is np.float32
    assert got['other'] == props['other'] and got['other'][1] is not props['other'][1]
