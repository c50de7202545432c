"""Features / FeaturesCollection / Utterance / Utterances against the behaviours the reference's own
tests pin (reference test/test_features.py, test/test_utterances.py, test/test_base.py).  Host-only:
the 'mfcc' features are a fixed random matrix with the processor's times and properties."""

import os

import numpy as np
import pytest

from conftest import GOLDEN
from shennong_amd import Features, FeaturesCollection, Utterances
from shennong_amd.logger import get_logger
from shennong_amd.processor import MfccProcessor
from shennong_amd.utterances import Utterance

WAV = os.path.join(GOLDEN, 'test.wav')
WAV_8K = os.path.join(GOLDEN, 'test.8k.wav')
RNG = np.random.default_rng(3)


@pytest.fixture(scope='module')
def mfcc():
    proc = MfccProcessor()
    return Features(RNG.standard_normal((140, 13)).astype(np.float32), proc.times(140),
                    properties=proc.get_properties(vtln_warp=1.0))


# ---- Features ------------------------------------------------------------------------------------------
@pytest.mark.parametrize('args, kwargs, message', [
    ((0, 0), dict(properties=0), 'data must be a numpy array'),
    ((np.asarray([0]), 0), dict(properties=0), 'times must be a numpy array'),
    ((np.asarray([0]), np.asarray([0])), dict(properties=0), 'properties must be a dictionnary'),
    ((np.asarray([0]), np.asarray([0])), dict(properties={0: 0}), 'data dimension must be 2'),
    ((np.asarray([[0], [0]]), np.zeros((2, 2, 2))), {}, 'times dimension must be 1 or 2'),
    ((np.ones((10, 3)), np.ones((10, 3))), {}, 'times shape[1] must be 2, it is 3'),
])
def test_features_init_bad(args, kwargs, message):
    with pytest.raises(ValueError) as err:
        Features(*args, **kwargs)
    assert message in str(err.value)


def test_features_not_finite_and_unsorted():
    data = RNG.random((12, 2))
    data[2, 1] = np.nan
    with pytest.raises(ValueError, match='data contains non-finite numbers'):
        Features(data, np.ones((12,)))
    with pytest.raises(ValueError, match='times is not sorted in increasing order'):
        Features(RNG.random((10, 3)), RNG.random((10, 2)))
    # 1-D times (the stop column of the processor's times) are valid
    assert Features(RNG.random((10, 5)), MfccProcessor().times(10)[:, 1], validate=False).is_valid()


def test_features_dict_roundtrip(mfcc):
    a = mfcc._to_dict()
    assert Features._from_dict(a) == mfcc
    with pytest.raises(ValueError, match='missing keys: times'):
        Features._from_dict({'data': a['data'], 'properties': a['properties']})


def test_features_equal_and_close(mfcc):
    assert mfcc == mfcc and mfcc.is_close(mfcc)
    same = mfcc.copy()
    assert mfcc == same and mfcc.is_close(same)
    wider = mfcc.concatenate(mfcc)  # not the same shape
    assert not mfcc == wider and not mfcc.is_close(wider)
    as64 = mfcc.copy(dtype=np.float64)  # not the same dtype: not equal, but close
    assert not mfcc == as64 and mfcc.is_close(as64)
    other = Features(mfcc.data, mfcc.times, properties={'foo': 0})
    assert not mfcc == other and not mfcc.is_close(other)
    shifted = Features(mfcc.data, mfcc.times + 1, properties=mfcc.properties)
    assert not mfcc == shifted and not mfcc.is_close(shifted) and not mfcc.is_close(shifted, atol=1)
    moved = Features(mfcc.data + 1, mfcc.times, properties=mfcc.properties)
    assert not mfcc == moved and not mfcc.is_close(moved) and mfcc.is_close(moved, atol=1)


def test_features_validate_and_copy(mfcc):
    with pytest.raises(ValueError, match='mismatch in number of frames'):
        Features(mfcc.data, mfcc.times[:-2, :], validate=False).validate()
    dup = mfcc.copy()  # new arrays
    assert dup == mfcc and dup is not mfcc
    assert dup.data is not mfcc.data and dup.times is not mfcc.times
    assert dup.properties is not mfcc.properties
    shared = Features(mfcc.data, mfcc.times, properties=mfcc.properties, validate=False)
    assert shared == mfcc and shared.data is mfcc.data and shared.times is mfcc.times
    assert shared.properties is mfcc.properties
    for bad in (9.12, 0, -10):
        with pytest.raises(ValueError):
            mfcc.copy(subsample=bad)
    assert mfcc.copy(subsample=2).shape == (70, 13)


def test_features_of_a_batch_own_their_times_and_properties():
    """the utterances of one launch are built on SHARED times / properties objects; what each Features
    hands out is its own copy (made when first read), equal to what the eager constructor holds"""
    times = np.arange(8, dtype=np.float64).reshape(4, 2)
    shared = {'mfcc': {'num_ceps': 13, 'table': np.arange(3.0)}, 'pipeline': [{'name': 'mfcc', 'columns': [0, 1]}]}
    data = np.ones((8, 2), dtype=np.float32)
    a = Features._of_batch(data[:4], times, shared, {'speaker': 'anna', 'audio': {'file': None, 'duration': 1.0}})
    b = Features._of_batch(data[4:], times, shared, None)
    eager = Features(data[:4], times.copy(), properties=dict(
        shared, speaker='anna', audio={'file': None, 'duration': 1.0}))
    assert a == eager and a.is_valid() and a.nframes == 4 and a.ndims == 2
    assert b.properties.keys() == shared.keys() and 'speaker' not in shared
    assert a.times is not times and a.times is a.times and a.properties is a.properties
    a.times[0, 0] = 99.0
    a.properties['mfcc']['num_ceps'] = 7
    a.properties['mfcc']['table'][0] = -1.0
    a.properties['pipeline'][0]['columns'][0] = 5
    a.properties['audio']['duration'] = 2.0
    assert times[0, 0] == 0.0 and b.times[0, 0] == 0.0
    assert shared['mfcc']['num_ceps'] == 13 and b.properties['mfcc']['num_ceps'] == 13
    assert shared['mfcc']['table'][0] == 0.0 and b.properties['pipeline'][0]['columns'] == [0, 1]
    c = a.copy()
    assert c == a and c.properties is not a.properties and a != eager
    assert Features._from_dict(b._to_dict()) == b


def test_features_of_a_batch_pickle():
    """pipeline output is built on a lazy history (pipeline._Meta) that holds closures and the
    batch's cache; pickling a Features (the .pkl serializer, joblib transport) makes its own times and
    properties first and sends only those"""
    import pickle
    from shennong_amd.pipeline import _Meta
    times = np.arange(8, dtype=np.float64).reshape(4, 2)
    cache = {}
    root = _Meta({'mfcc': {'num_ceps': 13}, 'pipeline': [{'name': 'mfcc', 'columns': [0, 12]}]}, 2, 4, times, 'mfcc')
    scale = 3.0
    lazy = root.derive(cache, ('cmvn', 0), lambda m: dict(m.properties, cmvn={'stats': np.full(3, scale)}))
    data = np.arange(8, dtype=np.float32).reshape(4, 2)
    a = Features._of_batch(data, times, lazy, {'speaker': 'anna'})
    b = Features._of_batch(data, times, lazy, {'speaker': 'anna'})
    back = pickle.loads(pickle.dumps(a))            # (never read before pickling)
    assert back == b and back.properties['cmvn']['stats'][0] == 3.0 and back._shared is None
    col = FeaturesCollection(x=Features._of_batch(data, times, lazy, None))
    again = pickle.loads(pickle.dumps(col))
    assert again['x'] == col['x'] and type(again) is FeaturesCollection
    assert a._shared == (None, None, None)           # (a Features that was read no longer holds the batch's history)


def test_features_json_properties_of_a_batch(tmp_path):
    """serializers write the properties of a pipeline batch without materialising them per utterance: the text
    `Features._json_properties` splices (shared history encoded once + this utterance's own entries) is the JSON
    of `properties`, and an archive written from such features reads back equal"""
    import json
    from shennong_amd import serializers
    from shennong_amd.pipeline import _Meta, _utterance_properties
    times = np.arange(8, dtype=np.float64).reshape(4, 2)
    cache = {}
    root = _Meta({'mfcc': {'num_ceps': 13}, 'pipeline': [{'name': 'mfcc', 'columns': [0, 12]}]}, 2, 4, times, 'mfcc')
    lazy = root.derive(cache, ('cmvn', 0), lambda m: dict(m.properties, cmvn={'stats': np.arange(6.0).reshape(2, 3)}))

    class Utt:
        audio_file, tstart, tstop, duration, speaker = None, 0.5, 1.5, 1.0, 'anna'
    data = np.arange(16, dtype=np.float32).reshape(8, 2)
    a = Features._of_batch(data[:4], times, lazy, (_utterance_properties, Utt, 16000))
    b = Features._of_batch(data[4:], times, lazy, None)
    plain = Features(data[:4], times.copy(), properties={'x': {'y': np.float32(2.5)}})
    empty = Features(data[:4], times.copy())
    for feat in (a, b, plain, empty):
        text = feat._json_properties(serializers._dumps)
        assert json.loads(text, object_hook=serializers._decode_arrays).keys() == feat.properties.keys()
        again = json.loads(serializers._dumps(feat.properties))
        assert json.loads(text) == again
    assert lazy._json is not None and a._json_properties(serializers._dumps).count('"speaker": "anna"') == 1
    col = FeaturesCollection(a=Features._of_batch(data[:4], times, lazy, (_utterance_properties, Utt, 16000)),
                             b=Features._of_batch(data[4:], times, lazy, None), c=plain, d=empty)
    col.save(str(tmp_path / 'x.ark'))
    back = FeaturesCollection.load(str(tmp_path / 'x.ark'))
    assert back == col and back['a'].properties['audio']['tstop'] == 1.5
    assert np.array_equal(back['a'].properties['cmvn']['stats'], np.arange(6.0).reshape(2, 3))


def test_features_concatenate(mfcc, capsys):
    both = mfcc.concatenate(mfcc)
    assert both.nframes == mfcc.nframes and both.ndims == 2 * mfcc.ndims
    assert both.properties != mfcc.properties
    assert both.properties['mfcc'] == mfcc.properties['mfcc']
    with pytest.raises(ValueError, match='times are not equal'):
        mfcc.concatenate(Features(mfcc.data, mfcc.times + 1))
    f1 = Features(RNG.random((12, 2)), np.ones((12,)))
    f2 = Features(RNG.random((10, 2)), np.ones((10,)))
    with pytest.raises(ValueError, match='features have a different number of frames'):
        f1.concatenate(f2, tolerance=0)
    with pytest.raises(ValueError, match='features differs number of frames, and greater than '):
        f1.concatenate(f2, tolerance=1)
    assert f1.concatenate(f2, tolerance=2, log=get_logger('test', 'info')).shape == (10, 4)
    assert 'WARNING' in capsys.readouterr().err
    assert f2.concatenate(f1, tolerance=2, log=get_logger('test', 'warning')).shape == (10, 4)
    assert 'WARNING' in capsys.readouterr().err


def test_collection(mfcc):
    assert FeaturesCollection().is_valid()
    assert FeaturesCollection(mfcc=mfcc).is_valid()
    assert not FeaturesCollection(mfcc=Features(np.asarray([0]), 0, validate=False)).is_valid()
    f1 = Features(RNG.random((10, 2)), np.ones((10,)))
    f2 = Features(RNG.random((10, 2)), np.ones((10,)))
    fc1 = FeaturesCollection(f1=f1, f2=f2)
    fc2 = FeaturesCollection(f1=f1, f2=Features(f2.data + 1, f2.times))
    assert fc1.is_close(fc1) and not fc1.is_close(fc2) and fc1.is_close(fc2, atol=1)
    assert not fc1.is_close(FeaturesCollection(f1=f1, f3=f2))


def test_collection_partition_and_trim():
    fc = FeaturesCollection(
        f1=Features(RNG.random((10, 2)), np.ones((10,))), f2=Features(RNG.random((10, 2)), np.ones((10,))),
        f3=Features(RNG.random((5, 2)), np.ones((5,))))
    with pytest.raises(ValueError, match='not defined in the partition index: f3'):
        fc.partition({'f1': 'p1', 'f2': 'p1'})
    parts = fc.partition({'f1': 'p1', 'f2': 'p1', 'f3': 'p2'})
    assert sorted(parts) == ['p1', 'p2']
    assert sorted(parts['p1']) == ['f1', 'f2'] and sorted(parts['p2']) == ['f3']
    assert all(p.is_valid() for p in parts.values())
    fc = parts['p1']
    with pytest.raises(ValueError, match='Vad keys are different from this keys.'):
        fc.trim({'f3': np.ones(10, bool), 'f4': np.ones(10, bool)})
    with pytest.raises(ValueError, match='Vad arrays must be arrays of bool.'):
        fc.trim({'f1': np.arange(10), 'f2': np.arange(10)})
    with pytest.raises(ValueError, match='Vad arrays length must be equal to the number of frames.'):
        fc.trim({'f1': np.ones(10, bool), 'f2': np.ones(5, bool)})
    trimmed = fc.trim({'f1': np.array([True] * 7 + [False] * 3), 'f2': np.array([True] * 5 + [False] * 5)})
    assert trimmed['f1'].shape == (7, 2) and trimmed['f2'].shape == (5, 2)
    assert all(f.is_valid() for f in trimmed.values())


# ---- Utterance -----------------------------------------------------------------------------------------
@pytest.mark.parametrize('args, message', [
    (((),), 'invalid utterance format'), ((0,), 'invalid utterance format'),
    ((0, 0, 0, 0, 0, 0), 'invalid utterance format'), ((0, 0), '0: file not found'),
    ((0, WAV, None, 1), 'both tstart and tstop must be defined or None'),
    ((0, WAV, 0, None), 'both tstart and tstop must be defined or None'),
    ((0, WAV, 'spk', 1, 0), 'we must have 0 <= tstart < tstop'),
    ((0, WAV, -1, 0), 'we must have 0 <= tstart < tstop'),
    ((0, WAV, 'abc', 0), 'cannot cast tstart as float'),
    ((0, WAV, 1, 'abc'), 'cannot cast tstop as float')])
def test_utterance_bad(args, message):
    with pytest.raises(ValueError) as err:
        Utterance(*args)
    assert message in str(err.value)


@pytest.mark.parametrize('fmt, args, speaker, interval, text', [
    (1, (), None, None, ''), (2, ('spk',), 'spk', None, ' spk'),
    (3, (0, 1), None, (0, 1), ' 0.0 1.0'), (4, ('spk', 0, 1), 'spk', (0, 1), ' spk 0.0 1.0')])
def test_utterance_formats(audio, fmt, args, speaker, interval, text):
    utt = Utterance('name', WAV, *args)
    assert utt.format == fmt and utt.name == 'name' and utt.audio_file == WAV
    assert utt.speaker == speaker
    assert str(utt) == f'name {WAV}{text}'
    if interval is None:
        assert utt.tstart is None and utt.tstop is None
        assert utt.duration == pytest.approx(audio.duration)
        assert utt.load_audio() == audio
    else:
        assert (utt.tstart, utt.tstop) == interval and utt.duration == 1
        assert np.all(utt.load_audio().data == audio.data[:16000])


def test_utterance_truncate(audio):
    with pytest.warns(UserWarning) as warn:
        utt = Utterance('name', WAV, 'spk', 0, 10)
    assert 'asking interval (0.0, 10.0)' in warn[0].message.args[0]
    assert utt.duration == pytest.approx(audio.duration)
    assert utt.load_audio() == audio
    with pytest.warns(UserWarning) as warn:
        utt = Utterance('name', WAV, 'spk', 1, 5)
    assert 'asking interval (1.0, 5.0)' in warn[0].message.args[0]
    assert utt.duration + 1 == pytest.approx(audio.duration)
    assert np.all(utt.load_audio().data == audio.data[16000:])


# ---- Utterances ----------------------------------------------------------------------------------------
@pytest.mark.parametrize('items, message', [
    ([], 'empty input utterances'), ([(0,)], 'invalid utterance format: (0,)'),
    ([('utt1', WAV), 0], 'utterance must be an iterable'),
    ([('utt1', WAV), ('utt2', WAV, 'spk')], 'utterances format is not homogeneous'),
    ([('utt1', WAV), ('utt1', WAV)], 'duplicates found')])
def test_utterances_bad(items, message):
    with pytest.raises(ValueError) as err:
        Utterances(items)
    assert message in str(err.value)


@pytest.mark.parametrize('with_speakers', [True, False])
def test_utterances_index(with_speakers):
    spk = (lambda s: (s,)) if with_speakers else (lambda s: ())
    utterances = Utterances([
        ['utt1', WAV, *spk('spk1'), 0, 1],  # a list is fine too
        ('utt2', WAV_8K, *spk('spk1'), 0, 1.2), ('utt3', WAV, *spk('spk2'), 0, 1)])
    assert len(utterances) == 3
    assert [u.name for u in utterances] == ['utt2', 'utt1', 'utt3']  # sorted by audio file
    assert utterances['utt1'].name == 'utt1' and utterances['utt1'].tstop == 1
    fmt = 4 if with_speakers else 3
    assert utterances.format() == fmt and utterances.format(type=int) == fmt
    assert utterances.format(type=str) == (
        '<utterance-id> <audio-file> <speaker-id> <tstart> <tstop>' if with_speakers
        else '<utterance-id> <audio-file> <tstart> <tstop>')
    assert utterances.has_speakers() is with_speakers
    if with_speakers:
        assert {k: len(v) for k, v in utterances.by_speaker().items()} == {'spk1': 2, 'spk2': 1}
    else:
        with pytest.raises(ValueError, match='utterances have no speaker information'):
            utterances.by_speaker()
    assert list(utterances.by_name().keys()) == ['utt2', 'utt1', 'utt3']
    assert utterances.duration() == 3.2


def test_utterances_save_load(tmpdir):
    filename = str(tmpdir / 'utts')
    utts = Utterances([('utt1', WAV, 0, 1), ('utt2', WAV, 0, 1.2), ('utt3', WAV, 0, 1)])
    utts.save(filename)
    assert Utterances.load(filename) == utts
    with pytest.raises(ValueError):
        Utterances.load('/spam/spam/i/love/spam')


@pytest.mark.parametrize('shuffle', (False, True))
def test_utterances_fit_to_duration(shuffle):
    with pytest.raises(ValueError, match='utterances have no speaker information'):
        Utterances([('utt1', WAV, 0, 0.5), ('utt2', WAV, 0, 1)]).fit_to_duration(10)
    with pytest.raises(ValueError, match='duration must be a positive number'):
        Utterances([('utt1', WAV, 'spk', 0, 0.5)]).fit_to_duration(0)
    utts = Utterances([('utt1', WAV, 'spk1', 0, 0.5), ('utt2', WAV, 'spk1', 0, 1),
                       ('utt3', WAV, 'spk2', 0, 1.2)])
    fit = utts.fit_to_duration(1, shuffle=shuffle)
    assert fit.duration() == 2
    if not shuffle:
        assert fit == Utterances([('utt1', WAV, 'spk1', 0, 0.5), ('utt2', WAV, 'spk1', 0, 0.5),
                                  ('utt3', WAV, 'spk2', 0, 1)])
    message = 'speaker spk2: only 1.2s of audio available but 1.5s requested'
    with pytest.raises(ValueError) as err:
        utts.fit_to_duration(1.5, shuffle=shuffle)
    assert message in str(err.value)
    with pytest.warns(UserWarning) as warn:
        fit = utts.fit_to_duration(1.5, shuffle=shuffle, truncate=True)
    assert message in warn[0].message.args[0]
    if not shuffle:
        assert fit == utts


# ---- BaseProcessor parameters, logger (reference test/test_base.py, test/test_logger.py) -------------------
def test_base_get_set_params():
    from shennong_amd.base import BaseProcessor

    class NoSignature(BaseProcessor):
        def __init__(self, *params):
            pass

    class Nested(BaseProcessor):
        def __init__(self, a, mfcc):
            self.a = a
            self.mfcc = mfcc

    assert BaseProcessor._get_param_names() == []
    with pytest.raises(RuntimeError, match='specify their parameters in the signature'):
        NoSignature().get_params()
    inner = MfccProcessor()
    nested = Nested(1, inner)
    assert inner.get_params() == {
        k.replace('mfcc__', ''): v for k, v in nested.get_params().items() if 'mfcc__' in k}
    assert nested.set_params() == nested
    with pytest.raises(ValueError, match='invalid parameter spam'):
        nested.set_params(spam=True)
    nested.set_params(mfcc__sample_rate=2)
    assert nested.mfcc.sample_rate == 2
    assert nested.get_params()['mfcc__sample_rate'] == 2


@pytest.mark.parametrize('level', ['debug', 'info', 'warning', 'error'])
def test_logger_levels(capsys, level):
    from shennong_amd.logger import null_logger
    quiet = null_logger()
    for emit in (quiet.debug, quiet.info, quiet.warning, quiet.error):
        emit('NOTHING')
    captured = capsys.readouterr()
    assert not captured.out and not captured.err
    log = get_logger('test', level=level)
    log.debug('DEBUG')
    log.info('INFO')
    log.warning('WARNING')
    log.error('ERROR')
    captured = capsys.readouterr()
    assert not captured.out
    order = ['debug', 'info', 'warning', 'error']
    for name in order:
        assert (name.upper() in captured.err) is (order.index(name) >= order.index(level))
    with pytest.raises(ValueError, match='invalid logging level'):
        get_logger('test', level='bad')
