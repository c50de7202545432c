"""Serializers: the reference's test/test_serializers.py re-expressed against
shennong_amd.serializers.  Host-only: the features are built from a fixed random matrix shaped like
the MFCCs of test.wav, no device call."""

import os
import shutil

import numpy as np
import pytest

from shennong_amd import Features, FeaturesCollection
from shennong_amd import serializers
from shennong_amd.logger import get_logger
from shennong_amd.processor import MfccProcessor

log = get_logger('test', 'info')

SERIALIZERS = [
    serializers.NumpySerializer, serializers.MatlabSerializer, serializers.PickleSerializer,
    serializers.KaldiSerializer, serializers.CsvSerializer]


@pytest.fixture(scope='module')
def mfcc():
    proc = MfccProcessor()
    data = np.random.default_rng(0).standard_normal((140, 13)).astype(np.float32)
    return Features(data, proc.times(140), properties=proc.get_properties(vtln_warp=1.0))


@pytest.fixture()
def mfcc_col(mfcc):
    return FeaturesCollection(mfcc=mfcc)


def _name(serializer):
    return 'feats.ark' if serializer is serializers.KaldiSerializer else 'feats'


def _serializer_for(name_or_none, filename):
    return serializers.get_serializer(FeaturesCollection, filename, log, name_or_none)


def test_serializer_lookup():
    """by name, by file extension, and the ways both can fail"""
    for name, cls in serializers.supported_serializers().items():
        filename = 'foo.ark' if name == 'kaldi' else 'foo.file'
        assert isinstance(_serializer_for(name, filename), cls)
        assert not os.path.isfile(filename), 'nothing is created before save()'
    for ext, cls in serializers.supported_extensions().items():
        assert isinstance(_serializer_for(None, 'foo' + ext), cls)
    for args, message in (
            ((int, 'foo', log, None), 'must be shennong.features.FeaturesCollection'),
            ((FeaturesCollection, 'foo.spam', log, None), 'invalid extension .spam'),
            ((FeaturesCollection, 'foo.spam', log, 'spam'), 'invalid serializer spam'),
            # (h5features needs a binding this backend does not ship)
            ((FeaturesCollection, 'foo.h5f', log, None), 'invalid extension .h5f'),
            ((FeaturesCollection, 'foo.file', log, 'kaldi'),
             'the file extension must be ".ark", it is ".file"')):
        with pytest.raises(ValueError) as err:
            serializers.get_serializer(*args)
        assert message in str(err.value)


def test_save_and_load_preconditions(tmpdir, mfcc, mfcc_col):
    with pytest.raises(IOError, match='file not found'):
        _serializer_for(None, 'foo.npz').load()
    taken = str(tmpdir.join('foo.npz'))
    open(taken, 'w').write('something')
    with pytest.raises(IOError, match='file already exists'):
        _serializer_for(None, taken).save(mfcc_col)
    free = _serializer_for(None, str(tmpdir.join('bar.npz')))
    with pytest.raises(ValueError, match='features must be FeaturesCollection but are Features'):
        free.save(mfcc)
    with pytest.raises(ValueError, match='features are not valid'):
        free.save(FeaturesCollection(mfcc=Features(data=mfcc.data, times=0, validate=False)))


def _collections(mfcc):
    """name -> (collection, extra check on what was loaded back)"""
    wide_text = dict(mfcc.properties, comments='使用人口について正確な統計はないが、日本国')
    return {
        'plain': (FeaturesCollection(mfcc=mfcc), lambda c: (
            c['mfcc'].dtype == np.float32
            and c['mfcc'].properties['pipeline'] == [{'name': 'mfcc', 'columns': [0, 12]}])),
        'times_1d': (FeaturesCollection(mfcc=Features(
            np.random.default_rng(4).random((10, 5)), MfccProcessor().times(10)[:, 1])),
            lambda c: c['mfcc'].times.shape == (10,)),
        'utf8': (FeaturesCollection({'æðÐ': Features(mfcc.data, mfcc.times, wide_text)}),
                 lambda c: 'æðÐ' in c),
        'two_dtypes': (FeaturesCollection(mfcc32=mfcc, mfcc64=mfcc.copy(dtype=np.float64)),
                       lambda c: (c['mfcc64'].dtype, c['mfcc32'].dtype) == (np.float64, np.float32)),
    }


@pytest.mark.parametrize('case', ['plain', 'times_1d', 'utf8', 'two_dtypes'])
@pytest.mark.parametrize('serializer', SERIALIZERS)
def test_round_trip(mfcc, serializer, case, tmpdir):
    collection, also = _collections(mfcc)[case]
    target = str(tmpdir.join(_name(serializer)))
    serializer(FeaturesCollection, target, log).save(collection)
    assert os.path.exists(target)
    loaded = serializer(FeaturesCollection, target, log).load()
    assert loaded == collection and also(loaded)


@pytest.mark.parametrize('scp', [True, False])
def test_kaldiserializer(mfcc_col, tmpdir, scp):
    mfcc_col.save(str(tmpdir.join('foo.ark')), scp=scp)
    for f in ('foo.ark', 'foo.times.ark', 'foo.properties.json'):
        assert os.path.isfile(str(tmpdir.join(f)))
    if scp:
        lines = open(str(tmpdir.join('foo.scp'))).read().split('\n')
        assert lines[0].startswith('mfcc ') and ':' in lines[0]
        key, where = lines[0].split(' ')
        ark, offset = where.rsplit(':', 1)
        blob = open(ark, 'rb').read()
        assert blob[int(offset):int(offset) + 5] == b'\0BDM '   # the scp points at the binary marker
        assert os.path.isfile(str(tmpdir.join('foo.times.scp')))
    assert FeaturesCollection.load(str(tmpdir.join('foo.ark'))) == mfcc_col


def test_kaldiserializer_baditems(tmpdir, mfcc_col):
    col2 = FeaturesCollection(one=mfcc_col['mfcc'], two=mfcc_col['mfcc'])
    mfcc_col.save(str(tmpdir.join('one.ark')))
    col2.save(str(tmpdir.join('two.ark')))
    os.remove(str(tmpdir.join('two.times.ark')))
    shutil.copyfile(str(tmpdir.join('one.times.ark')), str(tmpdir.join('two.times.ark')))
    with pytest.raises(ValueError) as err:
        FeaturesCollection.load(str(tmpdir.join('two.ark')))
    assert 'items differ in data and times' in str(err.value)
    os.remove(str(tmpdir.join('one.properties.json')))
    shutil.copyfile(str(tmpdir.join('two.properties.json')), str(tmpdir.join('one.properties.json')))
    with pytest.raises(ValueError) as err:
        FeaturesCollection.load(str(tmpdir.join('one.ark')))
    assert 'items differ in data and properties' in str(err.value)


@pytest.mark.parametrize('missing', ['foo.ark', 'foo.times.ark', 'foo.properties.json'])
def test_kaldiserializer_badfile(tmpdir, mfcc_col, missing):
    filename = str(tmpdir.join('foo.ark'))
    mfcc_col.save(filename)
    os.remove(str(tmpdir.join(missing)))
    with pytest.raises(IOError) as err:
        FeaturesCollection.load(filename)
    assert 'file not found: {}'.format(str(tmpdir.join(missing))) in str(err.value)


def test_csvserializer_bad(tmpdir, mfcc_col):
    np.savetxt(str(tmpdir.join('foo.csv')), mfcc_col['mfcc'].data)
    with pytest.raises(ValueError) as err:
        FeaturesCollection.load(str(tmpdir), serializer='csv')
    assert 'failed to parse header' in str(err.value)
    np.savetxt(str(tmpdir.join('foo.csv')), mfcc_col['mfcc'].data, header='data_dtype',
               comments='# ')
    with pytest.raises(ValueError) as err:
        FeaturesCollection.load(str(tmpdir), serializer='csv')
    assert 'failed to parse header' in str(err.value)
    with pytest.raises(OSError) as err:
        FeaturesCollection.load(str(tmpdir.join('notexistingfolder')))
    assert 'directory not found' in str(err.value)
    with pytest.raises(IOError) as err:
        mfcc_col.save(str(tmpdir))
    assert 'already exists: ' in str(err.value)


@pytest.mark.parametrize('serializer, with_props', [
    (s, p) for s in serializers.supported_serializers() for p in (True, False)])
def test_no_properties(tmpdir, mfcc_col, serializer, with_props):
    filename = str(tmpdir.join('feats.ark' if serializer == 'kaldi' else 'feats'))
    mfcc_col.save(filename, serializer=serializer, with_properties=with_props)
    mfcc_col2 = FeaturesCollection.load(filename, serializer=serializer)
    if with_props:
        assert mfcc_col == mfcc_col2
    else:
        assert mfcc_col != mfcc_col2
        for name in mfcc_col:
            assert mfcc_col2[name].properties == {}
            assert np.all(mfcc_col[name].data == mfcc_col2[name].data)
            assert np.all(mfcc_col[name].times == mfcc_col2[name].times)


def test_partition_and_trim(mfcc):
    """reference features_collection.py partition / trim"""
    col = FeaturesCollection(a=mfcc, b=mfcc.copy(), c=mfcc.copy())
    parts = col.partition({'a': 's1', 'b': 's2', 'c': 's1'})
    assert sorted(parts) == ['s1', 's2'] and sorted(parts['s1']) == ['a', 'c']
    with pytest.raises(ValueError) as err:
        col.partition({'a': 's1'})
    assert 'not defined in the partition index' in str(err.value)
    vad = {k: np.arange(140) % 2 == 0 for k in col}
    trimmed = col.trim(vad)
    assert all(t.shape == (70, 13) for t in trimmed.values())
    assert np.array_equal(trimmed['a'].data, mfcc.data[::2])
    with pytest.raises(ValueError):
        col.trim({'a': vad['a']})
    with pytest.raises(ValueError):
        col.trim({k: v.astype(int) for k, v in vad.items()})


@pytest.mark.parametrize('scp', [True, False])
@pytest.mark.parametrize('with_properties', [True, False])
def test_kaldi_stream_writer(tmpdir, mfcc, scp, with_properties):
    """batches appended one after the other give byte for byte the files of one
    FeaturesCollection.save of all the items, and load back equal"""
    rng = np.random.default_rng(1)
    items = FeaturesCollection()
    for i, n in enumerate((140, 3, 77, 1, 25)):
        proc = MfccProcessor()
        items[f'utt{i}-é'] = Features(
            rng.standard_normal((n, 13)).astype(np.float32), proc.times(n),
            properties=proc.get_properties(vtln_warp=1.0 + i / 10))
    items['times1d'] = Features(np.ones((4, 2), np.float64), np.arange(4.0), validate=True)
    keys = list(items)
    streamed = str(tmpdir.join('streamed.ark'))
    with serializers.KaldiStreamWriter(streamed, scp=scp, with_properties=with_properties) as w:
        w.write({k: items[k] for k in keys[:2]})
        w.write({})
        w.write({k: items[k] for k in keys[2:5]})
        w.write({keys[5]: items[keys[5]]})
        with pytest.raises(ValueError, match='already written'):
            w.write({keys[0]: items[keys[0]]})
    with pytest.raises(ValueError, match='closed'):
        w.write({})
    w.close()  # idempotent
    items.save(str(tmpdir.join('oneshot.ark')), scp=scp, with_properties=with_properties)
    suffixes = ['.ark', '.times.ark', '.properties.json'] + (['.scp', '.times.scp'] if scp else [])
    for suffix in suffixes:
        a = open(str(tmpdir.join('streamed' + suffix)), 'rb').read()
        b = open(str(tmpdir.join('oneshot' + suffix)), 'rb').read()
        if suffix.endswith('.scp'):
            b = b.replace(b'oneshot', b'streamed')
        assert a == b, suffix
    assert not os.path.exists(str(tmpdir.join('streamed.scp'))) or scp
    loaded = FeaturesCollection.load(streamed)
    assert list(loaded.keys()) == keys
    for k in keys:
        assert np.array_equal(loaded[k].data, items[k].data) and loaded[k].dtype == items[k].dtype
        assert np.array_equal(loaded[k].times, items[k].times)
        assert (loaded[k].properties == items[k].properties) is with_properties or \
            not items[k].properties
    # never overwrites, wrong extension refused
    with pytest.raises(IOError, match='already exists'):
        serializers.KaldiStreamWriter(streamed)
    with pytest.raises(ValueError, match='extension must be'):
        serializers.KaldiStreamWriter(str(tmpdir.join('x.npz')))


def test_kaldi_stream_writer_float_matrices(tmpdir, mfcc):
    """double=False: Kaldi float matrices (FM), half the bytes, float32 features round-trip exactly"""
    name = str(tmpdir.join('f32.ark'))
    with serializers.KaldiStreamWriter(name, double=False) as w:
        w.write({'a': mfcc, 'b': mfcc})
    blob = open(name, 'rb').read()
    assert blob.count(b'\0BFM ') == 2 and b'\0BDM ' not in blob
    assert len(blob) < 2 * (mfcc.data.size * 4 + 64)
    loaded = FeaturesCollection.load(name)
    assert loaded['a'] == mfcc and loaded['b'].dtype == np.float32


def _random_properties(rng, depth=0):
    def leaf():
        kind = int(rng.integers(9))
        return [lambda: int(rng.integers(-1000, 1000)), lambda: float(rng.standard_normal()),
                lambda: bool(rng.integers(2)), lambda: 'str%d é' % int(rng.integers(100)),
                lambda: np.float32(rng.standard_normal()), lambda: np.int64(rng.integers(1000)),
                lambda: rng.standard_normal((int(rng.integers(1, 4)), int(rng.integers(1, 5)))),
                lambda: [int(x) for x in rng.integers(0, 50, size=int(rng.integers(0, 4)))],
                lambda: rng.standard_normal(int(rng.integers(1, 6))).astype(np.float32)][kind]()
    out = {}
    for k in range(int(rng.integers(1, 6))):
        if depth < 2 and rng.integers(3) == 0:
            out[f'k{k}'] = _random_properties(rng, depth + 1)
        elif rng.integers(6) == 0:
            out[f'k{k}'] = [_random_properties(rng, 2) for _ in range(int(rng.integers(1, 3)))]
        else:
            out[f'k{k}'] = leaf()
    return out


@pytest.mark.parametrize('serializer', SERIALIZERS)
@pytest.mark.parametrize('seed', range(6))
def test_random_collections_round_trip(tmpdir, serializer, seed):
    """random collections (ragged frame counts incl. one frame, float32 / float64 data, 1-D and 2-D
    times, nested properties with arrays, numpy scalars, lists, unicode) survive every file format"""
    rng = np.random.default_rng(100 + seed)
    coll = FeaturesCollection()
    cols = int(rng.integers(1, 20))
    for i in range(int(rng.integers(1, 6))):
        n = int(rng.integers(1, 60))
        data = rng.standard_normal((n, cols)).astype(rng.choice([np.float32, np.float64]))
        start = np.arange(n) * 0.01
        times = start if rng.integers(2) else np.stack([start, start + 0.025], axis=1)
        # (the .mat format squeezes arrays and turns lists into arrays - in the reference too: it gets
        # the kind of properties the processors really write)
        props = ({'mfcc': {'dither': 0.0, 'window_type': 'povey', 'snip_edges': True, 'num_ceps': 13},
                  'pipeline': [{'name': 'mfcc', 'columns': [0, cols - 1]}]}
                 if serializer is serializers.MatlabSerializer else _random_properties(rng))
        coll[f'item-{i}-ü'] = Features(data, times, properties=props)
    name = str(tmpdir.join(_name(serializer)))
    coll.save(name, serializer=serializer.__name__.replace('Serializer', '').lower())
    loaded = FeaturesCollection.load(name, serializer=serializer.__name__.replace('Serializer', '').lower())
    assert list(sorted(loaded.keys())) == list(sorted(coll.keys()))
    for k in coll:
        assert loaded[k].shape == coll[k].shape, k
        if serializer is not serializers.MatlabSerializer:
            assert loaded[k].dtype == coll[k].dtype, k
        assert np.array_equal(loaded[k].data, coll[k].data), k
        assert np.array_equal(loaded[k].times, coll[k].times) and loaded[k].times.shape == coll[k].times.shape
        if serializer is serializers.MatlabSerializer:  # (.mat holds doubles)
            assert loaded[k].is_close(coll[k].copy(dtype=np.float64))
            continue
        assert loaded[k] == coll[k], (k, loaded[k].properties, coll[k].properties)
