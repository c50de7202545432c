"""Features files: numpy .npz and Kaldi .ark, one-shot and streamed, and the matlab /
pickle / csv interchange formats of the reference (reference test/test_serializers.py).  Host-only: the
features are a fixed random matrix shaped like the MFCCs of test.wav, no device call."""

import os

import numpy as np
import pytest

from shennong_amd import Features, FeaturesCollection
from shennong_amd import serializers
from shennong_amd.processor import MfccProcessor

FORMATS = [('numpy', 'feats.npz'), ('kaldi', 'feats.ark'), ('matlab', 'feats.mat'), ('pickle', 'feats.pkl'),
           ('csv', 'feats')]


@pytest.fixture(scope='module')
def mfcc():
    proc = MfccProcessor()
    data = np.random.default_rng(0).standard_normal((140, 13)).astype(np.float32)
    return Features(data, proc.times(140), properties=proc.get_properties(vtln_warp=1.0))


@pytest.fixture()
def mfcc_col(mfcc):
    return FeaturesCollection(mfcc=mfcc)


def test_format_lookup(tmp_path, mfcc_col):
    for args, message in (
            (('foo.spam', None), 'invalid extension .spam'),
            (('foo.h5f', None), 'h5features package, which is not installed'),
            (('foo.npz', 'spam'), 'invalid serializer spam'),
            (('foo.file', 'kaldi'), 'the file extension must be ".ark", it is ".file"')):
        with pytest.raises(ValueError, match=message):
            mfcc_col.save(str(tmp_path / args[0]), serializer=args[1])
    with pytest.raises(ValueError, match='features must be FeaturesCollection'):
        serializers.save({'a': 1}, str(tmp_path / 'x.npz'))
    assert not list(tmp_path.iterdir()), 'nothing is created by a refused save'


@pytest.mark.parametrize('fmt, name', FORMATS)
@pytest.mark.parametrize('with_properties', [True, False])
def test_round_trip(tmp_path, mfcc_col, fmt, name, with_properties):
    target = str(tmp_path / name)
    mfcc_col.save(target, with_properties=with_properties)
    with pytest.raises(IOError, match='already exists'):
        mfcc_col.save(target)
    back = FeaturesCollection.load(target)
    assert back.keys() == mfcc_col.keys()
    got, want = back['mfcc'], mfcc_col['mfcc']
    assert got.dtype == want.dtype and got.times.dtype == want.times.dtype
    assert np.array_equal(got.data, want.data) and np.array_equal(got.times, want.times)
    assert got.properties == (want.properties if with_properties else {})
    assert FeaturesCollection.load(target, serializer=fmt) == back


@pytest.mark.parametrize('fmt, name', FORMATS)
def test_bad_files(tmp_path, mfcc_col, fmt, name):
    with pytest.raises(IOError, match='directory not found' if fmt == 'csv' else 'file not found'):
        FeaturesCollection.load(str(tmp_path / name))
    invalid = FeaturesCollection(mfcc=Features(
        np.full((2, 2), np.nan, dtype=np.float32), np.arange(2, dtype=np.float64), validate=False))
    with pytest.raises(ValueError, match='features are not valid'):
        invalid.save(str(tmp_path / name))


def test_reference_entry_points(tmp_path, mfcc_col):
    """supported_extensions / supported_serializers / get_serializer of reference serializers.py:20-109"""
    assert list(serializers.supported_extensions()) == ['.npz', '.mat', '.pkl', '.h5f', '.ark', '']
    assert list(serializers.supported_serializers()) == ['numpy', 'matlab', 'pickle', 'h5features', 'kaldi', 'csv']
    with pytest.raises(ValueError, match='must be shennong.features.FeaturesCollection'):
        serializers.get_serializer(Features, 'x.npz', None)
    writer = serializers.get_serializer(FeaturesCollection, str(tmp_path / 'x.pkl'), None)
    assert writer.filename == str(tmp_path / 'x.pkl')
    writer.save(mfcc_col)
    assert serializers.get_serializer(FeaturesCollection, writer.filename, None, 'pickle').load() == mfcc_col


def test_shapes_survive_matlab_and_csv(tmp_path):
    """one-frame items, one-column items, 1-D and 2-D times: matlab squeezes vectors, csv is a flat table"""
    rng = np.random.default_rng(1)
    col = FeaturesCollection(
        one_frame=Features(rng.standard_normal((1, 3)).astype(np.float32), np.array([[0, 0.025]])),
        one_column=Features(rng.standard_normal((2, 1)), np.array([0.0, 0.01])),
        square=Features(rng.standard_normal((2, 2)), np.array([[0.0, 0.025], [0.01, 0.035]]),
                        properties={'pipeline': [{'name': 'mfcc', 'columns': [0, 1]}], 'mfcc': {'warp': 1.0}}))
    for name in ('shapes.mat', 'shapes'):
        col.save(str(tmp_path / name))
        back = FeaturesCollection.load(str(tmp_path / name))
        assert back == col
        for key in col:
            assert back[key].shape == col[key].shape and back[key].times.shape == col[key].times.shape


def test_kaldi_sidecars(tmp_path, mfcc_col):
    target = str(tmp_path / 'feats.ark')
    mfcc_col.save(target, scp=True)
    root = str(tmp_path / 'feats')
    for suffix in ('.ark', '.times.ark', '.scp', '.times.scp', '.properties.json'):
        assert os.path.isfile(root + suffix), suffix
    key, where = open(root + '.scp').read().split()
    ark, offset = where.rsplit(':', 1)
    assert key == 'mfcc' and ark == root + '.ark'
    with open(ark, 'rb') as stream:
        stream.seek(int(offset))
        assert stream.read(5) == b'\0BDM '
    os.remove(root + '.times.ark')
    with pytest.raises(IOError, match='file not found'):
        FeaturesCollection.load(target)


@pytest.mark.parametrize('scp', [False, True])
@pytest.mark.parametrize('with_properties', [False, True])
def test_stream_writer_equals_one_shot(tmp_path, scp, with_properties):
    """KaldiStreamWriter fed batch by batch writes byte for byte what save() writes in one go"""
    rng = np.random.default_rng(5)
    proc = MfccProcessor()
    items = FeaturesCollection()
    for i, n in enumerate((1, 7, 140, 3)):
        items[f'utt{i}'] = Features(rng.standard_normal((n, 13)).astype(np.float32), proc.times(n),
                                    properties=proc.get_properties(vtln_warp=1.0 + i))
    one_shot, streamed = str(tmp_path / 'a.ark'), str(tmp_path / 'b.ark')
    items.save(one_shot, scp=scp, with_properties=with_properties)
    names = list(items)
    with serializers.KaldiStreamWriter(streamed, scp=scp, with_properties=with_properties) as w:
        w.write({k: items[k] for k in names[:1]})
        w.write({k: items[k] for k in names[1:]})
        with pytest.raises(ValueError, match='item already written'):
            w.write({names[0]: items[names[0]]})
    for suffix in ('.ark', '.times.ark', '.properties.json') + (('.scp', '.times.scp') if scp else ()):
        a = open(one_shot[:-4] + suffix, 'rb').read()
        b = open(streamed[:-4] + suffix, 'rb').read()
        if suffix.endswith('.scp'):
            b = b.replace(b'b.', b'a.')
        assert a == b, suffix
    back = FeaturesCollection.load(streamed)
    assert back == items if with_properties else all(
        np.array_equal(back[k].data, items[k].data) and back[k].properties == {} for k in items)
    with pytest.raises(ValueError, match='writer is closed'):
        w.write({})
    with pytest.raises(IOError, match='already exists'):
        serializers.KaldiStreamWriter(streamed)
    with pytest.raises(ValueError, match='must be ".ark"'):
        serializers.KaldiStreamWriter(str(tmp_path / 'x.npz'))


def test_float_matrices_and_single_frames(tmp_path):
    proc = MfccProcessor()
    data = np.random.default_rng(1).standard_normal((1, 13)).astype(np.float32)
    one = FeaturesCollection(single=Features(data, proc.times(1)),
                             flat=Features(np.ones((3, 2), np.float32), np.arange(3, dtype=np.float64)))
    name = str(tmp_path / 'f.ark')
    with serializers.KaldiStreamWriter(name, double=False) as w:
        w.write(one)
    assert os.path.getsize(name) < 200
    back = FeaturesCollection.load(name)
    assert back == one and back['single'].times.shape == (1, 2) and back['flat'].times.shape == (3,)
