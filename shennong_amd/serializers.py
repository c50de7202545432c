"""Features files: numpy ``.npz`` and Kaldi ``.ark`` (+ ``.scp``), nothing else.

At GPU rates the writer decides the wall time of a corpus run (the reference reports 2:30 to write
38 h of MFCC as npz, features_collection.py:19-26), so the only formats kept are the two the hot path
needs: one self-contained ``.npz`` per collection, and the Kaldi binary archive, which can be
appended to batch by batch (`KaldiStreamWriter`) while ``pipeline.extract_features_streamed`` runs.
Both layouts are the reference's own (reference serializers.py:224-247 and :392-505), so files stay
interchangeable with it; the matlab / pickle / csv / h5features / json layouts of the reference are
out of scope here.

Kaldi table entry ([KALDI-UPSTREAM] util/kaldi-holder-inl.h, matrix/kaldi-matrix.cc Write):
``<key> \\0B`` + ``DM `` (double) or ``FM `` (float) + ``\\4<int32 rows>\\4<int32 cols>`` + row-major
values.  A collection is three files: ``<root>.ark`` (data), ``<root>.times.ark`` and
``<root>.properties.json`` (properties + the original dtypes; numpy arrays inside the properties use
the ``{"__ndarray__": ..., "dtype": ..., "shape": ...}`` encoding the reference's JSON writer uses).
"""

import copy
import json
import os
import struct

import numpy as np

from shennong_amd.features import Features


# ---- the two formats ---------------------------------------------------------------------------------
def _format_of(filename, serializer):
    """'numpy' or 'kaldi', from the explicit name or the file extension"""
    by_extension = {'.npz': 'numpy', '.ark': 'kaldi'}
    if serializer is not None:
        if serializer not in by_extension.values():
            raise ValueError(
                f'invalid serializer {serializer}, must be in {list(by_extension.values())}')
        return serializer
    ext = os.path.splitext(str(filename))[1]
    if ext not in by_extension:
        raise ValueError(f'invalid extension {ext}, must be in {list(by_extension)}')
    return by_extension[ext]


def save(features, filename, serializer=None, with_properties=True, log=None, **kwargs):
    """Writes a FeaturesCollection; IOError if the file exists, ValueError if the collection is not
    valid.  `compress` (numpy, default True), `scp` / `double` (kaldi)."""
    filename = str(filename)
    fmt = _format_of(filename, serializer)
    if type(features).__name__ != 'FeaturesCollection':
        raise ValueError(
            f'features must be FeaturesCollection but are {type(features).__name__}')
    if os.path.isfile(filename):
        raise IOError(f'file already exists: {filename}')
    if not features.is_valid():
        raise ValueError('features are not valid')
    if log:
        log.info('writing %s', filename)
    if fmt == 'numpy':
        entries = {name: f._to_dict(with_properties=with_properties) for name, f in features.items()}
        write = np.savez_compressed if kwargs.get('compress', True) is True else np.savez
        with open(filename, 'wb') as stream:
            write(stream, features=entries, allow_pickle=True)
    else:
        with KaldiStreamWriter(filename, scp=kwargs.get('scp', False),
                               with_properties=with_properties,
                               double=kwargs.get('double', True)) as writer:
            writer.write(features)


def load(cls, filename, serializer=None, log=None):
    """Reads a collection back as `cls` (FeaturesCollection)"""
    filename = str(filename)
    fmt = _format_of(filename, serializer)
    for test, problem in ((os.path.isfile, 'found'), (lambda f: os.access(f, os.R_OK), 'readable')):
        if not test(filename):
            raise IOError(f'file not {problem}: {filename}')
    if log:
        log.info('loading %s', filename)
    if fmt == 'numpy':
        with open(filename, 'rb') as stream:
            stored = np.load(stream, allow_pickle=True)['features'].item()
        features = cls((name, Features._from_dict(entry, validate=False))
                       for name, entry in stored.items())
    else:
        features = cls(_read_kaldi_collection(filename))
    if not features.is_valid():  # pragma: nocover
        raise ValueError(f'features not valid in "{filename}"')
    return features


# ---- properties JSON -----------------------------------------------------------------------------------
class _ArrayEncoder(json.JSONEncoder):
    def default(self, obj):
        if isinstance(obj, np.ndarray):
            return {'__ndarray__': obj.tolist(), 'dtype': str(obj.dtype),
                    'shape': list(obj.shape), 'Corder': True}
        if isinstance(obj, np.generic):
            return obj.item()
        return super().default(obj)


def _decode_arrays(dct):
    if '__ndarray__' in dct:
        return np.asarray(dct['__ndarray__'], dtype=dct.get('dtype')).reshape(dct.get('shape', -1))
    return dct


# ---- Kaldi binary tables -----------------------------------------------------------------------------
def _write_matrix(stream, key, mat, double):
    """Appends one table entry; returns the offset an scp line points to (just after ``<key> ``)"""
    mat = np.ascontiguousarray(mat, dtype=np.float64 if double else np.float32)
    stream.write(key.encode('utf-8') + b' ')
    offset = stream.tell()
    stream.write(b'\0BDM ' if double else b'\0BFM ')
    stream.write(b'\4' + struct.pack('<i', mat.shape[0]) + b'\4' + struct.pack('<i', mat.shape[1]))
    stream.write(mat.tobytes())
    return offset


def _read_table(ark):
    """key -> float64 matrix of a binary archive of double or float matrices"""
    with open(ark, 'rb') as stream:
        blob = stream.read()
    out, pos = {}, 0
    while pos < len(blob):
        end = blob.index(b' ', pos)
        key, pos = blob[pos:end].decode('utf-8'), end + 1
        kind = blob[pos:pos + 5]
        if kind not in (b'\0BDM ', b'\0BFM '):
            raise ValueError(f'{ark}: not a binary Kaldi archive of float or double matrices')
        dtype = np.dtype('<f8' if kind == b'\0BDM ' else '<f4')
        rows, cols = struct.unpack('<xixi', blob[pos + 5:pos + 15])
        pos += 15
        out[key] = np.frombuffer(blob, dtype=dtype, count=rows * cols, offset=pos).reshape(
            rows, cols).astype(np.float64)
        pos += rows * cols * dtype.itemsize
    return out


def _read_kaldi_collection(filename):
    root, ext = os.path.splitext(filename)
    if ext != '.ark':
        raise ValueError(
            'when saving to Kaldi ark format, the file extension must be '
            '".ark", it is "{}"'.format(ext))
    sidecars = {'properties': root + '.properties.json', 'times': root + '.times.ark'}
    for name in sidecars.values():
        if not os.path.isfile(name):
            raise IOError('file not found: {}'.format(name))
    with open(sidecars['properties'], 'r', encoding='utf-8') as stream:
        properties = json.loads(stream.read(), object_hook=_decode_arrays)
    times, data = _read_table(sidecars['times']), _read_table(filename)
    if properties.keys() != data.keys():
        raise ValueError('invalid features: items differ in data and properties')
    if times.keys() != data.keys():
        raise ValueError('invalid features: items differ in data and times')
    for key in data:
        props = properties[key]
        t = times[key]
        # 1-D times were written as one row; a [1, 2] times of a single-frame item stays 2-D
        if t.shape[0] == 1 and not (data[key].shape[0] == 1 and t.shape[1] == 2):
            t = t.reshape(t.shape[1])
        yield key, Features(
            data[key].astype(props['__dtype_data__']), t.astype(props['__dtype_times__']),
            properties={k: v for k, v in props.items() if '__dtype_' not in k}, validate=False)


class KaldiStreamWriter:
    """Incremental writer of the Kaldi layout (``<root>.ark``, ``<root>.times.ark``, optional
    ``.scp`` indexes, ``<root>.properties.json`` at close) for features that are produced batch by
    batch (pipeline.extract_features_streamed): the archives are appended to as the batches arrive,
    so the corpus never sits in host memory.  ``FeaturesCollection.save('x.ark')`` is one `write` of
    the whole collection through this class; ``double=False`` writes the data as Kaldi float
    matrices (half the bytes, float32 features lose nothing; the reference writes doubles).

    >>> with KaldiStreamWriter('corpus.ark', scp=True) as writer:       # doctest: +SKIP
    ...     extract_features_streamed(config, utterances, writer.write)
    """
    def __init__(self, filename, scp=False, with_properties=True, double=True, log=None):
        root, ext = os.path.splitext(str(filename))
        if ext != '.ark':
            raise ValueError(
                'when saving to Kaldi ark format, the file extension must be '
                '".ark", it is "{}"'.format(ext))
        self._root, self._double, self._with_properties, self._log = root, double, with_properties, log
        names = [root + '.ark', root + '.times.ark', root + '.properties.json']
        if scp:
            names += [root + '.scp', root + '.times.scp']
        for name in names:
            if os.path.exists(name):
                raise IOError('file already exists: {}'.format(name))
        self._properties = {}
        self._data = open(root + '.ark', 'wb')
        self._times = open(root + '.times.ark', 'wb')
        self._data_scp = open(root + '.scp', 'w', encoding='utf-8') if scp else None
        self._times_scp = open(root + '.times.scp', 'w', encoding='utf-8') if scp else None

    def write(self, features):
        """Appends the items of a FeaturesCollection (or any ``name -> Features`` mapping)"""
        if self._data is None:
            raise ValueError('writer is closed')
        for key, feat in features.items():
            if key in self._properties:
                raise ValueError('item already written: {}'.format(key))
            offset = _write_matrix(self._data, key, feat.data, self._double)
            if self._data_scp:
                self._data_scp.write(f'{key} {self._root}.ark:{offset}\n')
            offset = _write_matrix(self._times, key, np.atleast_2d(feat.times), True)
            if self._times_scp:
                self._times_scp.write(f'{key} {self._root}.times.ark:{offset}\n')
            props = copy.deepcopy(feat.properties) if self._with_properties else {}
            props['__dtype_data__'] = str(feat.dtype)
            props['__dtype_times__'] = str(feat.times.dtype)
            self._properties[key] = props

    def close(self):
        if self._data is None:
            return
        for stream in (self._data, self._times, self._data_scp, self._times_scp):
            if stream is not None:
                stream.close()
        self._data = None
        with open(self._root + '.properties.json', 'wt', encoding='utf-8') as stream:
            stream.write(json.dumps(self._properties, indent=4, cls=_ArrayEncoder, ensure_ascii=False))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
