"""Features files: numpy ``.npz``, Kaldi ``.ark`` (+ ``.scp``), matlab ``.mat``, pickle ``.pkl`` and a
directory of ``.csv`` files - the formats of reference shennong/serializers.py that need no package
missing here (h5features is not installed and says so), with the reference's file layouts, so files stay
interchangeable with it, and its entry points (`supported_extensions`, `supported_serializers`,
`get_serializer(cls, filename, log, serializer)` -> an object with `save` / `load`).

At GPU rates the writer decides the wall time of a corpus run (the reference reports 2:30 to write
38 h of MFCC as npz, features_collection.py:19-26): the two formats on the hot path are one
self-contained ``.npz`` per collection and the Kaldi binary archive, which can be appended to batch by
batch (`KaldiStreamWriter`) while ``pipeline.extract_features_streamed`` runs; mat / pickle / csv are
plain host-side writers for interchange.

Kaldi table entry ([KALDI-UPSTREAM] util/kaldi-holder-inl.h, matrix/kaldi-matrix.cc Write):
``<key> \\0B`` + ``DM `` (double) or ``FM `` (float) + ``\\4<int32 rows>\\4<int32 cols>`` + row-major
values.  A collection is three files: ``<root>.ark`` (data), ``<root>.times.ark`` and
``<root>.properties.json`` (properties + the original dtypes; numpy arrays inside the properties use
the ``{"__ndarray__": ..., "dtype": ..., "shape": ...}`` encoding the reference's JSON writer uses).
"""

import copy
import json
import os
import pickle
import struct

import numpy as np

from shennong_amd.features import Features


# ---- one (writer, reader) pair per format ---------------------------------------------------------------
def _write_numpy(features, filename, with_properties, compress=True):
    entries = {name: f._to_dict(with_properties=with_properties) for name, f in features.items()}
    write = np.savez_compressed if compress is True else np.savez
    with open(filename, 'wb') as stream:
        write(stream, features=entries, allow_pickle=True)


def _read_numpy(cls, filename):
    with open(filename, 'rb') as stream:
        stored = np.load(stream, allow_pickle=True)['features'].item()
    return cls((name, Features._from_dict(entry, validate=False)) for name, entry in stored.items())


def _write_kaldi(features, filename, with_properties, scp=False, double=True):
    with KaldiStreamWriter(filename, scp=scp, with_properties=with_properties, double=double) as writer:
        writer.write(features)


def _read_kaldi(cls, filename):
    return cls(_read_kaldi_collection(filename))


def _write_matlab(features, filename, with_properties, compress=True):
    """one struct per item with the fields data / times (/ properties), reference serializers.py:250-264"""
    import scipy.io
    scipy.io.savemat(filename, {name: f._to_dict(with_properties=with_properties)
                                for name, f in features.items()},
                     long_field_names=True, appendmat=False, do_compression=compress)


def _read_matlab(cls, filename):
    import scipy.io
    # the shapes of data / times from an unsqueezed read (a one-frame item is a [1, D] matrix, not a vector),
    # the properties from a simplified one (nested dicts instead of struct arrays)
    shaped = scipy.io.loadmat(filename, appendmat=False, mat_dtype=True)
    stored = scipy.io.loadmat(filename, appendmat=False, mat_dtype=True, simplify_cells=True)

    def plain(value):
        # matlab keeps no lists and no scalars: cell arrays and 0-d / 1 x n arrays come back in their place
        if isinstance(value, dict):
            return {k: plain(v) for k, v in value.items()}
        if isinstance(value, (list, tuple)):
            return [plain(v) for v in value]
        if isinstance(value, np.ndarray):
            if value.dtype == object:
                return [plain(v) for v in value.tolist()]
            return value.item() if value.ndim == 0 else value
        return value
    out = cls()
    for name, entry in stored.items():
        if name.startswith('__'):
            continue
        properties = plain(entry.get('properties', {})) or {}
        stages = properties.get('pipeline')
        if stages is not None:
            # a list of ONE stage collapses into that stage, and its [first, last] columns into an array
            stages = stages if isinstance(stages, list) else [stages]
            properties['pipeline'] = [
                {k: (np.asarray(v).astype(int).tolist() if k == 'columns' else v) for k, v in st.items()}
                for st in stages]
        data, times = shaped[name]['data'][0, 0], shaped[name]['times'][0, 0]
        if times.shape == (1, data.shape[0]) and times.shape != (data.shape[0], 2):
            times = times[0]  # (a vector of T start times was written as a 1 x T row)
        out[name] = Features(data, times, properties, validate=False)
    return out


def _write_pickle(features, filename, with_properties):
    if not with_properties:
        features = type(features)((name, Features(f.data, f.times, validate=False))
                                  for name, f in features.items())
    with open(filename, 'wb') as stream:
        pickle.dump(features, stream)


def _read_pickle(cls, filename):
    with open(filename, 'rb') as stream:
        return pickle.load(stream)


def _write_csv(features, dirname, with_properties):
    """a directory with `<item>.csv` ([times | data], the dtypes and the width in a header line) and
    `<item>.json` (properties), reference serializers.py:508-547"""
    os.makedirs(dirname)
    for name, feat in features.items():
        times = feat.times.reshape(feat.nframes, -1)
        np.savetxt(os.path.join(dirname, name + '.csv'), np.hstack((times, feat.data)), comments='# ',
                   header=f'data_dtype = {feat.dtype}, times_dtype = {feat.times.dtype}, '
                          f'features_ndims = {feat.ndims}')
        if with_properties and feat.properties:
            with open(os.path.join(dirname, name + '.json'), 'wt', encoding='utf-8') as stream:
                stream.write(json.dumps(feat.properties, indent=4, cls=_ArrayEncoder))


def _read_csv(cls, dirname):
    out = cls()
    for entry in sorted(os.listdir(dirname)):
        if not entry.endswith('.csv'):
            continue
        path = os.path.join(dirname, entry)
        with open(path, 'r', encoding='utf-8') as stream:
            header = stream.readline().strip()
        try:
            fields = dict(part.split(' = ') for part in header.lstrip('# ').split(', '))
            data_dtype, times_dtype = np.dtype(fields['data_dtype']), np.dtype(fields['times_dtype'])
            ndims = int(fields['features_ndims'])
        except (KeyError, ValueError, TypeError):
            raise ValueError(f'failed to parse header from {path}') from None
        table = np.atleast_2d(np.loadtxt(path))
        times = table[:, :table.shape[1] - ndims].astype(times_dtype)
        properties = {}
        sidecar = path[:-4] + '.json'
        if os.path.isfile(sidecar):
            with open(sidecar, 'r', encoding='utf-8') as stream:
                properties = json.loads(stream.read(), object_hook=_decode_arrays)
        out[entry[:-4]] = Features(table[:, table.shape[1] - ndims:].astype(data_dtype),
                                   times[:, 0] if times.shape[1] == 1 else times, properties, validate=False)
    return out


def _no_h5features(*args, **kwargs):
    raise ValueError('the h5features format needs the h5features package, which is not installed here; '
                     'use .npz, .ark, .mat, .pkl or a csv directory')


# name -> (extension, writer, reader); '' = a directory
_FORMATS = {
    'numpy': ('.npz', _write_numpy, _read_numpy),
    'matlab': ('.mat', _write_matlab, _read_matlab),
    'pickle': ('.pkl', _write_pickle, _read_pickle),
    'h5features': ('.h5f', _no_h5features, _no_h5features),
    'kaldi': ('.ark', _write_kaldi, _read_kaldi),
    'csv': ('', _write_csv, _read_csv),
}


class FeaturesSerializer:
    """One features file in one format: `save(features, with_properties=True, **kwargs)` and `load()`
    (reference serializers.py:112-221; `compress` for numpy / matlab, `scp` / `double` for kaldi)"""
    def __init__(self, cls, filename, log, name):
        self._cls, self._filename, self._log, self.name = cls, str(filename), log, name
        _, self._write, self._read = _FORMATS[name]

    filename = property(lambda self: self._filename, doc='The file (csv: the directory) read or written')

    def save(self, features, with_properties=True, **kwargs):
        if type(features).__name__ != 'FeaturesCollection':
            raise ValueError(
                f'features must be FeaturesCollection but are {type(features).__name__}')
        if os.path.exists(self.filename) if self.name == 'csv' else os.path.isfile(self.filename):
            raise IOError(f'file already exists: {self.filename}')
        if not features.is_valid():
            raise ValueError('features are not valid')
        if self._log:
            self._log.info('writing %s', self.filename)
        self._write(features, self.filename, with_properties, **kwargs)

    def load(self, **kwargs):
        if self.name == 'csv':
            if not os.path.isdir(self.filename):
                raise IOError(f'directory not found: {self.filename}')
        else:
            for test, problem in ((os.path.isfile, 'found'), (lambda f: os.access(f, os.R_OK), 'readable')):
                if not test(self.filename):
                    raise IOError(f'file not {problem}: {self.filename}')
        if self._log:
            self._log.info('loading %s', self.filename)
        features = self._read(self._cls, self.filename, **kwargs)
        if not features.is_valid():  # pragma: nocover
            raise ValueError(f'features not valid in "{self.filename}"')
        return features


def supported_serializers():
    """format name -> a factory ``(cls, filename, log) -> FeaturesSerializer``"""
    def factory(name):
        return lambda cls, filename, log=None: FeaturesSerializer(cls, filename, log, name)
    return {name: factory(name) for name in _FORMATS}


def supported_extensions():
    """file extension ('' = a directory of csv files) -> the same factories"""
    by_name = supported_serializers()
    return {ext: by_name[name] for name, (ext, _, _) in _FORMATS.items()}


def get_serializer(cls, filename, log=None, serializer=None):
    """The serializer of `filename`, from its extension or from the format name `serializer`"""
    if cls.__name__ != 'FeaturesCollection':
        raise ValueError('The `cls` parameter must be shennong.features.FeaturesCollection')
    if serializer is None:
        ext = os.path.splitext(str(filename))[1]
        try:
            return supported_extensions()[ext](cls, filename, log)
        except KeyError:
            raise ValueError(
                f'invalid extension {ext}, must be in {list(supported_extensions())}') from None
    try:
        return supported_serializers()[serializer](cls, filename, log)
    except KeyError:
        raise ValueError(
            f'invalid serializer {serializer}, must be in {list(supported_serializers())}') from None


def save(features, filename, serializer=None, with_properties=True, log=None, **kwargs):
    """Writes a FeaturesCollection; IOError if the file exists, ValueError if the collection is not valid"""
    cls = type(features) if type(features).__name__ == 'FeaturesCollection' else _collection_class()
    get_serializer(cls, filename, log, serializer).save(features, with_properties=with_properties, **kwargs)


def load(cls, filename, serializer=None, log=None):
    """Reads a collection back as `cls` (FeaturesCollection)"""
    return get_serializer(cls, filename, log, serializer).load()


def _collection_class():
    from shennong_amd.features import FeaturesCollection
    return FeaturesCollection


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


def _json_default(obj):
    """numpy arrays and scalars inside properties (the callback form of _ArrayEncoder: with no indentation the
    C encoder of the json module stays in use)"""
    if isinstance(obj, np.ndarray):
        return {'__ndarray__': obj.tolist(), 'dtype': str(obj.dtype), 'shape': list(obj.shape), 'Corder': True}
    if isinstance(obj, np.generic):
        return obj.item()
    raise TypeError(f'Object of type {type(obj).__name__} is not JSON serializable')


def _dumps(obj):
    return json.dumps(obj, default=_json_default, ensure_ascii=False)


def _pwrite_all(fd, buffers, offset):
    """`buffers` one after the other at `offset` of `fd` (one positioned gather write); a short write is finished
    buffer by buffer"""
    total = sum(len(memoryview(b).cast('B')) for b in buffers)
    done = os.pwritev(fd, buffers, offset) if total else 0
    if done == total:
        return total
    pos = offset + done
    for buf in buffers:
        view = memoryview(buf).cast('B')
        if done >= len(view):
            done -= len(view)
            continue
        view, done = view[done:], 0
        while len(view):
            n = os.pwrite(fd, view, pos)
            view, pos = view[n:], pos + n
    return total


class KaldiStreamWriter:
    """Incremental writer of the Kaldi layout (``<root>.ark``, ``<root>.times.ark``, optional
    ``.scp`` indexes, ``<root>.properties.json``) for features that are produced batch by
    batch (pipeline.extract_features_streamed): the archives are appended to as the batches arrive,
    so the corpus never sits in host memory.  ``FeaturesCollection.save('x.ark')`` is one `write` of
    the whole collection through this class; ``double=False`` writes the data as Kaldi float
    matrices (half the bytes, float32 features lose nothing; the reference writes doubles).

    At GPU rates the writer is what a corpus run waits for (the reference: 2:30 min for
    38 h of MFCC, features_collection.py:19-26), so a batch is written as a batch (round 6): the records go out as
    gather writes of up to 512 records (`os.pwritev`: headers and rows, the float32 rows straight from the batch's
    block without a copy), times that the utterances of a batch share are not copied per utterance, and the
    properties are encoded as they arrive (one JSON object per item, the C encoder; the part a batch shares is
    encoded once per processing history, `Features._json_properties`) instead of being deep-copied, kept until
    `close` and dumped with indentation.  One 4 800-utterance batch of 123 columns (704 MB of rows): 0.27 s as
    float matrices, 0.42 s as doubles, against 0.66 / 0.78 s (`tools/profile_ark_writer.py`,
    `profiles/r06_ark_writer.txt`); the rest is the page cache (2.5-3.4 GB/s of one file: writing disjoint
    stretches from 2 - 8 threads gains nothing, the file's lock serialises them - measured, not kept).

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
        self._names = set()
        self._data = os.open(root + '.ark', os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o644)
        self._times = os.open(root + '.times.ark', os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o644)
        self._data_pos = self._times_pos = 0
        self._props = open(root + '.properties.json', 'wt', encoding='utf-8')
        self._props.write('{')
        self._data_scp = open(root + '.scp', 'w', encoding='utf-8') if scp else None
        self._times_scp = open(root + '.times.scp', 'w', encoding='utf-8') if scp else None

    @staticmethod
    def _records(items, arrays, dtype, start):
        """(header, array) per item, the scp offsets, the byte position behind the last record"""
        kind = b'\0BDM ' if dtype == np.float64 else b'\0BFM '
        records, offsets, pos = [], [], start
        for (key, _), mat in zip(items, arrays):
            name = key.encode('utf-8') + b' '
            head = name + kind + b'\4' + struct.pack('<i', mat.shape[0]) + b'\4' + struct.pack('<i', mat.shape[1])
            records.append((pos, head, mat))
            offsets.append(pos + len(name))
            pos += len(head) + mat.size * dtype.itemsize
        return records, offsets, pos

    @staticmethod
    def _flush(fd, records, dtype):
        """conversion (if any) and gather writes of up to 512 records"""
        for i in range(0, len(records), 512):
            part = records[i:i + 512]
            buffers = []
            for _, head, mat in part:
                buffers.append(head)
                if mat.size:
                    buffers.append(np.ascontiguousarray(mat, dtype=dtype))   # (no copy when it is that already)
            _pwrite_all(fd, buffers, part[0][0])

    def write(self, features):
        """Appends the items of a FeaturesCollection (or any ``name -> Features`` mapping)"""
        if self._data is None:
            raise ValueError('writer is closed')
        items = list(features.items())
        seen = set()
        for key, _ in items:
            if key in self._names or key in seen:
                raise ValueError('item already written: {}'.format(key))
            seen.add(key)
        dtype = np.dtype(np.float64 if self._double else np.float32)
        records, offsets, end = self._records(items, [feat.data for _, feat in items], dtype, self._data_pos)
        self._flush(self._data, records, dtype)
        self._data_pos = end
        if self._data_scp:
            self._data_scp.write(''.join(f'{key} {self._root}.ark:{off}\n' for (key, _), off in zip(items, offsets)))
        f64 = np.dtype(np.float64)
        times = [np.atleast_2d(feat._times_view()) for _, feat in items]
        records, offsets, end = self._records(items, times, f64, self._times_pos)
        self._flush(self._times, records, f64)
        self._times_pos = end
        if self._times_scp:
            self._times_scp.write(''.join(f'{key} {self._root}.times.ark:{off}\n'
                                          for (key, _), off in zip(items, offsets)))
        lines = []
        for key, feat in items:
            text = feat._json_properties(_dumps) if self._with_properties else '{}'
            tail = '"__dtype_data__": %s, "__dtype_times__": %s}' % (
                _dumps(str(feat.dtype)), _dumps(str(feat._times_view().dtype)))
            text = '{' + tail if text == '{}' else text[:-1] + ', ' + tail
            lines.append('%s: %s' % (_dumps(key), text))
        if lines:
            self._props.write((',\n' if self._names else '\n') + ',\n'.join(lines))
        self._names |= seen

    def close(self):
        if self._data is None:
            return
        for fd in (self._data, self._times):
            os.close(fd)
        for stream in (self._data_scp, self._times_scp):
            if stream is not None:
                stream.close()
        self._data = None
        self._props.write('\n}\n')
        self._props.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
