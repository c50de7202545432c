"""Saves and loads features collections to/from various file formats

Same classes, entry points, file layouts and error behaviour as reference shennong/serializers.py
(``supported_extensions``:20-36, ``get_serializer``:58-109, ``FeaturesSerializer``:112-221, numpy
:224-247, matlab :250-330, pickle :333-351, kaldi :392-505, csv :508-600), without its third-party
bindings: Kaldi's binary archives (`DM` double matrices, optional scp index) are written and read
directly, and the properties JSON uses json_tricks' array encoding (``{"__ndarray__": ..., "dtype": ...,
"shape": ...}``) so that files stay readable by the reference.  h5features is not available here.
"""

import abc
import copy
import copyreg
import json
import os
import pickle
import struct

import numpy as np
import scipy.io

from shennong_amd.features import Features
from shennong_amd.utils import array2list


def supported_extensions():
    """File extensions mapped to their serializer class"""
    return {
        '.npz': NumpySerializer,
        '.mat': MatlabSerializer,
        '.pkl': PickleSerializer,
        '.ark': KaldiSerializer,
        '': CsvSerializer}


def supported_serializers():
    """Serializers names mapped to their class"""
    return {
        'numpy': NumpySerializer,
        'matlab': MatlabSerializer,
        'pickle': PickleSerializer,
        'kaldi': KaldiSerializer,
        'csv': CsvSerializer}


def get_serializer(cls, filename, log, serializer=None):
    """Returns the file serializer from filename extension or serializer name

    Raises
    ------
    ValueError
        If the serializer class cannot be guessed, or if `cls` is not FeaturesCollection
    """
    if cls.__name__ != 'FeaturesCollection':
        raise ValueError(
            'The `cls` parameter must be shennong.features.FeaturesCollection')
    if serializer is None:
        ext = os.path.splitext(str(filename))[1]
        try:
            serializer = supported_extensions()[ext]
        except KeyError:
            raise ValueError(
                'invalid extension {}, must be in {}'.format(
                    ext, list(supported_extensions().keys()))) from None
    else:
        try:
            serializer = supported_serializers()[serializer]
        except KeyError:
            raise ValueError(
                'invalid serializer {}, must be in {}'.format(
                    serializer, list(supported_serializers().keys()))) from None
    return serializer(cls, str(filename), log)


# ---- JSON with numpy support (json_tricks-compatible encoding) ---------------------------------------
class _NumpyEncoder(json.JSONEncoder):
    def default(self, obj):
        if isinstance(obj, np.ndarray):
            return {'__ndarray__': obj.tolist(), 'dtype': str(obj.dtype),
                    'shape': list(obj.shape), 'Corder': True}
        if isinstance(obj, np.generic):
            return obj.item()
        return super().default(obj)


def _json_hook(dct):
    if '__ndarray__' in dct:
        return np.asarray(dct['__ndarray__'], dtype=dct.get('dtype')).reshape(
            dct.get('shape', -1))
    return dct


def _json_dumps(data):
    return json.dumps(data, indent=4, cls=_NumpyEncoder, ensure_ascii=False)


def _json_loads(text):
    return json.loads(text, object_hook=_json_hook)


class FeaturesSerializer(metaclass=abc.ABCMeta):
    """Base class of a features file serializer"""
    def __init__(self, cls, filename, log):
        self._features_collection = cls
        self._filename = filename
        self._log = log

    @property
    def filename(self):
        """Name of the file to read or write"""
        return self._filename

    @abc.abstractmethod
    def _save(self, features, with_properties):  # pragma: nocover
        pass

    def _check_save(self):
        if os.path.isfile(self.filename):
            raise IOError(f'file already exists: {self.filename}')

    def save(self, features, with_properties=True, **kwargs):
        """Saves a collection of `features` to a file

        Raises IOError if the output file already exists, ValueError if the features are not a
        valid FeaturesCollection."""
        self._check_save()
        if not isinstance(features, self._features_collection):
            raise ValueError(
                'features must be {} but are {}'.format(
                    self._features_collection.__name__,
                    features.__class__.__name__))
        if not features.is_valid():
            raise ValueError('features are not valid')
        self._save(features, with_properties, **kwargs)

    @abc.abstractmethod
    def _load(self):  # pragma: nocover
        pass

    def _check_load(self):
        if not os.path.isfile(self.filename):
            raise IOError(f'file not found: {self.filename}')
        if not os.access(self.filename, os.R_OK):
            raise IOError(f'file not readable: {self.filename}')

    def load(self, **kwargs):
        """Returns the collection of features stored in the file

        Raises IOError if the file does not exist or cannot be read, ValueError if the features
        cannot be loaded or are not valid."""
        self._check_load()
        features = self._load(**kwargs)
        if not features.is_valid():  # pragma: nocover
            raise ValueError(f'features not valid in "{self.filename}"')
        return features


class NumpySerializer(FeaturesSerializer):
    """Saves and loads features to/from the numpy '.npz' format"""
    def _save(self, features, with_properties, compress=True):
        self._log.info('writing %s', self.filename)
        data = {k: v._to_dict(with_properties=with_properties)
                for k, v in features.items()}
        save = np.savez_compressed if compress is True else np.savez
        with open(self.filename, 'wb') as stream:
            save(stream, features=data, allow_pickle=True)

    def _load(self):
        self._log.info('loading %s', self.filename)
        with open(self.filename, 'rb') as stream:
            data = np.load(stream, allow_pickle=True)['features'].tolist()
        features = self._features_collection()
        for k, v in data.items():
            features[k] = Features._from_dict(v, validate=False)
        return features


class MatlabSerializer(FeaturesSerializer):
    """Saves and loads features to/from the matlab '.mat' format"""
    def _save(self, features, with_properties, compress=True):
        self._log.info('writing %s', self.filename)
        data = {k: v._to_dict(with_properties=with_properties)
                for k, v in features.items()}
        scipy.io.savemat(
            self.filename, data, long_field_names=True,
            appendmat=False, do_compression=compress)

    def _load(self):
        self._log.info('loading %s', self.filename)
        data = self._check_keys(scipy.io.loadmat(
            self.filename, appendmat=False, squeeze_me=True,
            mat_dtype=True, struct_as_record=False))
        features = self._features_collection()
        for k, v in data.items():
            if k not in ('__header__', '__version__', '__globals__'):
                mat, times = self._unsqueeze(v['data'], v['times'])
                if 'properties' in v:
                    features[k] = Features(
                        mat, times,
                        self._make_list(self._check_keys(v['properties'])),
                        validate=False)
                else:
                    features[k] = Features(mat, times, validate=False)
        return features

    @staticmethod
    def _unsqueeze(data, times):
        """`loadmat(squeeze_me=True)` (needed for the properties, as in the reference) also collapses
        a single frame or a single column of the matrices: give them their two dimensions back"""
        data, times = np.asarray(data), np.asarray(times)
        if data.ndim == 2:
            if data.shape[0] == 1 and times.ndim < 2:
                times = times.reshape((1, 2)) if times.size == 2 else times.reshape((1,))
            return data, times
        if data.ndim == 0:
            return data.reshape((1, 1)), (times.reshape((1, 2)) if times.size == 2 else times.reshape((1,)))
        if times.ndim == 2:  # several frames of one column
            return data.reshape((times.shape[0], -1)), times
        if times.ndim == 0:  # one frame, 1-D times
            return data.reshape((1, -1)), times.reshape((1,))
        if times.shape[0] == data.shape[0] and times.shape[0] != 2:  # one column, 1-D times
            return data.reshape((-1, 1)), times
        if times.shape[0] == 2:  # one frame, (start, stop) times
            return data.reshape((1, -1)), times.reshape((1, 2))
        return data.reshape((-1, 1)), times

    @staticmethod
    def _is_struct(obj):
        return obj.__class__.__name__ == 'mat_struct'

    @classmethod
    def _check_keys(cls, data):
        for key in data:
            if cls._is_struct(data[key]):
                data[key] = cls._todict(data[key])
            elif isinstance(data[key], (list, np.ndarray)) and any(
                    cls._is_struct(d) for d in np.atleast_1d(data[key]).ravel()):
                data[key] = [cls._todict(dd) for dd in data[key]]
        return data

    @classmethod
    def _todict(cls, matobj):
        data = {}
        for strg in matobj._fieldnames:
            elem = matobj.__dict__[strg]
            if cls._is_struct(elem):
                data[strg] = cls._todict(elem)
            elif isinstance(elem, np.ndarray) and elem.dtype == object and any(
                    cls._is_struct(d) for d in elem.ravel()):
                data[strg] = [cls._todict(d) for d in elem.ravel()]
            else:
                data[strg] = elem
        return data

    @staticmethod
    def _make_list(properties):
        if 'pipeline' in properties:
            # the matlab format collapses a list of a single element into that element
            if isinstance(properties['pipeline'], list):
                properties['pipeline'] = [
                    array2list(p) for p in properties['pipeline']]
            else:
                properties['pipeline'] = [array2list(properties['pipeline'])]
        return properties


class _NoPropertiesPickler(pickle.Pickler):
    """Implements the with_properties=False for PickleSerializer"""
    dispatch_table = copyreg.dispatch_table.copy()
    dispatch_table[Features] = lambda obj: (
        obj.__class__, (obj.data, obj.times, None, False))


class PickleSerializer(FeaturesSerializer):
    """Saves and loads features to/from the Python pickle format"""
    def _save(self, features, with_properties):
        self._log.info('writing %s', self.filename)
        pickler = pickle.Pickler if with_properties else _NoPropertiesPickler
        with open(self.filename, 'wb') as stream:
            pickler(stream).dump(features)

    def _load(self):
        self._log.info('loading %s', self.filename)
        with open(self.filename, 'rb') as stream:
            return pickle.load(stream)


# ---- Kaldi binary archives ------------------------------------------------------------------------------
def _write_kaldi_matrix(stream, key, mat, double=True):
    """One table entry; returns the offset an scp line points to (just after ``<key> ``).  `double`:
    a Kaldi double matrix (``DM``, what the reference writes) or a float matrix (``FM``, half the
    bytes; Kaldi tools and this module read both)"""
    mat = np.ascontiguousarray(mat, dtype=np.float64 if double else np.float32)
    stream.write(key.encode('utf-8') + b' ')
    offset = stream.tell()
    stream.write(b'\0BDM ' if double else b'\0BFM ')
    stream.write(b'\4' + struct.pack('<i', mat.shape[0]))
    stream.write(b'\4' + struct.pack('<i', mat.shape[1]))
    stream.write(mat.tobytes())
    return offset


def _write_kaldi_ark(ark, matrices, scp=None):
    """Binary table of double matrices: ``<key> \\0B DM \\4<rows>\\4<cols><float64 row-major>``
    ([KALDI-UPSTREAM] util/kaldi-holder-inl.h, matrix/kaldi-matrix.cc Write)"""
    index = []
    with open(ark, 'wb') as stream:
        for key, mat in matrices.items():
            index.append((key, _write_kaldi_matrix(stream, key, mat)))
    if scp:
        with open(scp, 'w', encoding='utf-8') as stream:
            for key, offset in index:
                stream.write(f'{key} {ark}:{offset}\n')


def _read_kaldi_ark(ark):
    out = {}
    with open(ark, 'rb') as stream:
        blob = stream.read()
    pos = 0
    while pos < len(blob):
        end = blob.index(b' ', pos)
        key = blob[pos:end].decode('utf-8')
        pos = end + 1
        if blob[pos:pos + 2] != b'\0B':
            raise ValueError(f'{ark}: not a binary Kaldi archive')
        pos += 2
        token_end = blob.index(b' ', pos)
        token = blob[pos:token_end]
        pos = token_end + 1
        if token not in (b'DM', b'FM'):
            raise ValueError(f'{ark}: unsupported Kaldi object {token!r}')
        dtype = np.float64 if token == b'DM' else np.float32
        rows = struct.unpack('<i', blob[pos + 1:pos + 5])[0]
        cols = struct.unpack('<i', blob[pos + 6:pos + 10])[0]
        pos += 10
        nbytes = rows * cols * np.dtype(dtype).itemsize
        out[key] = np.frombuffer(blob, dtype=dtype, count=rows * cols, offset=pos).reshape(
            (rows, cols)).astype(np.float64)
        pos += nbytes
    return out


class KaldiSerializer(FeaturesSerializer):
    """Saves and loads features to/from the Kaldi ark/scp format"""
    def __init__(self, cls, filename, log):
        super().__init__(cls, filename, log=log)
        filename_split = os.path.splitext(self.filename)
        if filename_split[1] != '.ark':
            raise ValueError(
                'when saving to Kaldi ark format, the file extension must be '
                '".ark", it is "{}"'.format(filename_split[1]))
        self._fileroot = filename_split[0]

    def _save(self, features, with_properties, scp=False):
        for suffix, get in (('', lambda v: v.data),
                            ('.times', lambda v: np.atleast_2d(v.times).copy())):
            ark = self._fileroot + suffix + '.ark'
            scp_file = self._fileroot + suffix + '.scp' if scp else None
            if scp:
                self._log.info('writing %s and %s', ark, scp_file)
            else:
                self._log.info('writing %s', ark)
            _write_kaldi_ark(ark, {k: get(v) for k, v in features.items()}, scp_file)

        # the matrices are written as doubles: the original dtypes go with the properties
        filename = self._fileroot + '.properties.json'
        self._log.info('writing %s', filename)
        if with_properties:
            data = {k: copy.deepcopy(v.properties) for k, v in features.items()}
        else:
            data = {k: {} for k in features}
        for k in data:
            data[k]['__dtype_data__'] = str(features[k].dtype)
            data[k]['__dtype_times__'] = str(features[k].times.dtype)
        with open(filename, 'wt', encoding='utf-8') as stream:
            stream.write(_json_dumps(data))

    def _load(self):
        filename = self._fileroot + '.properties.json'
        self._log.info('loading %s', filename)
        if not os.path.isfile(filename):
            raise IOError('file not found: {}'.format(filename))
        with open(filename, 'r', encoding='utf-8') as stream:
            properties = _json_loads(stream.read())

        ark = self._fileroot + '.times.ark'
        self._log.info('loading %s', ark)
        if not os.path.isfile(ark):
            raise IOError('file not found: {}'.format(ark))
        times = _read_kaldi_ark(ark)

        ark = self._fileroot + '.ark'
        self._log.info('loading %s', ark)
        data = _read_kaldi_ark(ark)

        # 1-D times were written as one row: back to 1-D (reference serializers.py does this for
        # every one-row matrix, which breaks the [1, 2] times of a single-frame item: those are told
        # apart by the number of frames of the data)
        for key, value in times.items():
            single_frame = (key in data and data[key].shape[0] == 1 and value.shape[1] == 2)
            if value.shape[0] == 1 and not single_frame:
                times[key] = value.reshape((value.shape[1]))

        if properties.keys() != data.keys():
            raise ValueError(
                'invalid features: items differ in data and properties')
        if times.keys() != data.keys():
            raise ValueError(
                'invalid features: items differ in data and times')
        return self._features_collection(
            **{k: Features(
                data[k].astype(properties[k]['__dtype_data__']),
                times[k].astype(properties[k]['__dtype_times__']),
                properties={
                    k: p for k, p in properties[k].items()
                    if '__dtype_' not in k},
                validate=False)
               for k in data.keys()})


class KaldiStreamWriter:
    """Incremental writer of the Kaldi layout above (``<root>.ark``, ``<root>.times.ark``,
    optional ``.scp`` indexes, ``<root>.properties.json`` at close) for features that are produced
    batch by batch (pipeline.extract_features_streamed): the archives are appended to as the
    batches arrive, so the corpus never sits in host memory.  What it writes loads back with
    ``FeaturesCollection.load(filename)`` / `KaldiSerializer` and is byte-identical to
    ``FeaturesCollection.save`` of the same items in the same order (``double=False`` writes the data
    as Kaldi float matrices instead: half the bytes, float32 features lose nothing).

    >>> with KaldiStreamWriter('corpus.ark', scp=True) as writer:       # doctest: +SKIP
    ...     extract_features_streamed(config, utterances, writer.write)
    """
    def __init__(self, filename, scp=False, with_properties=True, double=True, log=None):
        root, ext = os.path.splitext(filename)
        self._double = double  # False: float32 data matrices (times stay double)
        if ext != '.ark':
            raise ValueError(
                'when saving to Kaldi ark format, the file extension must be '
                '".ark", it is "{}"'.format(ext))
        self._root = root
        self._with_properties = with_properties
        self._log = log
        self._properties = {}
        self._files = [root + '.ark', root + '.times.ark', root + '.properties.json']
        if scp:
            self._files += [root + '.scp', root + '.times.scp']
        for name in self._files:
            if os.path.exists(name):
                raise IOError('file already exists: {}'.format(name))
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
            offset = _write_kaldi_matrix(self._data, key, feat.data, double=self._double)
            if self._data_scp:
                self._data_scp.write(f'{key} {self._root}.ark:{offset}\n')
            offset = _write_kaldi_matrix(self._times, key, np.atleast_2d(feat.times))
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
            stream.write(_json_dumps(self._properties))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class CsvSerializer(FeaturesSerializer):
    """Saves and loads features to/from the CSV format (one csv/json pair per item in a directory)"""
    def _check_load(self):
        if not os.path.isdir(self.filename):
            raise IOError(f'directory not found: {self.filename}')

    def _check_save(self):
        if os.path.exists(self.filename):
            raise IOError(f'already exists: {self.filename}')

    def _save(self, features, with_properties):
        os.makedirs(self.filename)
        self._log.info('writing directory "%s"', self.filename)
        for name, feat in features.items():
            csv_file = os.path.join(self.filename, name + '.csv')
            self._log.debug('writing %s', csv_file)
            np.savetxt(
                csv_file,
                np.hstack((
                    feat.times.reshape((feat.nframes, 1))
                    if feat.times.ndim == 1 else feat.times,
                    feat.data)),
                header=(
                    f'data_dtype = {feat.dtype}, '
                    f'times_dtype = {feat.times.dtype}, '
                    f'features_ndims = {feat.ndims}'),
                comments='# ')
            if with_properties and feat.properties:
                json_file = os.path.join(self.filename, name + '.json')
                self._log.debug('writing %s', json_file)
                with open(json_file, 'wt', encoding='utf-8') as stream:
                    stream.write(_json_dumps(feat.properties))

    @staticmethod
    def _parse_header(csv_file):
        with open(csv_file, 'r', encoding='utf-8') as stream:
            header = stream.readline().strip()
        if not header or header[0] != '#':
            raise ValueError(f'failed to parse header from {csv_file}')
        header = header.split(', ')
        try:
            data_dtype = np.dtype(header[0].split('= ')[1])
            times_dtype = np.dtype(header[1].split('= ')[1])
            ndims = int(header[2].split('= ')[1])
        except (IndexError, TypeError):
            raise ValueError(f'failed to parse header from {csv_file}') from None
        return data_dtype, times_dtype, ndims

    def _load(self):
        self._log.info('loading directory "%s"', self.filename)
        names = sorted(os.listdir(self.filename))
        csv_files = [os.path.join(self.filename, n) for n in names if n.endswith('.csv')]
        json_files = [os.path.join(self.filename, n) for n in names if n.endswith('.json')]
        features = self._features_collection()
        for csv in csv_files:
            self._log.debug('loading %s', csv)
            data_dtype, times_dtype, ndims = self._parse_header(csv)
            data = np.atleast_2d(np.loadtxt(csv))
            times = data[:, :data.shape[1] - ndims].astype(times_dtype)
            if times.shape[1] == 1:
                times = times.flatten()
            data = data[:, data.shape[1] - ndims:].astype(data_dtype)
            properties = {}
            json_file = csv[:-len('.csv')] + '.json'
            if json_file in json_files:
                self._log.debug('loading %s', json_file)
                with open(json_file, 'r', encoding='utf-8') as stream:
                    properties = dict(_json_loads(stream.read()))
            name = os.path.basename(csv)[:-len('.csv')]
            features[name] = Features(
                data, times, properties=properties, validate=False)
        return features
