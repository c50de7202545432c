"""Features container: the output of every processor

Counterpart of reference shennong/features.py:62-437: a float matrix ``data [nframes, ndims]``, the
``times [nframes, 2]`` (start, stop) or ``[nframes]`` of its rows and a ``properties`` dictionary
that records how the matrix was made.  The per-frame Python loop of the reference's ``validate``
(features.py:342) is a vectorised check here; the data-independent half of ``concatenate`` is
shared with the device-resident pipeline.
"""

import collections

import numpy as np

from shennong_amd.logger import get_logger
from shennong_amd.utils import copy_properties, dict_equal


def _invalid(features):
    """Why `features` is not valid (first failing group of checks), or None"""
    data, times, properties = features.data, features.times, features.properties
    kinds = [message for wrong, message in (
        (not isinstance(data, np.ndarray), 'data must be a numpy array'),
        (not isinstance(times, np.ndarray), 'times must be a numpy array'),
        (not isinstance(properties, dict), 'properties must be a dictionnary')) if wrong]
    if kinds:
        return 'invalid features data types: ' + ', '.join(kinds)
    shapes = [message for wrong, message in (
        (data.ndim != 2, f'data dimension must be 2 but is {data.ndim}'),
        (times.ndim > 2, f'times dimension must be 1 or 2 but is {times.ndim}'),
        (times.ndim == 2 and times.shape[1] != 2,
         'times shape[1] must be 2, it is {}'.format(times.shape[1] if times.ndim == 2 else None)),
        (data.shape[:1] != times.shape[:1],
         'mismatch in number of frames: {} for data but {} for times'.format(
             data.shape[0] if data.ndim else None, times.shape[0] if times.ndim else None))) if wrong]
    if shapes:
        return 'invalid features dimensions: ' + ', '.join(shapes)
    # sorted rows: a stable sort (lexicographic for two columns) must be the identity
    order = np.argsort(times, kind='stable') if times.ndim == 1 else np.lexsort(times.T)
    if not np.array_equal(order, np.arange(data.shape[0])):
        return 'times is not sorted in increasing order'
    if not np.all(np.isfinite(data)):
        return 'data contains non-finite numbers (nan of infinity)'
    return None


class Features:
    """Features data with attached timestamps and properties"""
    # (no per-instance dictionary: a corpus run makes one of these per utterance)
    __slots__ = ('_data', '_times', '_properties', '_shared')

    def __init__(self, data, times, properties=None, validate=True):
        self._data, self._times = data, times
        self._properties = {} if properties is None else properties
        self._shared = None
        if validate is True:
            self.validate()

    @classmethod
    def _of_batch(cls, data, times, properties, extra=None):
        """One utterance of a batched launch: `times` and `properties` are objects SHARED by every
        utterance of the batch with the same frame count / the same history (plus this utterance's own
        `extra` entries).  The private copies every Features owns (reference features.py:62-168: both are
        plain attributes a caller may edit) are made when `times` / `properties` are first read - a corpus
        run that writes the matrices and never looks at them pays nothing per utterance."""
        self = cls.__new__(cls)
        self._data, self._times, self._properties = data, None, None
        self._shared = (times, properties, extra)
        return self

    data = property(lambda self: self._data, doc='The features matrix [nframes, ndims]')

    @property
    def times(self):
        """The times of the rows, [nframes, 2] or [nframes]"""
        if self._times is None:
            self._times = self._shared[0].copy()
            self._shared = (None,) + self._shared[1:]
        return self._times

    @property
    def properties(self):
        """How the features were made"""
        if self._properties is None:
            shared = self._shared[1]
            if type(shared) is not dict:   # (a pipeline history that makes its dictionary when first asked)
                shared = shared.properties
            self._properties = copy_properties(shared)
            extra = self._shared[2]
            if type(extra) is tuple:   # (function, arguments): this utterance's own entries, made when asked for
                self._properties.update(extra[0](*extra[1:]))
            elif extra:
                self._properties.update(copy_properties(extra))
            self._shared = (self._shared[0], None, None)   # (the batch's history is not kept alive by a read copy)
        return self._properties

    def _times_view(self):
        """The times without the private copy `times` makes (serializers: read only)"""
        return self._times if self._times is not None else self._shared[0]

    def _json_properties(self, dumps):
        """JSON text of `properties` (an object).  For an utterance of a batched pipeline run whose properties
        nobody has read, the part it shares with the batch is encoded once per processing history (and kept on
        it) and this utterance's own entries are appended: the text a writer needs without 150 000 private copies
        of dictionaries that hold statistics arrays."""
        if self._properties is None and type(self._shared[1]) is not dict:
            history, extra = self._shared[1], self._shared[2]
            text = history.json(dumps)
            if type(extra) is tuple:
                extra = extra[0](*extra[1:])
            if extra:
                own = dumps(extra)
                text = own if text == '{}' else text[:-1] + ', ' + own[1:]
            return text
        return dumps(self.properties)

    def __getstate__(self):
        """Pickling (the `.pkl` serializer, joblib / multiprocessing transport): the times and properties a
        batched launch shares between utterances are made this utterance's own first - the shared history
        holds closures and the whole batch's cache, neither of which can or should travel"""
        return {'_data': self._data, '_times': self.times, '_properties': self.properties,
                '_shared': None}

    def __setstate__(self, state):
        for name, value in state.items():
            setattr(self, name, value)

    dtype = property(lambda self: self.data.dtype)
    shape = property(lambda self: self.data.shape)
    nframes = property(lambda self: self.shape[0])
    ndims = property(lambda self: self.shape[1])

    # ---- validity, comparison ---------------------------------------------------------------------
    def validate(self):
        """Raises a ValueError if the features are not in a valid state"""
        reason = _invalid(self)
        if reason:
            raise ValueError(reason)

    def is_valid(self):
        return _invalid(self) is None

    def _same_frame(self, other):
        """Same shape, times and properties (what `==` and `is_close` both require)"""
        return (self.shape == other.shape and dict_equal(self.properties, other.properties)
                and np.array_equal(self.times, other.times))

    def __eq__(self, other):
        if self is other:
            return True
        return (self.dtype == other.dtype and self._same_frame(other)
                and bool(np.array_equal(self.data, other.data)))

    def is_close(self, other, rtol=1e-5, atol=1e-8):
        """True if data is allclose and shape / times / properties are equal"""
        if self is other:
            return True
        return self._same_frame(other) and bool(
            np.allclose(self.data, other.data, atol=atol, rtol=rtol))

    # ---- copies -------------------------------------------------------------------------------------
    def _to_dict(self, with_properties=True):
        out = {'data': self.data, 'times': self.times}
        if with_properties:
            out['properties'] = self.properties
        return out

    @staticmethod
    def _from_dict(features, validate=True):
        missing = {'data', 'times'} - set(features)
        if missing:
            raise ValueError(
                'cannot read features from dict, missing keys: {}'.format(', '.join(missing)))
        return Features(features['data'], features['times'],
                        properties=features.get('properties', {}), validate=validate)

    def copy(self, dtype=None, subsample=None):
        """A deep copy, optionally converted to `dtype` and keeping one frame out of `subsample`"""
        step = 1 if subsample is None else subsample
        if not isinstance(step, int) or step <= 0:
            raise ValueError(
                f'subsample must be a strictly positive integer, it is: {subsample}')
        data, times = self.data[::step], self.times[::step]
        if dtype:
            data, times = data.astype(dtype), times.astype(dtype)
        else:
            data, times = data.copy(), times.copy()
        return Features(data, times, properties=copy_properties(self.properties), validate=False)

    # ---- column-wise concatenation ------------------------------------------------------------------
    @staticmethod
    def _concatenate_meta(nframes, ndims, times, properties, other_nframes, other_times,
                          other_properties, tolerance, log):
        """Frame count, times and properties of a column-wise concatenation (the data-independent
        part of :func:`concatenate`, shared with the device-resident pipeline)"""
        diff = abs(nframes - other_nframes)
        if diff and not tolerance:
            raise ValueError('features have a different number of frames')
        if diff > tolerance:
            raise ValueError(
                'features differs number of frames, and greater than tolerance: '
                '|{} - {}| > {}'.format(nframes, other_nframes, tolerance))
        if diff:
            log.warning(
                'features differs in number of frames, but within tolerance '
                '(|%s - %s| <= %s), trim the longest one', nframes, other_nframes, tolerance)
        rows = min(nframes, other_nframes)
        if not np.allclose(times[:rows], other_times[:rows]):
            raise ValueError('times are not equal')
        merged = copy_properties(properties)
        appended = copy_properties(other_properties)
        stages = merged.setdefault('pipeline', [])
        for stage in appended.pop('pipeline', []):
            first, last = stage['columns']
            stage['columns'] = [first + ndims, last + ndims]  # the appended columns come after ours
            stages.append(stage)
        pipeline = merged.pop('pipeline')
        merged.update(appended)
        merged['pipeline'] = pipeline
        return rows, times[:rows], merged

    def concatenate(self, other, tolerance=0, log=get_logger('features', 'info')):
        """Column-wise concatenation with `other` (reference features.py:350-437): the longer of
        the two is trimmed when the frame counts differ by at most `tolerance`"""
        rows, times, properties = self._concatenate_meta(
            self.nframes, self.ndims, self.times, self.properties,
            other.nframes, other.times, other.properties, tolerance, log)
        return Features(np.hstack((self.data[:rows], other.data[:rows])), times,
                        properties=properties)


class FeaturesCollection(dict):
    """A dict of Features indexed by utterance name (counterpart of reference
    shennong/features_collection.py:80-280)"""
    @classmethod
    def load(cls, filename, serializer=None, log=None):
        """Loads a FeaturesCollection from a numpy `.npz` or Kaldi `.ark` file (the format is guessed
        from the extension unless `serializer` names it: 'numpy' or 'kaldi')"""
        from shennong_amd import serializers
        return serializers.load(cls, filename, serializer=serializer, log=log)

    def save(self, filename, serializer=None, with_properties=True, log=None, **kwargs):
        """Saves the collection to a numpy `.npz` (`compress`) or Kaldi `.ark` (`scp`, `double`) file;
        raises IOError if the file already exists"""
        from shennong_amd import serializers
        serializers.save(self, filename, serializer=serializer, with_properties=with_properties,
                         log=log, **kwargs)

    def is_valid(self):
        return all(features.is_valid() for features in self.values())

    def is_close(self, other, rtol=1e-5, atol=1e-8):
        return self.keys() == other.keys() and all(
            self[name].is_close(other[name], rtol=rtol, atol=atol) for name in self)

    def partition(self, index):
        """Returns a partition of the collection as a dict of FeaturesCollection (e.g. one per
        speaker); `index` maps every item of the collection to its sub-collection"""
        unknown = sorted(set(self) - set(index))
        if unknown:
            raise ValueError(
                'following items are not defined in the partition index: ' + ', '.join(unknown))
        parts = collections.defaultdict(FeaturesCollection)
        for name, part in index.items():
            parts[part][name] = self[name]
        return dict(parts)

    def trim(self, vad):
        """Returns a new FeaturesCollection where each features has been trimmed with the
        corresponding boolean VAD array"""
        if vad.keys() != self.keys():
            raise ValueError('Vad keys are different from this keys.')
        if any(mask.dtype != np.dtype('bool') for mask in vad.values()):
            raise ValueError('Vad arrays must be arrays of bool.')
        if any(vad[name].shape[0] != self[name].nframes for name in self):
            raise ValueError('Vad arrays length must be equal to the number of frames.')
        return FeaturesCollection(
            (name, Features(feats.data[vad[name]], feats.times[vad[name]],
                            properties=feats.properties)) for name, feats in self.items())
