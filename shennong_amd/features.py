"""Features container: the output of every processor

Mirrors reference shennong/features.py:62-437 (data [nframes, ndims], times [nframes, 2] or
[nframes], properties dict; validate / __eq__ / is_close / copy / concatenate).  The reference's
per-frame Python loop in ``validate`` (features.py:342) is replaced by a vectorised check.
"""

import collections
import copy

import numpy as np

from shennong_amd.logger import get_logger
from shennong_amd.utils import dict_equal


class Features:
    """Features data with attached timestamps and properties"""
    def __init__(self, data, times, properties=None, validate=True):
        self._data = data
        self._times = times
        self._properties = {} if properties is None else properties
        if validate is True:
            self.validate()

    @property
    def data(self):
        return self._data

    @property
    def times(self):
        return self._times

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def shape(self):
        return self.data.shape

    @property
    def ndims(self):
        return self.shape[1]

    @property
    def nframes(self):
        return self.shape[0]

    @property
    def properties(self):
        return self._properties

    def _to_dict(self, with_properties=True):
        features = {'data': self.data, 'times': self.times}
        if with_properties:
            features['properties'] = self.properties
        return features

    @staticmethod
    def _from_dict(features, validate=True):
        missing_keys = {'data', 'times'} - set(features.keys())
        if missing_keys:
            raise ValueError(
                'cannot read features from dict, missing keys: {}'
                .format(', '.join(missing_keys)))
        return Features(
            features['data'], features['times'],
            properties=features.get('properties', {}), validate=validate)

    def __eq__(self, other):
        if self is other:
            return True
        if self.shape != other.shape or self.dtype != other.dtype:
            return False
        if not dict_equal(self.properties, other.properties):
            return False
        if not np.array_equal(self.times, other.times):
            return False
        return bool(np.array_equal(self.data, other.data))

    def is_close(self, other, rtol=1e-5, atol=1e-8):
        """True if data is allclose and shape / times / properties are equal"""
        if self is other:
            return True
        if self.shape != other.shape:
            return False
        if not dict_equal(self.properties, other.properties):
            return False
        if not np.array_equal(self.times, other.times):
            return False
        return bool(np.allclose(self.data, other.data, atol=atol, rtol=rtol))

    def copy(self, dtype=None, subsample=None):
        if subsample is None:
            subsample = 1
        elif not isinstance(subsample, int) or subsample <= 0:
            raise ValueError(
                f'subsample must be a strictly positive integer, '
                f'it is: {subsample}')
        if dtype:
            return Features(
                self.data[0:self.nframes:subsample].astype(dtype),
                self.times[0:self.nframes:subsample].astype(dtype),
                properties=copy.deepcopy(self.properties), validate=False)
        return Features(
            self.data[0:self.nframes:subsample].copy(),
            self.times[0:self.nframes:subsample].copy(),
            properties=copy.deepcopy(self.properties), validate=False)

    def is_valid(self):
        try:
            self.validate()
        except ValueError:
            return False
        return True

    def validate(self):
        """Raises a ValueError if the features are not in a valid state"""
        errors = []
        if not isinstance(self.data, np.ndarray):
            errors.append('data must be a numpy array')
        if not isinstance(self.times, np.ndarray):
            errors.append('times must be a numpy array')
        if not isinstance(self.properties, dict):
            errors.append('properties must be a dictionnary')
        if errors:
            raise ValueError(
                'invalid features data types: {}'.format(', '.join(errors)))
        if not self.data.ndim == 2:
            errors.append(
                'data dimension must be 2 but is {}'.format(self.data.ndim))
        if self.times.ndim > 2:
            errors.append('times dimension must be 1 or 2 but is {}'.format(
                self.times.ndim))
        if self.times.ndim == 2 and self.times.shape[1] != 2:
            errors.append('times shape[1] must be 2, it is {}'.format(
                self.times.shape[1]))
        nframes1 = self.data.shape[0]
        nframes2 = self.times.shape[0]
        if not nframes1 == nframes2:
            errors.append(
                'mismatch in number of frames: {} for data but {} '
                'for times'.format(nframes1, nframes2))
        if errors:
            raise ValueError(
                'invalid features dimensions: {}'.format(', '.join(errors)))
        # times must be sorted in increasing order (stable sort is identity)
        index = (np.argsort(self.times, kind='stable')
                 if self.times.ndim == 1 else np.lexsort(self.times.T))
        if not np.array_equal(index, np.arange(self.nframes)):
            raise ValueError('times is not sorted in increasing order')
        if not np.all(np.isfinite(self.data)):
            raise ValueError(
                'data contains non-finite numbers (nan of infinity)')

    @staticmethod
    def _concatenate_meta(nframes, ndims, times, properties, other_nframes, other_times,
                          other_properties, tolerance, log):
        """Frame count, times and properties of a column-wise concatenation (the data-independent
        part of :func:`concatenate`, shared with the device-resident pipeline)"""
        diff = abs(nframes - other_nframes)
        if diff:
            if not tolerance:
                raise ValueError('features have a different number of frames')
            if tolerance and diff > tolerance:
                raise ValueError(
                    'features differs number of frames, and '
                    'greater than tolerance: |{} - {}| > {}'.format(
                        nframes, other_nframes, tolerance))
            log.warning(
                'features differs in number of frames, but '
                'within tolerance (|%s - %s| <= %s), trim the longest one',
                nframes, other_nframes, tolerance)
        rows = min(nframes, other_nframes)
        times1, times2 = times[:rows], other_times[:rows]
        if not np.allclose(times1, times2):
            raise ValueError('times are not equal')
        properties = copy.deepcopy(properties)
        other_properties = copy.deepcopy(other_properties)
        properties.update(
            {k: v for k, v in other_properties.items() if k != 'pipeline'})
        if 'pipeline' not in properties:
            properties['pipeline'] = []
        if 'pipeline' in other_properties:
            for k in other_properties['pipeline']:
                properties['pipeline'].append(k)
                columns = properties['pipeline'][-1]['columns']
                properties['pipeline'][-1]['columns'] = [
                    columns[0] + ndims, columns[1] + ndims]
        return rows, times1, properties

    def concatenate(self, other, tolerance=0,
                    log=get_logger('features', 'info')):
        """Column-wise concatenation with `other` (reference features.py:350-437)"""
        rows, times, properties = self._concatenate_meta(
            self.nframes, self.ndims, self.times, self.properties,
            other.nframes, other.times, other.properties, tolerance, log)
        return Features(
            np.hstack((self.data[:rows], other.data[:rows])), times,
            properties=properties)


class FeaturesCollection(dict):
    """A dict of Features indexed by utterance name (mirror of reference
    shennong/features_collection.py:80-280)"""
    @classmethod
    def load(cls, filename, serializer=None, log=None):
        """Loads a FeaturesCollection from a `filename`; the serializer is guessed from the file
        extension when not specified (see shennong_amd.serializers)"""
        from shennong_amd.logger import get_logger
        from shennong_amd.serializers import get_serializer
        log = log or get_logger('serializer', 'warning')
        return get_serializer(cls, filename, log, serializer).load()

    def save(self, filename, serializer=None, with_properties=True, log=None, **kwargs):
        """Saves a FeaturesCollection to a `filename` (`compress` for numpy / matlab, `scp` for
        kaldi); raises IOError if the file already exists"""
        from shennong_amd.logger import get_logger
        from shennong_amd.serializers import get_serializer
        log = log or get_logger('serializer', 'warning')
        get_serializer(self.__class__, filename, log, serializer).save(
            self, with_properties=with_properties, **kwargs)

    def is_valid(self):
        return all(f.is_valid() for f in self.values())

    def is_close(self, other, rtol=1e-5, atol=1e-8):
        if not self.keys() == other.keys():
            return False
        return all(
            self[k].is_close(other[k], rtol=rtol, atol=atol) for k in self)

    def partition(self, index):
        """Returns a partition of the collection as a dict of FeaturesCollection (e.g. one per
        speaker); `index` maps every item of the collection to its sub-collection"""
        undefined_utts = set(self.keys()).difference(index.keys())
        if undefined_utts:
            raise ValueError(
                'following items are not defined in the partition index: {}'
                .format(', '.join(sorted(undefined_utts))))
        reverse_index = collections.defaultdict(list)
        for key, value in index.items():
            reverse_index[value].append(key)
        return {k: FeaturesCollection({item: self[item] for item in items})
                for k, items in reverse_index.items()}

    def trim(self, vad):
        """Returns a new FeaturesCollection where each features has been trimmed with the
        corresponding boolean VAD array"""
        if vad.keys() != self.keys():
            raise ValueError('Vad keys are different from this keys.')
        for key in vad.keys():
            if vad[key].dtype != np.dtype('bool'):
                raise ValueError('Vad arrays must be arrays of bool.')
            if vad[key].shape[0] != self[key].nframes:
                raise ValueError(
                    'Vad arrays length must be equal to the number of frames.')
        return FeaturesCollection({
            k: Features(
                self[k].data[vad[k]],
                self[k].times[vad[k]],
                properties=self[k].properties) for k in self.keys()})
