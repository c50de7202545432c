"""Cepstral mean variance normalization (CMVN) and sliding-window CMVN

Mirror of reference shennong/postprocessor/cmvn.py:83-500 over the HIP backend: statistics are
reduced per utterance on the GPU (plan kind CMVN, Kaldi AccCmvnStats), summed per speaker on the
host (or all-reduced over ranks, see shennong_amd/distributed.py), and applied by one kernel
(Kaldi ApplyCmvn / ApplyCmvnReverse).  Sliding-window normalisation is plan kind SLIDING_CMVN
(Kaldi SlidingWindowCmn).
"""


import numpy as np

from shennong_amd import _abi, _backend
from shennong_amd._options import FLAG, Configurable, Option
from shennong_amd.features import Features, FeaturesCollection
from shennong_amd.postprocessor.base import FeaturesPostProcessor
from shennong_amd.utils import copy_properties


def _cmvn_plan():
    return _backend.get_plan(_abi.default_options(_abi.KIND_CMVN))


def _fake_stats_for_dims(stats, skip_dims):
    """Kaldi FakeStatsForSomeDims: mean 0, variance 1 on the skipped dimensions"""
    stats = np.array(stats, dtype=np.float64, copy=True)
    count = stats[0, -1]
    for d in skip_dims:
        stats[0, d] = 0.0
        stats[1, d] = count
    return stats


def _require(ok, message, *values):
    """Argument checks of this module: the reference's ValueError texts (its tests match on them,
    test/postprocessor/test_cmvn.py), written once each as a format string"""
    if not ok:
        raise ValueError(message.format(*values))


_MESSAGES = {
    'dim': 'dimension must be a strictly positive integer, it is {}',
    'stats': 'stats must be an array of shape {}, but is shaped as {}',
    'weights_ndim': 'weights must have a single dimension but have {}',
    'weights_len': 'there is {} weights but {} feature frames, must be equal',
    'features_dim': 'features dimension is {} but CMVN dimension is {}',
    'count': 'insufficient accumulation of stats for CMVN, must be >= 1.0 but is {}',
    'skip': 'skipped dimensions must be in [0, {}[ but are in [{}, {}[',
    'collection_dims': 'features in the collection must have consistent dimensions but dimensions are: {}',
    'keys': 'keys differ for weights and features collection',
    'skip_collection': 'out of bounds dimensions in skip_dims, must be in [0, {}] but are in [{}, {}]',
}


class CmvnPostProcessor(FeaturesPostProcessor):
    """CMVN statistics of speech features and their application (reference cmvn.py:83-282)

    `dim` is the features dimension (a strictly positive int, else ValueError); `stats`, when given,
    are pre-accumulated statistics of shape [2, dim + 1]."""
    name = property(lambda self: 'cmvn')
    dim = property(lambda self: self._dim, doc='The dimension of features on which to compute CMVN')
    ndims = property(lambda self: self._dim)
    stats = property(lambda self: self._stats, doc=(
        'The accumulated CMVN statistics, shape [2, dim+1]: row 0 the sums (last: the weighted frame '
        'count), row 1 the sums of squares (last: unused)'))
    count = property(lambda self: self._stats[0, -1],
                     doc='The weighted total count of accumulated features frames')

    def __init__(self, dim, stats=None):
        super().__init__()
        _require(isinstance(dim, int) and dim > 0, _MESSAGES['dim'], dim)
        self._dim = dim
        shape = (2, dim + 1)
        self._stats = np.zeros(shape, dtype=np.float64)
        if stats is not None:
            given = np.asarray(stats)
            _require(given.shape == shape, _MESSAGES['stats'], shape, given.shape)
            self._stats[...] = given

    def get_properties(self, features):
        properties = super().get_properties(features)
        properties[self.name]['stats'] = self.stats
        return properties

    def _as_batch(self, stats):
        return stats.reshape((1, 2, self._dim + 1))

    def accumulate(self, features, weights=None):
        """Adds the statistics of `features` (one weight per frame when `weights` is given) to `stats`;
        ValueError for weights that are not a vector of one value per frame, or features of another dimension"""
        if weights is not None:
            _require(weights.ndim == 1, _MESSAGES['weights_ndim'], weights.ndim)
            _require(weights.shape[0] == features.nframes, _MESSAGES['weights_len'],
                     weights.shape[0], features.nframes)
        _require(features.ndims == self._dim, _MESSAGES['features_dim'], features.ndims, self._dim)
        _cmvn_plan().cmvn_accumulate(
            [np.asarray(features.data, dtype=np.float32)], self._as_batch(self._stats),
            weights=None if weights is None else [weights])

    def process(self, features, norm_vars=True, skip_dims=None, reverse=False):
        """`features` normalised with the accumulated statistics (mean only when `norm_vars` is false; the
        dimensions of `skip_dims` untouched; `reverse` undoes a normalisation); ValueError before any
        statistics were accumulated"""
        _require(self.count >= 1.0, _MESSAGES['count'], self.count)
        stats = self._stats
        if skip_dims:
            first, last = min(skip_dims), max(skip_dims)
            _require(first >= 0 and last < features.ndims, _MESSAGES['skip'], features.ndims, first, last)
            stats = _fake_stats_for_dims(stats, skip_dims)
        normalised = _cmvn_plan().cmvn_apply(
            [np.asarray(features.data, dtype=np.float32)], self._as_batch(stats),
            norm_vars=norm_vars, reverse=reverse)[0]
        return Features(normalised, features.times, properties=self.get_properties(features))


def apply_cmvn(feats_collection, by_collection=True, norm_vars=True,
               weights=None, skip_dims=None):
    """CMVN normalization of a collection of features, over the whole collection
    (`by_collection`) or independently for each item (reference cmvn.py:285-390).  One statistics
    launch and one apply launch cover the whole collection.  ValueError for inconsistent dimensions,
    weights whose keys or lengths do not fit, skipped dimensions out of range, or an empty count."""
    keys = list(feats_collection.keys())
    feats = [feats_collection[k] for k in keys]
    dims = sorted(set(f.ndims for f in feats))
    _require(len(dims) == 1, _MESSAGES['collection_dims'], dims)
    dim = dims[0]
    _require(weights is None or weights.keys() == feats_collection.keys(), _MESSAGES['keys'])
    if skip_dims is not None:
        first, last = min(skip_dims), max(skip_dims)
        _require(first >= 0 and last < dim, _MESSAGES['skip_collection'], dim - 1, first, last)
    if weights is not None and all(weights[k] is None for k in keys):
        weights = None
    if weights is not None:
        filled = {}
        for k, f in zip(keys, feats):
            w = np.ones(f.nframes, dtype=np.float32) if weights[k] is None else np.asarray(weights[k])
            _require(w.ndim == 1, _MESSAGES['weights_ndim'], w.ndim)
            _require(w.shape[0] == f.nframes, _MESSAGES['weights_len'], w.shape[0], f.nframes)
            filled[k] = w
        weights = filled
    n = len(feats)
    if n == 0:
        return FeaturesCollection()
    n_groups = 1 if by_collection else n
    groups = None if by_collection else np.arange(n, dtype=np.int32)
    mats = [np.asarray(f.data, dtype=np.float32) for f in feats]
    plan = _cmvn_plan()
    stats = np.zeros((n_groups, 2, dim + 1), dtype=np.float64)
    plan.cmvn_accumulate(
        mats, stats, groups=groups,
        weights=None if weights is None else [weights[k] for k in keys])
    for g in range(n_groups):
        _require(stats[g, 0, -1] >= 1.0, _MESSAGES['count'], stats[g, 0, -1])
    applied = stats
    if skip_dims:
        applied = np.stack(
            [_fake_stats_for_dims(stats[g], skip_dims)
             for g in range(n_groups)])
    datas = plan.cmvn_apply(mats, applied, groups=groups, norm_vars=norm_vars)
    out = FeaturesCollection()
    for u, (k, f) in enumerate(zip(keys, feats)):
        proc = CmvnPostProcessor(dim, stats=stats[0 if by_collection else u])
        out[k] = Features(
            datas[u], f.times, properties=proc.get_properties(f))
    return out


class SlidingWindowCmvnPostProcessor(Configurable, FeaturesPostProcessor):
    """Sliding-window mean (and variance) normalisation of speech features (Kaldi
    SlidingWindowCmn; same parameters and defaults as reference postprocessor/cmvn.py:400-500;
    plan kind SLIDING_CMVN on the HIP backend)

    `max_warnings` is accepted for compatibility: the reference passes it to Kaldi, which only uses
    it to limit log messages."""
    _kind = _abi.KIND_SLIDING_CMVN
    name = 'sliding_window_cmvn'

    def __init__(self, center=True, cmn_window=600, min_window=100,
                 max_warnings=5, normalize_variance=False):
        super().__init__()
        self._max_warnings = 5
        self._configure(locals())

    center = Option(
        'sliding_cmvn.center', 'Whether to center the window on the current frame', FLAG)
    cmn_window = Option('sliding_cmvn.cmn_window', 'Window size for average CMN computation')
    min_window = Option(
        'sliding_cmvn.min_window', 'Minimum CMN window used at start of decoding')
    normalize_variance = Option(
        'sliding_cmvn.normalize_variance', 'Whether to normalize variance to one', FLAG)

    @property
    def max_warnings(self):
        """Maximum warning to report per utterance"""
        return self._max_warnings

    @max_warnings.setter
    def max_warnings(self, value):
        self._max_warnings = int(value)

    @property
    def ndims(self):
        raise ValueError('output dimension for sliding '
                         'window CMVN processor depends on input')

    def get_properties(self, features):
        properties = self._extend_properties(features, features.ndims)
        properties[self.name] = self.get_params()
        return properties

    def process(self, features):
        """Applies sliding-window cepstral mean and/or variance normalization"""
        return self._process_batch([features])[0]

    def _process_batch(self, features_list):
        datas = _backend.get_plan(self._build_options()).run_post(
            [np.asarray(f.data, dtype=np.float32) for f in features_list], check_finite=True)
        return [Features(d, f.times, self.get_properties(f), validate=False)
                for d, f in zip(datas, features_list)]
