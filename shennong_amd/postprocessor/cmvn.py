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


class CmvnPostProcessor(FeaturesPostProcessor):
    """Computes CMVN statistics on speech features

    Parameters
    ----------
    dim : int
        The features dimension, must be strictly positive
    stats : array, shape = [2, dim+1]
        Preaccumulated CMVN statistics

    Raises
    ------
    ValueError
        If ``dim`` is not a strictly positive integer
    """
    def __init__(self, dim, stats=None):
        super().__init__()
        if not isinstance(dim, int) or dim <= 0:
            raise ValueError(
                'dimension must be a strictly positive integer, it is {}'
                .format(dim))
        self._dim = dim
        self._stats = np.zeros((2, dim + 1), dtype=np.float64)
        if stats is not None:
            stats = np.asarray(stats)
            if stats.shape != (2, self.dim + 1):
                raise ValueError(
                    'stats must be an array of shape {}, but is shaped as {}'
                    .format((2, self.dim + 1), stats.shape))
            self._stats[...] = stats

    @property
    def name(self):
        return 'cmvn'

    @property
    def dim(self):
        """The dimension of features on which to compute CMVN"""
        return self._dim

    @property
    def stats(self):
        """The accumulated CMVN statistics, shape [2, dim+1]: row 0 the sums (last: the
        weighted frame count), row 1 the sums of squares (last: unused)"""
        return self._stats

    @property
    def count(self):
        """The weighted total count of accumulated features frames"""
        return self.stats[0, -1]

    @property
    def ndims(self):
        return self.dim

    def get_properties(self, features):
        properties = super().get_properties(features)
        properties[self.name]['stats'] = self.stats
        return properties

    def accumulate(self, features, weights=None):
        """Accumulates the CMVN statistics of `features` (optionally frame-weighted)

        Raises
        ------
        ValueError
            If ``weights`` have more than one dimension or if ``weights`` length does not
            fit ``features`` dimension.
        """
        if weights is not None:
            if weights.ndim != 1:
                raise ValueError(
                    'weights must have a single dimension but have {}'
                    .format(weights.ndim))
            if weights.shape[0] != features.nframes:
                raise ValueError(
                    'there is {} weights but {} feature frames, must be equal'
                    .format(weights.shape[0], features.nframes))
        if features.ndims != self.dim:
            raise ValueError(
                'features dimension is {} but CMVN dimension is {}'.format(
                    features.ndims, self.dim))
        _cmvn_plan().cmvn_accumulate(
            [np.asarray(features.data, dtype=np.float32)],
            self._stats.reshape((1, 2, self.dim + 1)),
            weights=None if weights is None else [weights])

    def process(self, features, norm_vars=True, skip_dims=None, reverse=False):
        """Applies the accumulated CMVN statistics to the given ``features``

        Raises
        ------
        ValueError
            If no stats have been accumulated
        """
        if self.count < 1.0:
            raise ValueError(
                'insufficient accumulation of stats for CMVN, '
                'must be >= 1.0 but is {}'.format(self.count))
        stats = self._stats
        if skip_dims:
            dmin, dmax = min(skip_dims), max(skip_dims)
            if dmin < 0 or dmax >= features.ndims:
                raise ValueError(
                    'skipped dimensions must be in [0, {}[ but are in [{}, {}['
                    .format(features.ndims, dmin, dmax))
            stats = _fake_stats_for_dims(stats, skip_dims)
        data = _cmvn_plan().cmvn_apply(
            [np.asarray(features.data, dtype=np.float32)],
            stats.reshape((1, 2, self.dim + 1)),
            norm_vars=norm_vars, reverse=reverse)[0]
        return Features(
            data, features.times, properties=self.get_properties(features))


def apply_cmvn(feats_collection, by_collection=True, norm_vars=True,
               weights=None, skip_dims=None):
    """CMVN normalization of a collection of features, over the whole collection
    (`by_collection`) or independently for each item.  One statistics launch and one apply
    launch cover the whole collection.

    Raises
    ------
    ValueError
        If something goes wrong during CMVN processing.
    """
    dim = set(f.ndims for f in feats_collection.values())
    if not len(dim) == 1:
        raise ValueError(
            'features in the collection must have consistent dimensions '
            'but dimensions are: {}'.format(sorted(dim)))
    dim = list(dim)[0]

    if weights is not None and weights.keys() != feats_collection.keys():
        raise ValueError('keys differ for weights and features collection')

    if skip_dims is not None:
        sdmin, sdmax = min(skip_dims), max(skip_dims)
        if sdmin < 0 or sdmax >= dim:
            raise ValueError(
                'out of bounds dimensions in skip_dims, must be in [0, {}] '
                'but are in [{}, {}]'.format(dim - 1, sdmin, sdmax))

    keys = list(feats_collection.keys())
    feats = [feats_collection[k] for k in keys]
    if weights is not None and all(weights[k] is None for k in keys):
        weights = None
    if weights is not None:
        weights = {
            k: (np.ones(f.nframes, dtype=np.float32) if weights[k] is None
                else np.asarray(weights[k]))
            for k, f in zip(keys, feats)}
        for k, f in zip(keys, feats):
            w = weights[k]
            if w.ndim != 1:
                raise ValueError(
                    'weights must have a single dimension but have {}'
                    .format(w.ndim))
            if w.shape[0] != f.nframes:
                raise ValueError(
                    'there is {} weights but {} feature frames, must be equal'
                    .format(w.shape[0], f.nframes))
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
        if stats[g, 0, -1] < 1.0:
            raise ValueError(
                'insufficient accumulation of stats for CMVN, '
                'must be >= 1.0 but is {}'.format(stats[g, 0, -1]))
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
