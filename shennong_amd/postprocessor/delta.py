"""Delta / delta-delta: Features ---> DeltaPostProcessor ---> Features

Same parameters, outputs and error messages as reference shennong/postprocessor/delta.py:53-136
(Kaldi ComputeDeltas; plan kind DELTA on the HIP backend).
"""

import numpy as np

from shennong_amd import _abi, _backend
from shennong_amd._options import Configurable, Option, require
from shennong_amd.features import Features
from shennong_amd.postprocessor.base import FeaturesPostProcessor


class DeltaPostProcessor(Configurable, FeaturesPostProcessor):
    """Appends the time derivatives of the input columns, up to `order`"""
    _kind = _abi.KIND_DELTA
    name = 'delta'

    def __init__(self, order=2, window=2):
        super().__init__()
        self._configure(locals())

    order = Option('delta_order', 'Order of delta computation')
    window = Option(
        'delta_window',
        'The actual window size for each delta order is 1 + 2 * `window`; edges replicate the '
        'first or last frame',
        check=require(lambda v: 0 < v < 1000, 'window must be in [1, 999], it is {}'))

    @property
    def ndims(self):
        raise ValueError(
            'output dimension for delta processor depends on input')

    def get_properties(self, features):
        """The input's properties + this processor's parameters + one more pipeline stage covering
        the (order + 1) * ndims output columns"""
        properties = self._extend_properties(features, (self.order + 1) * features.ndims)
        properties[self.name] = {'order': self.order, 'window': self.window}
        return properties

    def process(self, features):
        """Compute deltas on `features`: [nframes, ncols] -> [nframes, ncols * (order + 1)]"""
        return self._process_batch([features])[0]

    def _process_batch(self, features_list):
        datas = _backend.get_plan(self._build_options()).run_post(
            [np.asarray(f.data, dtype=np.float32) for f in features_list], check_finite=True)
        return [Features(d, f.times, self.get_properties(f), validate=False)
                for d, f in zip(datas, features_list)]
