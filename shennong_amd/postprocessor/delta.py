"""Delta / delta-delta: Features ---> DeltaPostProcessor ---> Features
(mirror of reference shennong/postprocessor/delta.py:53-136 over the HIP backend)"""


import numpy as np

from shennong_amd import _abi, _backend
from shennong_amd.features import Features
from shennong_amd.postprocessor.base import FeaturesPostProcessor
from shennong_amd.utils import copy_properties


class DeltaPostProcessor(FeaturesPostProcessor):
    def __init__(self, order=2, window=2):
        super().__init__()
        self._order = 2
        self._window = 2
        self.order = order
        self.window = window

    @property
    def name(self):
        return 'delta'

    @property
    def order(self):
        """Order of delta computation"""
        return self._order

    @order.setter
    def order(self, value):
        self._order = int(value)

    @property
    def window(self):
        """The actual window size for each delta order is 1 + 2 * `window`; edges replicate
        the first or last frame"""
        return self._window

    @window.setter
    def window(self, value):
        if not 0 < value < 1000:
            raise ValueError(
                'window must be in [1, 999], it is {}'.format(value))
        self._window = int(value)

    @property
    def ndims(self):
        raise ValueError(
            'output dimension for delta processor depends on input')

    def get_properties(self, features):
        ndims = (self.order + 1) * features.ndims
        properties = copy_properties(features.properties)
        properties[self.name] = {
            'order': self.order,
            'window': self.window}
        if 'pipeline' not in properties:
            properties['pipeline'] = []
        properties['pipeline'].append({
            'name': self.name,
            'columns': [0, ndims - 1]})
        return properties

    def _build_options(self):
        opts = _abi.default_options(_abi.KIND_DELTA)
        opts.delta_order = self.order
        opts.delta_window = self.window
        return opts

    def process(self, features):
        """Compute deltas on `features`: [nframes, ncols] -> [nframes, ncols * (order + 1)]"""
        data = _backend.get_plan(self._build_options()).run_post(
            [np.asarray(features.data, dtype=np.float32)])[0]
        return Features(data, features.times, self.get_properties(features))

    def _process_batch(self, features_list):
        datas = _backend.get_plan(self._build_options()).run_post(
            [np.asarray(f.data, dtype=np.float32) for f in features_list], check_finite=True)
        return [Features(d, f.times, self.get_properties(f), validate=False)
                for d, f in zip(datas, features_list)]
