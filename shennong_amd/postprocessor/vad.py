"""Voice activity detection: Features ---> VadPostProcessor ---> Features

Mirror of reference shennong/postprocessor/vad.py:74-191 over the HIP backend (plan kind VAD,
Kaldi ComputeVadEnergy on the first column of the features).
"""

import numpy as np

from shennong_amd import _abi, _backend
from shennong_amd.features import Features
from shennong_amd.postprocessor.base import FeaturesPostProcessor


class VadPostProcessor(FeaturesPostProcessor):
    """Computes VAD on speech features"""
    def __init__(self, energy_threshold=5.0, energy_mean_scale=0.5,
                 frames_context=0, proportion_threshold=0.6):
        super().__init__()
        self._options = _abi.default_options(_abi.KIND_VAD).vad
        self.energy_threshold = energy_threshold
        self.energy_mean_scale = energy_mean_scale
        self.frames_context = frames_context
        self.proportion_threshold = proportion_threshold

    @property
    def name(self):
        return 'vad'

    @property
    def energy_threshold(self):
        """Constant term in energy threshold for MFCC0 for VAD"""
        return np.float32(self._options.energy_threshold)

    @energy_threshold.setter
    def energy_threshold(self, value):
        self._options.energy_threshold = value

    @property
    def energy_mean_scale(self):
        """Scale factor of the mean log-energy: the threshold is `s * mean + energy_threshold`"""
        return np.float32(self._options.energy_mean_scale)

    @energy_mean_scale.setter
    def energy_mean_scale(self, value):
        if value < 0:
            raise ValueError(
                'Energy mean scale must be >= 0, it is {}'.format(value))
        self._options.energy_mean_scale = value

    @property
    def frames_context(self):
        """Number of frames of context on each side of central frame"""
        return self._options.frames_context

    @frames_context.setter
    def frames_context(self, value):
        if value < 0:
            raise ValueError(
                'frames_context must be >= 0, it is {}'.format(value))
        self._options.frames_context = value

    @property
    def proportion_threshold(self):
        """Proportion of frames of the window that must exceed the threshold, in ]0, 1["""
        return np.float32(self._options.proportion_threshold)

    @proportion_threshold.setter
    def proportion_threshold(self, value):
        if value <= 0 or value >= 1:
            raise ValueError(
                'proportion_threshold must be in ]0, 1[, it is {}'
                .format(value))
        self._options.proportion_threshold = value

    @property
    def ndims(self):
        return 1

    def _build_options(self):
        opts = _abi.default_options(_abi.KIND_VAD)
        opts.vad = _abi.VadOptions.from_buffer_copy(bytes(self._options))
        return opts

    def process(self, features):
        """VAD decisions (uint8, 1 = voiced) [nframes, 1] from features whose first column
        is a log-energy"""
        return self._process_batch([features])[0]

    def _process_batch(self, features_list):
        datas = _backend.get_plan(self._build_options()).run_post(
            [np.asarray(f.data, dtype=np.float32) for f in features_list])
        return [Features(d.astype(np.uint8).reshape((-1, 1)), f.times,
                         properties=self.get_properties(f))
                for d, f in zip(datas, features_list)]
