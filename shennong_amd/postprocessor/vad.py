"""Voice activity detection: Features ---> VadPostProcessor ---> Features

Same parameters, outputs and error messages as reference shennong/postprocessor/vad.py:74-191 (Kaldi
ComputeVadEnergy on the first column of the features; plan kind VAD on the HIP backend).
"""

import numpy as np

from shennong_amd import _abi, _backend
from shennong_amd._options import F32, Configurable, Option, require
from shennong_amd.features import Features
from shennong_amd.postprocessor.base import FeaturesPostProcessor


class VadPostProcessor(Configurable, FeaturesPostProcessor):
    """Computes VAD on speech features"""
    _kind = _abi.KIND_VAD
    name = 'vad'
    ndims = 1

    def __init__(self, energy_threshold=5.0, energy_mean_scale=0.5,
                 frames_context=0, proportion_threshold=0.6):
        super().__init__()
        self._configure(locals())

    energy_threshold = Option(
        'vad.energy_threshold', 'Constant term in energy threshold for MFCC0 for VAD', F32)
    energy_mean_scale = Option(
        'vad.energy_mean_scale',
        'Scale factor of the mean log-energy: the threshold is `s * mean + energy_threshold`', F32,
        check=require(lambda v: v >= 0, 'Energy mean scale must be >= 0, it is {}'))
    frames_context = Option(
        'vad.frames_context', 'Number of frames of context on each side of central frame',
        check=require(lambda v: v >= 0, 'frames_context must be >= 0, it is {}'))
    proportion_threshold = Option(
        'vad.proportion_threshold',
        'Proportion of frames of the window that must exceed the threshold, in ]0, 1[', F32,
        check=require(lambda v: 0 < v < 1, 'proportion_threshold must be in ]0, 1[, it is {}'))

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
