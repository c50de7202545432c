"""Features post-processors"""

import shennong_amd.processor  # noqa: F401 (import order: the processors define the base classes)

from shennong_amd.postprocessor.cmvn import (
    CmvnPostProcessor, SlidingWindowCmvnPostProcessor, apply_cmvn)
from shennong_amd.postprocessor.delta import DeltaPostProcessor
from shennong_amd.postprocessor.vad import VadPostProcessor

__all__ = ['CmvnPostProcessor', 'DeltaPostProcessor',
           'SlidingWindowCmvnPostProcessor', 'VadPostProcessor', 'apply_cmvn']
