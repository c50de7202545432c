"""Features post-processors"""

from shennong_amd.postprocessor.delta import DeltaPostProcessor

__all__ = ['DeltaPostProcessor']
