"""Features extraction processors (same names as reference shennong/processor/__init__.py)"""

from shennong_amd.processor.energy import EnergyProcessor
from shennong_amd.processor.filterbank import FilterbankProcessor
from shennong_amd.processor.mfcc import MfccProcessor
from shennong_amd.processor.plp import PlpProcessor
from shennong_amd.processor.spectrogram import SpectrogramProcessor
from shennong_amd.processor.pitch_kaldi import (
    KaldiPitchProcessor, KaldiPitchPostProcessor)

__all__ = [
    'EnergyProcessor', 'FilterbankProcessor', 'MfccProcessor', 'PlpProcessor',
    'SpectrogramProcessor', 'KaldiPitchProcessor', 'KaldiPitchPostProcessor']
