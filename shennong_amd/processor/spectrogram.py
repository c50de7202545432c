"""Spectrogram features: Audio ---> SpectrogramProcessor ---> Features

Same parameters, defaults and outputs as reference shennong/processor/spectrogram.py:40-143; the
features come from the HIP backend (plan kind SPECTROGRAM).
"""

from shennong_amd import _abi, _backend
from shennong_amd._options import FLAG, Option
from shennong_amd.processor.base import (
    FeaturesProcessor, FramesProcessor, batch_features, check_signal)


class SpectrogramProcessor(FramesProcessor):
    """Log power spectrum, column 0 holds the frame log-energy"""
    _kind = _abi.KIND_SPECTROGRAM
    name = 'spectrogram'

    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0,
                 preemph_coeff=0.97, remove_dc_offset=True,
                 window_type='povey', round_to_power_of_two=True,
                 blackman_coeff=0.42, snip_edges=True,
                 energy_floor=0.0, raw_energy=True):
        FeaturesProcessor.__init__(self)
        self._configure(locals())

    energy_floor = Option('energy_floor', 'Floor on energy (absolute, not relative)')
    raw_energy = Option(
        'raw_energy', 'If true, compute energy before preemphasis and windowing', FLAG)

    @property
    def ndims(self):
        """Half of the padded window size, plus one"""
        return _backend.padded_window_size(self._record.frame) // 2 + 1

    def process(self, signal):
        """Compute spectrogram with the specified options (VTLN has no effect on spectrograms,
        reference spectrogram.py:113-118)"""
        return self._process_batch([signal])[0]

    def _process_batch(self, signals):
        for signal in signals:
            check_signal(self, signal)
        datas = self._run(self._build_options(), signals)
        return batch_features(datas, self.times, lambda _: self.get_properties())
