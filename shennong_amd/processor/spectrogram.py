"""Spectrogram features: Audio ---> SpectrogramProcessor ---> Features

Mirror of reference shennong/processor/spectrogram.py:40-143 over the HIP backend.
"""

from shennong_amd import _abi
from shennong_amd.features import Features
from shennong_amd.processor.base import FramesProcessor, batch_features, check_signal


class SpectrogramProcessor(FramesProcessor):
    """Log power spectrum, column 0 holds the frame log-energy"""
    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0,
                 preemph_coeff=0.97, remove_dc_offset=True,
                 window_type='povey', round_to_power_of_two=True,
                 blackman_coeff=0.42, snip_edges=True,
                 energy_floor=0.0, raw_energy=True):
        super().__init__(
            sample_rate=sample_rate,
            frame_shift=frame_shift,
            frame_length=frame_length,
            dither=dither,
            preemph_coeff=preemph_coeff,
            remove_dc_offset=remove_dc_offset,
            window_type=window_type,
            round_to_power_of_two=round_to_power_of_two,
            blackman_coeff=blackman_coeff,
            snip_edges=snip_edges)
        self._opts = _abi.default_options(_abi.KIND_SPECTROGRAM)
        self.energy_floor = energy_floor
        self.raw_energy = raw_energy

    @property
    def name(self):
        return 'spectrogram'

    @property
    def ndims(self):
        from shennong_amd import _backend
        return int(_backend.padded_window_size(self._frame_options) / 2 + 1)

    @property
    def energy_floor(self):
        return self._opts.energy_floor

    @energy_floor.setter
    def energy_floor(self, value):
        self._opts.energy_floor = value

    @property
    def raw_energy(self):
        return bool(self._opts.raw_energy)

    @raw_energy.setter
    def raw_energy(self, value):
        self._opts.raw_energy = bool(value)

    def _build_options(self):
        opts = self._options(_abi.KIND_SPECTROGRAM)
        opts.energy_floor = self._opts.energy_floor
        opts.raw_energy = self._opts.raw_energy
        return opts

    def process(self, signal):
        """Compute spectrogram with the specified options (VTLN has no effect on
        spectrograms, reference spectrogram.py:113-118)"""
        check_signal(self, signal)
        data = self._run(self._build_options(), [signal])[0]
        return Features(
            data, self.times(data.shape[0]), properties=self.get_properties())

    def _process_batch(self, signals):
        for signal in signals:
            check_signal(self, signal)
        datas = self._run(self._build_options(), signals)
        return batch_features(datas, self.times, lambda _: self.get_properties())
