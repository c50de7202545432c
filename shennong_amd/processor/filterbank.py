"""Mel-filterbank features: Audio ---> FilterbankProcessor ---> Features

Mirror of reference shennong/processor/filterbank.py:46-155 over the HIP backend.
"""

from shennong_amd import _abi
from shennong_amd.processor.base import MelFeaturesProcessor


class FilterbankProcessor(MelFeaturesProcessor):
    """Mel-filterbank features"""
    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True, num_bins=23, low_freq=20,
                 high_freq=0, vtln_low=100, vtln_high=-500,
                 use_energy=False, energy_floor=0.0, raw_energy=True,
                 htk_compat=False, use_log_fbank=True, use_power=True):
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
            snip_edges=snip_edges,
            num_bins=num_bins,
            low_freq=low_freq,
            high_freq=high_freq,
            vtln_low=vtln_low,
            vtln_high=vtln_high)
        self._opts = _abi.default_options(_abi.KIND_FBANK)
        self.use_energy = use_energy
        self.energy_floor = energy_floor
        self.raw_energy = raw_energy
        self.htk_compat = htk_compat
        self.use_log_fbank = use_log_fbank
        self.use_power = use_power

    @property
    def name(self):
        return 'filterbank'

    @property
    def use_energy(self):
        """Add an extra dimension with energy to the filterbank output"""
        return bool(self._opts.use_energy)

    @use_energy.setter
    def use_energy(self, value):
        self._opts.use_energy = bool(value)

    @property
    def energy_floor(self):
        """Floor on energy (absolute, not relative) in filterbanks"""
        return self._opts.energy_floor

    @energy_floor.setter
    def energy_floor(self, value):
        self._opts.energy_floor = value

    @property
    def raw_energy(self):
        """If true, compute energy before preemphasis and windowing"""
        return bool(self._opts.raw_energy)

    @raw_energy.setter
    def raw_energy(self, value):
        self._opts.raw_energy = bool(value)

    @property
    def htk_compat(self):
        """If True, put energy last"""
        return bool(self._opts.htk_compat)

    @htk_compat.setter
    def htk_compat(self, value):
        self._opts.htk_compat = bool(value)

    @property
    def use_log_fbank(self):
        """If true, produce log-filterbank, else produce linear"""
        return bool(self._opts.use_log_fbank)

    @use_log_fbank.setter
    def use_log_fbank(self, value):
        self._opts.use_log_fbank = bool(value)

    @property
    def use_power(self):
        """If true, use power, else use magnitude"""
        return bool(self._opts.use_power)

    @use_power.setter
    def use_power(self, value):
        self._opts.use_power = bool(value)

    @property
    def ndims(self):
        if self.use_energy:
            return self.num_bins + 1
        return self.num_bins

    def _build_options(self):
        opts = self._options(_abi.KIND_FBANK)
        for name in ('use_energy', 'energy_floor', 'raw_energy', 'htk_compat',
                     'use_log_fbank', 'use_power'):
            setattr(opts, name, getattr(self._opts, name))
        return opts
