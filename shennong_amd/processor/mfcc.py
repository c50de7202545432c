"""MFCC features: Audio ---> MfccProcessor ---> Features

Mirror of reference shennong/processor/mfcc.py:46-159 over the HIP backend.
"""

from shennong_amd import _abi
from shennong_amd.processor.base import MelFeaturesProcessor


class MfccProcessor(MelFeaturesProcessor):
    """Mel Frequency Cepstral Coeficients"""
    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True, num_bins=23, low_freq=20,
                 high_freq=0, vtln_low=100, vtln_high=-500,
                 num_ceps=13, use_energy=True, energy_floor=0.0,
                 raw_energy=True, cepstral_lifter=22.0,
                 htk_compat=False):
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
        self._opts = _abi.default_options(_abi.KIND_MFCC)
        self.num_ceps = num_ceps
        self.use_energy = use_energy
        self.energy_floor = energy_floor
        self.raw_energy = raw_energy
        self.cepstral_lifter = cepstral_lifter
        self.htk_compat = htk_compat

    @property
    def name(self):
        return 'mfcc'

    @property
    def num_ceps(self):
        """Number of cepstra in MFCC computation (including C0), <= `num_bins`"""
        return self._opts.num_ceps

    @num_ceps.setter
    def num_ceps(self, value):
        self._opts.num_ceps = value

    @property
    def use_energy(self):
        """Use energy (instead of C0) in MFCC computation"""
        return bool(self._opts.use_energy)

    @use_energy.setter
    def use_energy(self, value):
        self._opts.use_energy = bool(value)

    @property
    def energy_floor(self):
        """Floor on energy (absolute, not relative) in MFCC computation"""
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
    def cepstral_lifter(self):
        """Constant that controls scaling of MFCCs"""
        return self._opts.cepstral_lifter

    @cepstral_lifter.setter
    def cepstral_lifter(self, value):
        self._opts.cepstral_lifter = value

    @property
    def htk_compat(self):
        """If True, put energy or C0 last and use a factor of sqrt(2) on C0"""
        return bool(self._opts.htk_compat)

    @htk_compat.setter
    def htk_compat(self, value):
        self._opts.htk_compat = bool(value)

    @property
    def ndims(self):
        return self.num_ceps

    def _build_options(self):
        opts = self._options(_abi.KIND_MFCC)
        for name in ('num_ceps', 'use_energy', 'energy_floor', 'raw_energy',
                     'cepstral_lifter', 'htk_compat'):
            setattr(opts, name, getattr(self._opts, name))
        return opts
