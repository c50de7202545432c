"""PLP / RASTA-PLP features: Audio ---> PlpProcessor ---> Features

Mirror of reference shennong/processor/plp.py:263-676.  The reference runs the PLP recipe as a
per-frame Python loop over pykaldi primitives (plp.py:536-544) and keeps mutable per-instance
buffers (racy under process_all); here the recipe runs as three stateless kernels
(fused mel extraction -> RASTA scan per (utterance, bin) -> per-frame PLP tail).
"""

import numpy as np

from shennong_amd import _abi
from shennong_amd.processor.base import MelFeaturesProcessor


class PlpProcessor(MelFeaturesProcessor):
    """Perceptive linear predictive features"""
    def __init__(self, sample_rate=16000, frame_shift=0.01, frame_length=0.025,
                 rasta=False, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True, num_bins=23, low_freq=20, high_freq=0,
                 vtln_low=100, vtln_high=-500, lpc_order=12, num_ceps=13,
                 use_energy=True, energy_floor=0.0, raw_energy=True,
                 compress_factor=1.0/3.0, cepstral_lifter=22,
                 cepstral_scale=1.0, htk_compat=False):
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
        self._opts = _abi.default_options(_abi.KIND_PLP)
        self.rasta = rasta
        self.lpc_order = lpc_order
        self.num_ceps = num_ceps
        self.use_energy = use_energy
        self.energy_floor = energy_floor
        self.raw_energy = raw_energy
        self.compress_factor = compress_factor
        self.cepstral_lifter = cepstral_lifter
        self.cepstral_scale = cepstral_scale
        self.htk_compat = htk_compat

    @property
    def name(self):
        return 'plp'

    @property
    def rasta(self):
        """Whether to do RASTA filtering"""
        return bool(self._opts.rasta)

    @rasta.setter
    def rasta(self, value):
        self._opts.rasta = bool(value)

    @property
    def lpc_order(self):
        """Order of LPC analysis in PLP computation"""
        return self._opts.lpc_order

    @lpc_order.setter
    def lpc_order(self, value):
        self._opts.lpc_order = value

    @property
    def num_ceps(self):
        """Number of cepstra in PLP computation (including C0), in [1, lpc_order + 1]"""
        return self._opts.num_ceps

    @num_ceps.setter
    def num_ceps(self, value):
        value = int(value)
        if value <= 0:
            raise ValueError('num_ceps must be > 0')
        if value > self.lpc_order + 1:
            raise ValueError(
                'We must have num_ceps <= lpc_order+1, but {} > {}+1'.format(
                    value, self.lpc_order))
        self._opts.num_ceps = value

    @property
    def use_energy(self):
        """Use energy (instead of C0) for zeroth PLP feature"""
        return bool(self._opts.use_energy)

    @use_energy.setter
    def use_energy(self, value):
        self._opts.use_energy = bool(value)

    @property
    def energy_floor(self):
        """Floor on energy (absolute, not relative) in PLP computation"""
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
    def compress_factor(self):
        """Compression factor in PLP computation"""
        return np.float32(self._opts.compress_factor)

    @compress_factor.setter
    def compress_factor(self, value):
        self._opts.compress_factor = value

    @property
    def cepstral_lifter(self):
        """Constant that controls scaling of PLPs"""
        return self._opts.cepstral_lifter

    @cepstral_lifter.setter
    def cepstral_lifter(self, value):
        self._opts.cepstral_lifter = value

    @property
    def cepstral_scale(self):
        """Scaling constant in PLP computation"""
        return self._opts.cepstral_scale

    @cepstral_scale.setter
    def cepstral_scale(self, value):
        self._opts.cepstral_scale = value

    @property
    def htk_compat(self):
        """If True, put energy or C0 last"""
        return bool(self._opts.htk_compat)

    @htk_compat.setter
    def htk_compat(self, value):
        self._opts.htk_compat = bool(value)

    @property
    def ndims(self):
        return self.num_ceps

    def _build_options(self):
        opts = self._options(_abi.KIND_PLP)
        for name in ('rasta', 'lpc_order', 'num_ceps', 'use_energy',
                     'energy_floor', 'raw_energy', 'compress_factor',
                     'cepstral_lifter', 'cepstral_scale', 'htk_compat'):
            setattr(opts, name, getattr(self._opts, name))
        return opts
