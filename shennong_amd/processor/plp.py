"""PLP / RASTA-PLP features: Audio ---> PlpProcessor ---> Features

Same parameters, defaults and outputs as reference shennong/processor/plp.py:263-676.  The
reference runs the PLP recipe as a per-frame Python loop over pykaldi primitives (plp.py:536-544)
and keeps mutable per-instance buffers (racy under process_all); here the recipe runs
as three stateless kernels (fused mel extraction -> RASTA scan per (utterance, bin) -> per-frame PLP
tail; plan kind PLP).
"""

from shennong_amd import _abi
from shennong_amd._options import F32, FLAG, Option
from shennong_amd.processor.base import FeaturesProcessor, MelFeaturesProcessor


class PlpProcessor(MelFeaturesProcessor):
    """Perceptive linear predictive features"""
    _kind = _abi.KIND_PLP
    name = 'plp'

    def __init__(self, sample_rate=16000, frame_shift=0.01, frame_length=0.025,
                 rasta=False, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True, num_bins=23, low_freq=20, high_freq=0,
                 vtln_low=100, vtln_high=-500, lpc_order=12, num_ceps=13,
                 use_energy=True, energy_floor=0.0, raw_energy=True,
                 compress_factor=1.0/3.0, cepstral_lifter=22,
                 cepstral_scale=1.0, htk_compat=False):
        FeaturesProcessor.__init__(self)
        self._configure(locals())  # (in signature order: lpc_order is known when num_ceps is checked)

    rasta = Option('rasta', 'Whether to do RASTA filtering', FLAG)
    lpc_order = Option('lpc_order', 'Order of LPC analysis in PLP computation')
    use_energy = Option('use_energy', 'Use energy (instead of C0) for zeroth PLP feature', FLAG)
    energy_floor = Option(
        'energy_floor', 'Floor on energy (absolute, not relative) in PLP computation')
    raw_energy = Option(
        'raw_energy', 'If true, compute energy before preemphasis and windowing', FLAG)
    compress_factor = Option('compress_factor', 'Compression factor in PLP computation', F32)
    cepstral_lifter = Option('cepstral_lifter', 'Constant that controls scaling of PLPs')
    cepstral_scale = Option('cepstral_scale', 'Scaling constant in PLP computation')
    htk_compat = Option('htk_compat', 'If True, put energy or C0 last', FLAG)

    @property
    def num_ceps(self):
        """Number of cepstra in PLP computation (including C0), in [1, lpc_order + 1]"""
        return self._record.num_ceps

    @num_ceps.setter
    def num_ceps(self, value):
        value = int(value)
        if value <= 0:
            raise ValueError('num_ceps must be > 0')
        if value > self.lpc_order + 1:
            raise ValueError(
                'We must have num_ceps <= lpc_order+1, but {} > {}+1'.format(
                    value, self.lpc_order))
        self._record.num_ceps = value

    @property
    def ndims(self):
        return self.num_ceps
