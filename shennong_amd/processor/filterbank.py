"""Mel-filterbank features: Audio ---> FilterbankProcessor ---> Features

Same parameters, defaults and outputs as reference shennong/processor/filterbank.py:46-155; the
features come from the HIP backend (plan kind FBANK).
"""

from shennong_amd import _abi
from shennong_amd._options import FLAG, Option
from shennong_amd.processor.base import FeaturesProcessor, MelFeaturesProcessor


class FilterbankProcessor(MelFeaturesProcessor):
    """Mel-filterbank features"""
    _kind = _abi.KIND_FBANK
    name = 'filterbank'

    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True, num_bins=23, low_freq=20,
                 high_freq=0, vtln_low=100, vtln_high=-500,
                 use_energy=False, energy_floor=0.0, raw_energy=True,
                 htk_compat=False, use_log_fbank=True, use_power=True):
        FeaturesProcessor.__init__(self)
        self._configure(locals())

    use_energy = Option(
        'use_energy', 'Add an extra dimension with energy to the filterbank output', FLAG)
    energy_floor = Option('energy_floor', 'Floor on energy (absolute, not relative) in filterbanks')
    raw_energy = Option(
        'raw_energy', 'If true, compute energy before preemphasis and windowing', FLAG)
    htk_compat = Option('htk_compat', 'If True, put energy last', FLAG)
    use_log_fbank = Option(
        'use_log_fbank', 'If true, produce log-filterbank, else produce linear', FLAG)
    use_power = Option('use_power', 'If true, use power, else use magnitude', FLAG)

    @property
    def ndims(self):
        return self.num_bins + int(self.use_energy)
