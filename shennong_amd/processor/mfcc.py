"""MFCC features: Audio ---> MfccProcessor ---> Features

Same parameters, defaults and outputs as reference shennong/processor/mfcc.py:46-159; the features
come from the HIP backend (plan kind MFCC).
"""

from shennong_amd import _abi
from shennong_amd._options import FLAG, Option
from shennong_amd.processor.base import FeaturesProcessor, MelFeaturesProcessor


class MfccProcessor(MelFeaturesProcessor):
    """Mel Frequency Cepstral Coeficients"""
    _kind = _abi.KIND_MFCC
    name = 'mfcc'

    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True, num_bins=23, low_freq=20,
                 high_freq=0, vtln_low=100, vtln_high=-500,
                 num_ceps=13, use_energy=True, energy_floor=0.0,
                 raw_energy=True, cepstral_lifter=22.0,
                 htk_compat=False):
        FeaturesProcessor.__init__(self)
        self._configure(locals())

    num_ceps = Option(
        'num_ceps', 'Number of cepstra in MFCC computation (including C0), <= `num_bins`')
    use_energy = Option('use_energy', 'Use energy (instead of C0) in MFCC computation', FLAG)
    energy_floor = Option(
        'energy_floor', 'Floor on energy (absolute, not relative) in MFCC computation')
    raw_energy = Option(
        'raw_energy', 'If true, compute energy before preemphasis and windowing', FLAG)
    cepstral_lifter = Option('cepstral_lifter', 'Constant that controls scaling of MFCCs')
    htk_compat = Option(
        'htk_compat', 'If True, put energy or C0 last and use a factor of sqrt(2) on C0', FLAG)

    @property
    def ndims(self):
        return self.num_ceps

    def process_with_deltas(self, signal, vtln_warp=1.0):
        """``DeltaPostProcessor().process(self.process(signal, vtln_warp))`` through ONE plan and one
        call: rows are [cepstra | delta | delta-delta] (order 2, window 2); the cepstra stay in HBM
        between the MFCC kernel and the delta kernel (two launches: the one-launch form of round 2
        measured 13 % slower and is kept behind SNF_FUSED_DELTA=1; reference chain:
        postprocessor/delta.py:129-131 after processor/mfcc.py:86; BASELINE config 3)"""
        return self._process_batch_with_deltas([signal], vtln_warp=[vtln_warp])[0]

    def _process_batch_with_deltas(self, signals, vtln_warp=None):
        from shennong_amd.postprocessor.delta import DeltaPostProcessor
        from shennong_amd.processor.base import batch_features, check_signal
        for signal in signals:
            check_signal(self, signal)
        warps = [1.0] * len(signals) if vtln_warp is None else list(vtln_warp)
        opts = self._build_options()
        opts.append_deltas, opts.delta_order, opts.delta_window = 1, 2, 2
        datas = self._run(opts, signals, warps)
        delta = DeltaPostProcessor(order=2, window=2)

        def properties(warp):
            stage = type('Stage', (), {'properties': self.get_properties(vtln_warp=warp),
                                       'ndims': self.ndims})
            return delta.get_properties(stage)
        return batch_features(datas, self.times, properties, warps)
