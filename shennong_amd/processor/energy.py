"""Energy of window frames: Audio ---> EnergyProcessor ---> Features

Same parameters and outputs as reference shennong/processor/energy.py:55-185 (plan kind ENERGY:
float64 sum of squares of the Kaldi-processed window, floored, then compressed).  The reference
returns float64 data computed from float32 windows; this backend returns the same quantity
rounded once to float32 (what every Kaldi consumer downstream, e.g. the VAD, works in).

Known divergence: the reference hands `signal.data` to Kaldi at its native scale, whereas every
plan of this backend takes 16-bit samples, so non-int16 audio is first rescaled to the int16 full
scale (`Audio.astype`, an exact power of two).  For float audio in [-1, 1] the log-energy is therefore
2 ln(2^15) = 20.79 higher than the reference's (the energy itself 2^30 times), which matters to
absolute thresholds such as the VAD's; int16 audio - what the pipeline and the reference's own tests
use - is unaffected.  A warning is logged when it happens.
"""

import numpy as np


from shennong_amd import _abi
from shennong_amd._options import FLAG, Option
from shennong_amd.processor.base import (
    FeaturesProcessor, FramesProcessor, batch_features, check_signal)


class EnergyProcessor(FramesProcessor):
    """Energy (log, sqrt or raw) of the audio frames"""
    _kind = _abi.KIND_ENERGY
    name = 'energy'
    ndims = 1

    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True, raw_energy=True, compression='log'):
        FeaturesProcessor.__init__(self)
        self._configure(locals())

    raw_energy = Option(
        'raw_energy', 'If true, compute energy before preemphasis and windowing', FLAG)

    @property
    def compression(self):
        """Type of energy compression: 'off', 'log' or 'sqrt'"""
        code = self._record.compression
        return next(name for name, value in _abi.COMPRESSION.items() if value == code)

    @compression.setter
    def compression(self, value):
        if value not in _abi.COMPRESSION:
            raise ValueError(
                'compression must be in {}, it is {}'.format(
                    ', '.join(_abi.COMPRESSION.keys()), value))
        self._record.compression = _abi.COMPRESSION[value]

    def process(self, signal):
        """Computes energy on the input `signal` -> Features [nframes, 1]"""
        return self._process_batch([signal])[0]

    def _wrap_pinned(self, datas):
        # (an utterance without frames comes back as Kaldi's (0, 0) matrix: one column here)
        return batch_features([d if d.shape[0] else d.reshape((0, 1)) for d in datas],
                              self.times, lambda _: self.get_properties())

    def _process_batch(self, signals):
        for signal in signals:
            check_signal(self, signal)
        if any(signal.dtype != np.int16 for signal in signals):
            self.log.warning(
                'energy of non-int16 audio is computed after rescaling to the int16 full scale '
                '(the reference keeps the native scale: see the module documentation)')
        datas = self._run(self._build_options(), signals)
        # (an utterance without frames comes back as Kaldi's (0, 0) matrix: one column here)
        return batch_features([d if d.shape[0] else d.reshape((0, 1)) for d in datas],
                              self.times, lambda _: self.get_properties())
