"""Energy of window frames: Audio ---> EnergyProcessor ---> Features

Mirror of reference shennong/processor/energy.py:55-185 over the HIP backend (plan kind ENERGY:
float64 sum of squares of the Kaldi-processed window, floored, then compressed).  The reference
returns float64 data computed from float32 windows; this backend returns the same quantity
rounded once to float32 (what every Kaldi consumer downstream, e.g. the VAD, works in).
"""

from shennong_amd import _abi
from shennong_amd.features import Features
from shennong_amd.processor.base import FramesProcessor, batch_features, check_signal


class EnergyProcessor(FramesProcessor):
    """Energy (log, sqrt or raw) of the audio frames"""
    def __init__(self, sample_rate=16000, frame_shift=0.01,
                 frame_length=0.025, dither=1.0, preemph_coeff=0.97,
                 remove_dc_offset=True, window_type='povey',
                 round_to_power_of_two=True, blackman_coeff=0.42,
                 snip_edges=True, raw_energy=True, compression='log'):
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
        self._compression = 'log'
        self.compression = compression
        self.raw_energy = raw_energy

    @property
    def name(self):
        return 'energy'

    @property
    def ndims(self):
        return 1

    @property
    def compression(self):
        """Type of energy compression: 'off', 'log' or 'sqrt'"""
        return self._compression

    @compression.setter
    def compression(self, value):
        if value not in _abi.COMPRESSION:
            raise ValueError(
                'compression must be in {}, it is {}'.format(
                    ', '.join(_abi.COMPRESSION.keys()), value))
        self._compression = value

    @property
    def raw_energy(self):
        """If true, compute energy before preemphasis and windowing"""
        return self._raw_energy

    @raw_energy.setter
    def raw_energy(self, value):
        self._raw_energy = value

    def _build_options(self):
        opts = self._options(_abi.KIND_ENERGY)
        opts.raw_energy = 1 if self.raw_energy else 0
        opts.compression = _abi.COMPRESSION[self._compression]
        return opts

    def process(self, signal):
        """Computes energy on the input `signal` -> Features [nframes, 1]"""
        check_signal(self, signal)
        data = self._run(self._build_options(), [signal])[0]
        if data.shape[0] == 0:
            data = data.reshape((0, 1))
        return Features(
            data, self.times(data.shape[0]), properties=self.get_properties())

    def _process_batch(self, signals):
        for signal in signals:
            check_signal(self, signal)
        datas = self._run(self._build_options(), signals)
        return batch_features([d if d.shape[0] else d.reshape((0, 1)) for d in datas],
                              self.times, lambda _: self.get_properties())
