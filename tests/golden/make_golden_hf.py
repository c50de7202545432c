#!/usr/bin/env python
"""Writes tests/golden/third_party_hf.npz: log-mel filterbanks and power spectra of tests/golden/test.wav /
test.8k.wav computed by a THIRD-PARTY restatement of Kaldi's front end - `transformers.audio_utils`
(HuggingFace transformers, numpy, float64): `spectrogram(..., preemphasis, remove_dc_offset, window, power)` +
`mel_filter_bank(..., mel_scale="kaldi", triangularize_in_mel_space=True)` + `window_function("povey")`, the
code path its feature extractors use in place of `torchaudio.compliance.kaldi.fbank` (against which that
project tests it; torchaudio in turn is tested against Kaldi's binaries).

What this pins and what it does not.  It is not Kaldi and it is not the reference; it is an implementation of the
same published algorithm by other authors, with no code in common with oracle/kaldi_oracle.c or
oracle/spec_f64.py.  Agreement (tests/test_third_party_pin.py: 1e-4) covers ExtractWindow for snip_edges,
ProcessWindow's order (DC removal, pre-emphasis with x[0] -= c x[0], window), the povey / hanning / hamming /
rectangular windows, the 512- and 256-point power and magnitude spectra, Kaldi's mel scale and its triangles
built in the mel domain (bin width sr / N_fft, the Nyquist bin unused), low / high cut-offs, the FLT_EPSILON floor
and the log.  Two MFCC cases add scipy's orthonormal DCT-II (= Kaldi's ComputeDctMatrix) on the library's 23 log-mel energies,
with and without the cepstral lifter.  Not covered: dither, snip_edges = False (the library pads differently),
energies, VTLN, PLP, pitch.

Run where transformers is installed (it is in the build image; version recorded in the file):

    cd /tmp && python /root/repo/tests/golden/make_golden_hf.py
"""
import json
import os

import numpy as np
from scipy.io import wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
FLT_EPSILON = 1.192092955078125e-07

# name -> (wav, what this side must build, what the library is asked for)
#   ours: processor kind + constructor arguments of shennong_amd (and of the reference)
#   hf:   window name, frame / hop / fft lengths, power, preemphasis, remove_dc_offset, mel (bins, fmin, fmax), log
CASES = {
    'fbank23_povey': ('test.wav', ('filterbank', {'num_bins': 23}), {}),
    'fbank40_povey': ('test.wav', ('filterbank', {'num_bins': 40}), {}),
    'fbank80_povey': ('test.wav', ('filterbank', {'num_bins': 80}), {}),
    'fbank40_hanning': ('test.wav', ('filterbank', {'num_bins': 40, 'window_type': 'hanning'}), {'window': 'hann'}),
    'fbank40_hamming': ('test.wav', ('filterbank', {'num_bins': 40, 'window_type': 'hamming'}), {'window': 'hamming'}),
    'fbank40_rectangular': ('test.wav', ('filterbank', {'num_bins': 40, 'window_type': 'rectangular'}),
                            {'window': 'boxcar'}),
    'fbank40_no_preemphasis': ('test.wav', ('filterbank', {'num_bins': 40, 'preemph_coeff': 0.0}),
                               {'preemphasis': None}),
    'fbank40_no_dc_removal': ('test.wav', ('filterbank', {'num_bins': 40, 'remove_dc_offset': False}),
                              {'remove_dc_offset': False}),
    'fbank40_band_100_7600': ('test.wav', ('filterbank', {'num_bins': 40, 'low_freq': 100, 'high_freq': -400}),
                              {'fmin': 100.0, 'fmax': 7600.0}),
    'fbank40_linear': ('test.wav', ('filterbank', {'num_bins': 40, 'use_log_fbank': False}), {'log': None}),
    'fbank40_magnitude': ('test.wav', ('filterbank', {'num_bins': 40, 'use_power': False}), {'power': 1.0}),
    'fbank40_20ms_5ms': ('test.wav', ('filterbank', {'num_bins': 40, 'frame_length': 0.02, 'frame_shift': 0.005}),
                         {'frame': 320, 'hop': 80}),
    'fbank40_8k': ('test.8k.wav', ('filterbank', {'num_bins': 40, 'sample_rate': 8000}),
                   {'frame': 200, 'hop': 80, 'fft': 256, 'rate': 8000, 'fmax': 4000.0}),
    'fbank23_8k_hamming': ('test.8k.wav', ('filterbank', {'num_bins': 23, 'sample_rate': 8000,
                                                         'window_type': 'hamming'}),
                           {'frame': 200, 'hop': 80, 'fft': 256, 'rate': 8000, 'fmax': 4000.0, 'window': 'hamming'}),
    # power spectrum, bins 1 .. N/2 (column 0 of the reference's spectrogram is the frame energy)
    'spectrogram_povey': ('test.wav', ('spectrogram', {}), {'mel': None}),
    'spectrogram_8k_hanning': ('test.8k.wav', ('spectrogram', {'sample_rate': 8000, 'window_type': 'hanning'}),
                               {'mel': None, 'frame': 200, 'hop': 80, 'fft': 256, 'rate': 8000, 'window': 'hann'}),
}


def main():
    import transformers
    from transformers.audio_utils import mel_filter_bank, spectrogram, window_function
    out, meta = {}, {}
    for name, (wav, ours, hf) in CASES.items():
        rate, wave = wavfile.read(os.path.join(HERE, wav))
        assert rate == hf.get('rate', 16000) and wave.dtype == np.int16
        frame, hop, fft = hf.get('frame', 400), hf.get('hop', 160), hf.get('fft', 512)
        window = window_function(frame, hf.get('window', 'povey'), periodic=False)
        mel = None
        if hf.get('mel', 'kaldi') is not None:
            # (fft // 2 + 1 frequency bins: the library's bin width is sr / (2 (bins - 1)) = sr / N_fft)
            mel = mel_filter_bank(
                num_frequency_bins=fft // 2 + 1, num_mel_filters=ours[1]['num_bins'],
                min_frequency=hf.get('fmin', 20.0), max_frequency=hf.get('fmax', 8000.0), sampling_rate=rate,
                norm=None, mel_scale='kaldi', triangularize_in_mel_space=True)
        result = spectrogram(
            wave.astype(np.float64), window, frame_length=frame, hop_length=hop, fft_length=fft,
            power=hf.get('power', 2.0), center=False, preemphasis=hf.get('preemphasis', 0.97), mel_filters=mel,
            mel_floor=FLT_EPSILON, log_mel=hf.get('log', 'log') if mel is not None else None,
            remove_dc_offset=hf.get('remove_dc_offset', True), dtype=np.float64).T
        if mel is None:
            result = np.log(np.maximum(result, FLT_EPSILON))[:, 1:]     # bins 1 .. N/2, Kaldi's floor
        out[name] = np.ascontiguousarray(result, dtype=np.float32)   # (1e-7 relative: far below the 1e-4 of the comparison)
        meta[name] = {'wav': wav, 'processor': ours[0], 'params': dict(ours[1], dither=0),
                      'columns': 'bins 1..' if mel is None else 'all'}
    # MFCC without the energy column = Kaldi's cepstral lifter x the orthonormal DCT-II of the 23 log-mel energies:
    # the DCT is scipy's (norm='ortho' is Kaldi's ComputeDctMatrix: row 0 sqrt(1/N), row k sqrt(2/N) cos(pi/N (n+1/2) k)),
    # the log-mel energies are the library's; only the lifter 1 + Q/2 sin(pi i / Q), Q = 22, is written here
    from scipy.fft import dct
    lifter = 1.0 + 0.5 * 22.0 * np.sin(np.pi * np.arange(13) / 22.0)
    for name, source, params in (('mfcc13_no_energy', 'fbank23_povey', {}),
                                 ('mfcc13_no_energy_no_lifter', 'fbank23_povey', {'cepstral_lifter': 0.0})):
        rate, wave = wavfile.read(os.path.join(HERE, 'test.wav'))
        mel = mel_filter_bank(num_frequency_bins=257, num_mel_filters=23, min_frequency=20.0, max_frequency=8000.0,
                              sampling_rate=16000, norm=None, mel_scale='kaldi', triangularize_in_mel_space=True)
        logmel = spectrogram(wave.astype(np.float64), window_function(400, 'povey', periodic=False), frame_length=400,
                             hop_length=160, fft_length=512, power=2.0, center=False, preemphasis=0.97,
                             mel_filters=mel, mel_floor=FLT_EPSILON, log_mel='log', remove_dc_offset=True,
                             dtype=np.float64).T
        ceps = dct(logmel, type=2, norm='ortho', axis=1)[:, :13]
        if params.get('cepstral_lifter', 22.0):
            ceps = ceps * lifter
        out[name] = np.ascontiguousarray(ceps, dtype=np.float32)
        meta[name] = {'wav': 'test.wav', 'processor': 'mfcc', 'columns': 'all',
                      'params': dict(params, use_energy=False, dither=0)}
    out['__cases__'] = np.frombuffer(json.dumps(meta, sort_keys=True).encode(), dtype=np.uint8)
    out['__versions__'] = np.frombuffer(json.dumps(
        {'transformers': transformers.__version__, 'numpy': np.__version__}).encode(), dtype=np.uint8)
    target = os.path.join(HERE, 'third_party_hf.npz')
    np.savez_compressed(target, **out)
    print('wrote %s: %d cases, transformers %s' % (target, len(meta), transformers.__version__))


if __name__ == '__main__':
    main()
