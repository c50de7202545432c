"""The oracle (CPU) and the HIP path (`-m gpu`) against a THIRD-PARTY restatement of Kaldi's filterbank front
end: `transformers.audio_utils` (HuggingFace, numpy), the code its feature extractors run in place of
`torchaudio.compliance.kaldi.fbank`.  tests/golden/make_golden_hf.py made the fixture and says what the
agreement covers (framing, DC removal, pre-emphasis, windows, power / magnitude spectrum, Kaldi's mel banks,
floor, log; 16 kHz and 8 kHz) and what it does not (dither, centred frames, energies, VTLN, DCT, PLP, pitch).
Not Kaldi and not the reference - an implementation by other authors with no code in common with
oracle/kaldi_oracle.c or oracle/spec_f64.py; a misreading of Kaldi would have to be shared by three parties now."""
import json
import os

import numpy as np
import pytest
from scipy.io import wavfile

from conftest import assert_close
from oracle import oracle as orc
from shennong_amd.processor import FilterbankProcessor, MfccProcessor, SpectrogramProcessor

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = np.load(os.path.join(HERE, 'golden', 'third_party_hf.npz'))
CASES = json.loads(bytes(DATA['__cases__']).decode())
CLASSES = {'filterbank': FilterbankProcessor, 'spectrogram': SpectrogramProcessor, 'mfcc': MfccProcessor}


def _wave(name):
    return np.ascontiguousarray(wavfile.read(os.path.join(HERE, 'golden', name))[1], dtype=np.int16)


def _check(name, got):
    case, want = CASES[name], DATA[name]
    if case['columns'] != 'all':
        got = got[:, 1:]           # column 0 of the spectrogram is the frame energy
    assert got.shape == want.shape, (name, got.shape, want.shape)
    linear = case['params'].get('use_log_fbank', True) is False
    assert_close(np.asarray(got, dtype=np.float32), want, rtol=1e-4, atol=1.0 if linear else None,
                 what='third party %s' % name,
                 family={'spectrogram': 'spectrogram', 'mfcc': 'mfcc'}.get(case['processor'], 'fbank'))


def test_fixture_is_what_the_generator_lists():
    import ast
    tree = ast.parse(open(os.path.join(HERE, 'golden', 'make_golden_hf.py')).read())
    listed = next(ast.literal_eval(n.value) for n in tree.body
                  if isinstance(n, ast.Assign) and n.targets[0].id == 'CASES')
    assert set(listed) <= set(CASES) and len(CASES) >= 18
    versions = json.loads(bytes(DATA['__versions__']).decode())
    assert 'transformers' in versions


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_against_third_party(name):
    case = CASES[name]
    proc = CLASSES[case['processor']](**case['params'])
    _check(name, orc.compute(proc._build_options(), _wave(case['wav'])))


def test_third_party_library_still_says_the_same():
    """the fixture against the library itself when it is installed (it is in the build image): a change of its
    behaviour shows up here, not as a silent drift of the pin"""
    au = pytest.importorskip('transformers.audio_utils')
    wave = _wave('test.wav').astype(np.float64)
    mel = au.mel_filter_bank(num_frequency_bins=257, num_mel_filters=40, min_frequency=20.0, max_frequency=8000.0,
                             sampling_rate=16000, norm=None, mel_scale='kaldi', triangularize_in_mel_space=True)
    live = au.spectrogram(wave, au.window_function(400, 'povey', periodic=False), frame_length=400, hop_length=160,
                          fft_length=512, power=2.0, center=False, preemphasis=0.97, mel_filters=mel,
                          mel_floor=1.192092955078125e-07, log_mel='log', remove_dc_offset=True, dtype=np.float64).T
    np.testing.assert_allclose(live, DATA['fbank40_povey'], rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(CASES))
def test_hip_path_against_third_party(gpu, name):
    from shennong_amd import Audio
    case = CASES[name]
    proc = CLASSES[case['processor']](**case['params'])
    rate = case['params'].get('sample_rate', 16000)
    _check(name, proc.process(Audio(_wave(case['wav']), rate)).data)
