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


# ---- round 6: post-processing families against third-party code of the build image (scipy, scikit-learn) ----------
# Float64 libraries by other authors, no code in common with oracle/kaldi_oracle.c or oracle/spec_f64.py; each
# states a textbook operation that Kaldi's routine equals by definition (reference file:line beside each).
def _matrix(seed, frames, dim, scale=8.0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((frames, dim)) * scale + rng.uniform(-20, 20, size=dim)).astype(np.float32)


def _delta_by_scipy(x, order, window):
    """Kaldi ComputeDeltas (reference postprocessor/delta.py:129-131) as ONE correlation per
    order with edge replication: `scipy.ndimage.correlate1d(mode='nearest')` clamps t + j to [0, T - 1]"""
    from scipy.ndimage import correlate1d
    base = np.arange(-window, window + 1, dtype=np.float64)
    base /= np.sum(base ** 2)
    kernel, blocks = np.ones(1), []
    for _ in range(order + 1):
        blocks.append(correlate1d(x.astype(np.float64), kernel, axis=0, mode='nearest') if kernel.size > 1
                      else x.astype(np.float64))
        kernel = np.convolve(kernel, base)
    return np.hstack(blocks)


DELTA_CASES = [(1, 2), (2, 2), (2, 3), (3, 1), (2, 5)]


@pytest.mark.parametrize('order, window', DELTA_CASES)
def test_oracle_delta_against_scipy(order, window):
    for frames in (1, 3, 9, 140):      # (shorter than the kernel: the clamped edges overlap)
        x = _matrix(order * 10 + window + frames, frames, 13)
        got = orc.deltas(x, order, window)
        assert_close(got, _delta_by_scipy(x, order, window).astype(np.float32), rtol=1e-5, what='delta vs scipy')


def _cmvn_by_sklearn(xs, weights=None):
    """Kaldi's CMVN with variance normalisation (reference postprocessor/cmvn.py:180-282: mean and population
    variance of the accumulated frames, weighted) is scikit-learn's StandardScaler with `sample_weight`"""
    from sklearn.preprocessing import StandardScaler
    scaler = StandardScaler()
    stacked = np.concatenate(xs).astype(np.float64)
    scaler.fit(stacked, sample_weight=None if weights is None else np.concatenate(weights).astype(np.float64))
    return [scaler.transform(x.astype(np.float64)) for x in xs], scaler


def test_oracle_cmvn_against_sklearn():
    pytest.importorskip('sklearn')
    xs = [_matrix(5, 140, 13), _matrix(6, 77, 13), _matrix(7, 1, 13)]
    rng = np.random.default_rng(8)
    for weights in (None, [(rng.random(x.shape[0]) > 0.4).astype(np.float32) for x in xs]):
        stats = np.zeros((2, 14))
        for k, x in enumerate(xs):
            orc.cmvn_accumulate(x, weights=None if weights is None else weights[k], stats=stats)
        want, scaler = _cmvn_by_sklearn(xs, weights)
        count = sum(x.shape[0] for x in xs) if weights is None else float(sum(w.sum() for w in weights))
        assert stats[0, -1] == pytest.approx(count)
        np.testing.assert_allclose(stats[0, :-1] / stats[0, -1], scaler.mean_, rtol=1e-9)
        for x, w in zip(xs, want):
            np.testing.assert_allclose(orc.cmvn_apply(x, stats), w, rtol=1e-4, atol=2e-5)


def _sliding_by_scipy(x, window, normalize_variance):
    """interior frames of Kaldi's centred SlidingWindowCmn (reference postprocessor/cmvn.py:382-470: the mean - and
    the variance - of the `window` frames around t, t - window / 2 ... t + window / 2 - 1) as a moving average:
    `scipy.ndimage.uniform_filter1d` with its origin moved so that the window starts at t - window / 2"""
    from scipy.ndimage import uniform_filter1d
    x = x.astype(np.float64)
    # (scipy's window of an even size is t - size / 2 ... t + size / 2 - 1 and of an odd size t -+ size // 2: Kaldi's)
    mean = uniform_filter1d(x, size=window, axis=0, mode='nearest')
    out = x - mean
    if normalize_variance:
        var = uniform_filter1d(x * x, size=window, axis=0, mode='nearest') - mean ** 2
        out = out / np.sqrt(np.maximum(var, 1e-10))
    half = window // 2
    return out, slice(half, x.shape[0] - (window - half) + 1)


@pytest.mark.parametrize('window, normalize_variance', [(60, False), (60, True), (101, False)])
def test_oracle_sliding_cmvn_against_scipy(window, normalize_variance):
    x = _matrix(21, 400, 13)
    got = orc.sliding_cmn(x, center=True, cmn_window=window, min_window=20, normalize_variance=normalize_variance)
    want, interior = _sliding_by_scipy(x, window, normalize_variance)
    assert interior.stop - interior.start > 250
    np.testing.assert_allclose(got[interior], want[interior], rtol=1e-4, atol=2e-4)


def _rasta_by_scipy(x):
    """the RASTA filter of a log-domain trajectory [frames, bands] (reference processor/plp.py:100-168, whose
    own test replays rasta_py's lfilter form, test/processor/test_plp.py:94-124): numerator 0.2, 0.1, 0, -0.1,
    -0.2, one pole at 0.94; the first four outputs are zero while the FIR part charges from a state that
    `lfilter_zi` scales by the first frame, the pole takes part from the fifth frame on"""
    import scipy.signal
    numer = np.array([0.2, 0.1, 0.0, -0.1, -0.2])
    out = np.zeros_like(x, dtype=np.float64)
    for band in range(x.shape[1]):
        column = x[:, band].astype(np.float64)
        state = scipy.signal.lfilter_zi(numer, [1.0]) * column[0]
        _, state = scipy.signal.lfilter(numer, [1.0], column[:4], zi=state)
        out[4:, band], _ = scipy.signal.lfilter(numer, [1.0, -0.94], column[4:], zi=state)
    return out


def test_oracle_rasta_against_scipy():
    frames = 80
    t = np.arange(frames)
    x = np.stack([np.sin(2 * np.pi * t / 16.0), np.random.default_rng(2).random(frames),
                  (t == 0).astype(np.float64), np.linspace(-3, 5, frames)], axis=1).astype(np.float32)
    got = orc.rasta(x, do_log=False)
    np.testing.assert_allclose(got, _rasta_by_scipy(x), rtol=1e-5, atol=1e-6)
    assert np.all(got[:4] == 0)


@pytest.mark.gpu
def test_hip_post_processing_against_third_parties(gpu):
    """the same third-party statements against the HIP path (delta, CMVN +- weights, sliding CMVN kernels through
    the post-processor classes)"""
    pytest.importorskip('sklearn')
    from shennong_amd import Features
    from shennong_amd.postprocessor import (
        CmvnPostProcessor, DeltaPostProcessor, SlidingWindowCmvnPostProcessor)

    def feats(x):
        return Features(x, np.arange(x.shape[0], dtype=np.float64) * 0.01,
                        properties={'pipeline': [{'name': 'mfcc', 'columns': [0, x.shape[1] - 1]}], 'mfcc': {}})
    for order, window in DELTA_CASES:
        for frames in (1, 3, 9, 140):
            x = _matrix(order * 10 + window + frames, frames, 13)
            got = DeltaPostProcessor(order=order, window=window).process(feats(x)).data
            assert_close(got, _delta_by_scipy(x, order, window).astype(np.float32), rtol=1e-5, what='delta vs scipy')
    xs = [_matrix(5, 140, 13), _matrix(6, 77, 13), _matrix(7, 1, 13)]
    rng = np.random.default_rng(8)
    for weights in (None, [(rng.random(x.shape[0]) > 0.4).astype(np.float32) for x in xs]):
        cmvn = CmvnPostProcessor(13)
        for k, x in enumerate(xs):
            cmvn.accumulate(feats(x), weights=None if weights is None else weights[k])
        want, scaler = _cmvn_by_sklearn(xs, weights)
        np.testing.assert_allclose(cmvn.stats[0, :-1] / cmvn.count, scaler.mean_, rtol=1e-9)
        for x, w in zip(xs, want):
            np.testing.assert_allclose(cmvn.process(feats(x)).data, w, rtol=1e-4, atol=2e-5)
    x = _matrix(21, 400, 13)
    for window, normalize_variance in ((60, False), (60, True), (101, False)):
        got = SlidingWindowCmvnPostProcessor(center=True, cmn_window=window, min_window=20,
                                             normalize_variance=normalize_variance).process(feats(x)).data
        want, interior = _sliding_by_scipy(x, window, normalize_variance)
        np.testing.assert_allclose(got[interior], want[interior], rtol=1e-4, atol=2e-4)


# ---- the PLP tail (reference processor/plp.py:548-626) from third-party primitives ---------------------------------
# mel energies: the oracle's LINEAR filterbank outputs - the family the HuggingFace front end pins above - then, in
# float64: equal loudness (a closed formula of the bins' centre frequencies), cube root (np.cbrt), autocorrelation =
# the IDFT of the duplicated-edge spectrum = scipy's DCT-I / (2 (n - 1)), LPC = the solution of the Toeplitz normal
# equations (scipy.linalg.solve_toeplitz instead of Kaldi's Durbin recursion), LPC -> cepstrum = the cepstrum of the
# all-pole filter 1 / A(z) read off a 4096-point FFT of log(1 / A) (instead of the recursion of plp.py:149-168), lifter.
def _plp_tail_by_scipy(mel, sample_rate, lpc_order, num_ceps, lifter, low_freq=20.0):
    import scipy.fft
    import scipy.linalg
    nbins = mel.shape[1]
    to_mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)   # noqa: E731
    step = (to_mel(0.5 * sample_rate) - to_mel(low_freq)) / (nbins + 1)
    centre = 700.0 * (np.exp((to_mel(low_freq) + step * np.arange(1, nbins + 1)) / 1127.0) - 1.0)
    fsq = centre ** 2
    loudness = (fsq / (fsq + 1.6e5)) ** 2 * ((fsq + 1.44e6) / (fsq + 9.61e6))
    x = np.cbrt(mel.astype(np.float64) * loudness)
    x = np.concatenate([x[:, :1], x, x[:, -1:]], axis=1)                      # first and last bins duplicated
    autocorr = scipy.fft.dct(x, type=1, axis=1)[:, :lpc_order + 1] / (2.0 * (x.shape[1] - 1))
    out = np.zeros((mel.shape[0], num_ceps))
    for t, r in enumerate(autocorr):
        a = scipy.linalg.solve_toeplitz(r[:lpc_order], -r[1:lpc_order + 1])   # R a = -r
        residual = r[0] + a @ r[1:lpc_order + 1]
        spectrum = np.fft.fft(np.concatenate([[1.0], a]), 4096)
        cepstrum = np.real(np.fft.ifft(-np.log(spectrum)))                    # of 1 / A(z), minimum phase
        out[t, 0] = max(np.log(residual), np.finfo(float).eps)
        out[t, 1:] = cepstrum[1:num_ceps]
    if lifter:
        out *= 1.0 + 0.5 * lifter * np.sin(np.pi * np.arange(num_ceps) / lifter)
    return out


PLP_CASES = [dict(), dict(num_ceps=9, lpc_order=10), dict(cepstral_lifter=0), dict(num_bins=30, lpc_order=14)]


def _plp_case(params, wave, compute_plp, compute_fbank):
    from shennong_amd.processor import PlpProcessor
    nbins = params.get('num_bins', 23)
    plp = PlpProcessor(dither=0, use_energy=False, **params)
    bank = FilterbankProcessor(dither=0, num_bins=nbins, use_log_fbank=False)
    want = _plp_tail_by_scipy(compute_fbank(bank, wave), 16000, plp.lpc_order, plp.num_ceps, plp.cepstral_lifter)
    got = compute_plp(plp, wave)
    assert got.shape == want.shape
    # (float32 Durbin on float32 mel energies against float64 linear algebra: the parity tolerance plus an absolute
    # term for cepstra near a zero crossing - measured need 9.4e-6 on values up to 6, asserted at twice that)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize('params', PLP_CASES)
def test_oracle_plp_tail_against_scipy(params):
    _plp_case(params, _wave('test.wav'), lambda p, w: orc.compute(p._build_options(), w),
              lambda p, w: orc.compute(p._build_options(), w))


@pytest.mark.gpu
@pytest.mark.parametrize('params', PLP_CASES)
def test_hip_plp_tail_against_scipy(gpu, params):
    from shennong_amd import Audio
    _plp_case(params, _wave('test.wav'), lambda p, w: p.process(Audio(w, 16000)).data,
              lambda p, w: p.process(Audio(w, 16000)).data)
