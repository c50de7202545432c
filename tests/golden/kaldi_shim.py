"""numpy stand-ins for the ~25 pykaldi primitives that the reference's PLP path calls, used
ONLY by make_golden_plp.py in the build container to run the reference's own Python control flow
(`PlpProcessor._compute` / `_compute_frame` / `_extract_window` / `_process_window`,
shennong/processor/plp.py:171-260, :510-626) without Kaldi.

What this pins and what it does not: the GLUE is the reference's - the order of the operations, which
energy is used when, the floors, the slicing of the cepstrum, the RASTA insertion point, the HTK reorder,
the Python-float (double) arithmetic that leaks in between float32 vectors.  The PRIMITIVES are stand-ins:
float32 storage like Kaldi's BaseFloat vectors, each operation evaluated in float64 from the float64
restatement (oracle/spec_f64.py) and rounded once when it is stored - within float32 round-off of Kaldi's
own kernels, not bit-identical to them.
"""
import sys
import types

import numpy as np

sys.path.insert(0, '/root/repo')
from oracle import spec_f64  # noqa: E402


class Vector:
    def __init__(self, arg=0):
        if isinstance(arg, np.ndarray):
            self.a = arg           # a view: SubVector semantics
        else:
            self.a = np.zeros(int(arg), dtype=np.float32)

    # -- shape
    @property
    def dim(self):
        return self.a.shape[0]

    def resize_(self, n, kind=None):
        self.a = np.zeros(int(n), dtype=np.float32)

    def numpy(self):
        return self.a

    def __len__(self):
        return self.a.shape[0]

    # -- element access: an index gives a Python float, a slice a view
    def __getitem__(self, i):
        if isinstance(i, slice):
            return Vector(self.a[i])
        return float(self.a[i])

    def __setitem__(self, i, v):
        if isinstance(v, Vector):
            v = v.a
        self.a[i] = v

    # -- arithmetic (float64 evaluation, one rounding at the store)
    def set_zero_(self):
        self.a[:] = 0

    def sum(self):
        return float(np.float32(self.a.astype(np.float64).sum()))

    def add_(self, c):
        self.a[:] = self.a.astype(np.float64) + np.float64(np.float32(c))

    def scale_(self, c):
        self.a[:] = self.a.astype(np.float64) * np.float64(np.float32(c))

    def mul_elements_(self, other):
        self.a[:] = self.a.astype(np.float64) * other.a.astype(np.float64)

    def apply_pow_(self, p):
        self.a[:] = self.a.astype(np.float64) ** np.float64(np.float32(p))

    def add_mat_vec_(self, alpha, mat, trans, vec, beta):
        self.a[:] = alpha * (mat.a.astype(np.float64) @ vec.a.astype(np.float64)) + beta * self.a.astype(np.float64)


class SubVector(Vector):
    def __init__(self, data):
        self.a = np.asarray(data).astype(np.float32)   # (int16 samples widened to float32, like pykaldi)


class Matrix:
    def __init__(self, rows=0, cols=0):
        self.a = np.zeros((int(rows), int(cols)), dtype=np.float32)

    def __getitem__(self, key):
        return Vector(self.a[key])

    def numpy(self):
        return self.a


class FrameExtractionOptions:
    def __init__(self):
        self.samp_freq, self.frame_shift_ms, self.frame_length_ms = 16000.0, 10.0, 25.0
        self.dither, self.preemph_coeff, self.remove_dc_offset = 1.0, 0.97, True
        self.window_type, self.round_to_power_of_two, self.blackman_coeff = 'povey', True, 0.42
        self.snip_edges = True

    def __setattr__(self, k, v):
        if k in ('samp_freq', 'frame_shift_ms', 'frame_length_ms', 'dither', 'preemph_coeff', 'blackman_coeff'):
            v = float(np.float32(v))     # BaseFloat fields
        object.__setattr__(self, k, v)

    def window_shift(self):
        return int(self.samp_freq * 0.001 * self.frame_shift_ms)

    def window_size(self):
        return int(self.samp_freq * 0.001 * self.frame_length_ms)

    def padded_window_size(self):
        n = self.window_size()
        if not self.round_to_power_of_two:
            return n
        p = 1
        while p < n:
            p *= 2
        return p


class MelBanksOptions:
    def __init__(self):
        self.num_bins, self.low_freq, self.high_freq, self.vtln_low, self.vtln_high = 23, 20.0, 0.0, 100.0, -500.0


class PlpOptions:
    def __init__(self):
        self.frame_opts, self.mel_opts = FrameExtractionOptions(), MelBanksOptions()
        self.lpc_order, self.num_ceps, self.use_energy, self.energy_floor, self.raw_energy = 12, 13, True, 0.0, True
        self.compress_factor, self.cepstral_lifter, self.cepstral_scale, self.htk_compat = 1.0 / 3.0, 22, 1.0, False

    def __setattr__(self, k, v):
        if k in ('energy_floor', 'compress_factor', 'cepstral_scale'):
            v = float(np.float32(v))
        object.__setattr__(self, k, v)


class FeatureWindowFunction:
    @classmethod
    def from_options(cls, opts):
        self = cls()
        self.window = Vector(opts.window_size())
        self.window.a[:] = spec_f64.window_function(opts.window_size(), opts.window_type, opts.blackman_coeff)
        return self


class MelBanks:
    def __init__(self, mel_opts, frame_opts, vtln_warp):
        self.w, self.centers = spec_f64.mel_banks_vtln(
            mel_opts.num_bins, frame_opts.samp_freq, frame_opts.padded_window_size(), mel_opts.low_freq,
            mel_opts.high_freq, mel_opts.vtln_low, mel_opts.vtln_high, float(vtln_warp))
        self.w = self.w.astype(np.float32)   # Kaldi stores float32 weights

    def compute(self, power_spectrum, mel_energies_out):
        n = self.w.shape[1]
        mel_energies_out.a[:] = self.w.astype(np.float64) @ power_spectrum.a[:n].astype(np.float64)


def _num_frames(nsamples, opts, flush=True):
    return spec_f64.num_frames(nsamples, opts.window_shift(), opts.window_size(), opts.snip_edges)


def _first_sample_of_frame(frame, opts):
    if opts.snip_edges:
        return frame * opts.window_shift()
    return opts.window_shift() * frame + opts.window_shift() // 2 - opts.window_size() // 2


def _preemphasize(window, coeff):
    x = window.a.astype(np.float64)
    c = np.float64(np.float32(coeff))
    y = x.copy()
    y[1:] = x[1:] - c * x[:-1]
    y[0] = x[0] - c * x[0]
    window.a[:] = y


def _dither(window, value):
    raise AssertionError('the fixtures use dither = 0')


def _real_fft(v, forward):
    assert forward
    n = v.dim
    spec = np.fft.rfft(v.a.astype(np.float64))
    out = np.empty(n)
    out[0], out[1] = spec[0].real, spec[n // 2].real
    out[2::2], out[3::2] = spec[1:n // 2].real, spec[1:n // 2].imag
    v.a[:] = out


def _compute_power_spectrum(v):
    n = v.dim
    x = v.a.astype(np.float64)
    p = np.empty(n // 2 + 1)
    p[0], p[n // 2] = x[0] * x[0], x[1] * x[1]
    p[1:n // 2] = x[2::2] ** 2 + x[3::2] ** 2
    v.a[:n // 2 + 1] = p


def _init_idft_bases(n_bases, dim):
    m = Matrix(n_bases, dim)
    m.a[:] = spec_f64.idft_bases(n_bases, dim)
    return m


def _compute_lifter_coeffs(q, vec):
    vec.a[:] = 1 + 0.5 * q * np.sin(np.pi * np.arange(vec.dim) / q)


def _get_equal_loudness_vector(banks):
    v = Vector(len(banks.centers))
    v.a[:] = spec_f64.equal_loudness(banks.centers)
    return v


def _compute_lpc(autocorr, lpc_out):
    lpc, e = spec_f64.durbin(autocorr.a.astype(np.float64))
    lpc_out.a[:] = lpc
    return float(np.float32(-np.log(1.0 / np.float64(np.float32(e)))))


def install():
    """puts the stand-ins where `import kaldi...` finds them"""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    common = mod('kaldi.matrix.common',
                 MatrixResizeType=types.SimpleNamespace(UNDEFINED=0, SET_ZERO=1, COPY_DATA=2),
                 MatrixTransposeType=types.SimpleNamespace(NO_TRANS=0, TRANS=1))
    functions = mod('kaldi.matrix.functions', real_fft=_real_fft,
                    vec_vec=lambda a, b: float(np.float32(np.dot(a.a.astype(np.float64), b.a.astype(np.float64)))))
    matrix = mod('kaldi.matrix', Vector=Vector, SubVector=SubVector, Matrix=Matrix, common=common,
                 functions=functions)
    bmath = mod('kaldi.base.math', log=lambda x: float(np.log(np.float64(x))))
    base = mod('kaldi.base', math=bmath)
    window = mod('kaldi.feat.window', FrameExtractionOptions=FrameExtractionOptions,
                 FeatureWindowFunction=FeatureWindowFunction, num_frames=_num_frames,
                 first_sample_of_frame=_first_sample_of_frame, preemphasize=_preemphasize, dither=_dither)
    mel = mod('kaldi.feat.mel', MelBanksOptions=MelBanksOptions, MelBanks=MelBanks,
              compute_lifter_coeffs=_compute_lifter_coeffs, get_equal_loudness_vector=_get_equal_loudness_vector,
              compute_lpc=_compute_lpc)
    plp = mod('kaldi.feat.plp', PlpOptions=PlpOptions)
    ffunc = mod('kaldi.feat.functions', init_idft_bases=_init_idft_bases,
                compute_power_spectrum=_compute_power_spectrum)
    feat = mod('kaldi.feat', window=window, mel=mel, plp=plp, functions=ffunc)
    mod('kaldi', matrix=matrix, base=base, feat=feat)
