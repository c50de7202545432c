/*
 * kaldi_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Scalar float32 restatement of the algorithms that bootphon/shennong's hot path executes through
 * pykaldi (Kaldi src/feat, src/matrix — a third-party dependency that is NOT vendored in
 * /root/reference: conda package `shennong-pykaldi`, version unpinned, reference environment.yml:7)
 * and of the reference's own Python PLP / RASTA recipe (shennong/processor/plp.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product (shennong_amd/) never does.
 *
 * PARITY STATUS: "parity unpinned" for coefficient VALUES of spectrogram / fbank / MFCC / PLP /
 * pitch / delta — the reference's tests hold no golden coefficient vectors and Kaldi cannot be run
 * here.  What IS pinned (tests/test_oracle_pins.py): frame counts for every shape
 * the reference tests assert, the window known answers of shennong/window.py:43-49, the
 * Frames/boundaries literals of test/test_frames.py, the energy identity of
 * test/processor/test_energy.py:36-44, the exp(50) known answer of test_plp.py:77-81, the htk_compat
 * rules of test_mfcc.py:100-111 / test_plp.py:50-61, RASTA and lpc2cepstrum against fixtures
 * generated from the reference's runnable numpy code (tests/golden/make_golden.py), an independent
 * float64 numpy restatement of EVERY family (oracle/spec_f64.py: fbank / MFCC / spectrogram / pitch,
 * and since round 4 PLP + RASTA, VTLN-warped banks, delta, CMVN, sliding CMVN, pitch post-processing;
 * tests/test_spec_f64.py, profiles/r04_f64_report.txt), and the PLP recipe against the reference's OWN
 * control flow (plp.py:171-260, :510-626) run over numpy stand-ins of the pykaldi primitives
 * (tests/golden/make_golden_plp.py: glue pinned, primitives are stand-ins).  The float32 round-off of
 * Kaldi's own kernels remains unpinned.
 *
 * THIRD-PARTY PIN (round 5): tests/test_third_party_pin.py compares this file's fbank / spectrogram / MFCC (18
 * cases: 23 / 40 / 80 bins, four windows, with and without pre-emphasis and DC removal, band limits, linear and
 * magnitude banks, 8 kHz, other frame lengths, DCT +- lifter) with HuggingFace transformers' numpy restatement of
 * Kaldi's front end (transformers.audio_utils 5.15.0, the stand-in of torchaudio.compliance.kaldi.fbank in its
 * feature extractors) + scipy's orthonormal DCT-II, at 1e-4: an implementation by other authors, tested upstream
 * against torchaudio (itself tested against Kaldi's binaries).  Not Kaldi, not the reference; PLP, pitch, delta,
 * energies, VTLN and centred frames have no such counterpart.
 *
 * HOW TO PIN IT (round 5): tests/golden/make_golden_kaldi.py, run where bootphon/shennong and its pykaldi are
 * installed (never here, never on the GPU box), writes tests/golden/reference_kaldi.npz - the reference's
 * own outputs for 30 cases on tests/golden/test.wav / test.8k.wav (spectrogram, fbank-23/40, MFCC, PLP
 * +- RASTA, energy, pitch, pitch post-processing, delta, VAD, CMVN +- VAD weights, sliding CMVN; dither 0).
 * tests/test_kaldi_pin.py compares this file's functions (and, under -m gpu, the HIP path) with it at 1e-4
 * as soon as the file exists, and skips with that explanation until then.
 *
 * Each function cites the reference file:line it follows, or [KALDI-UPSTREAM] + the Kaldi source
 * file when the algorithm lives in Kaldi (restated from the published sources).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA contraction, IEEE float32 ops)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include <pthread.h>

#include "../include/shennong_amd.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#define M_2PI 6.283185307179586476925286766559005
#ifndef M_SQRT2
#define M_SQRT2 1.4142135623730950488016887
#endif

#define ORC_API __attribute__((visibility("default")))

static __thread char g_err[512];
static int orc_fail(const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return -2;
}
ORC_API const char* orc_last_error(void) { return g_err; }

/* ------------------------------------------------------------------------------------------ */
/* BLAS-like float32 primitives (Kaldi VecVec / Sum: BaseFloat accumulation)                   */
/* ------------------------------------------------------------------------------------------ */
static float vecvec(const float* a, const float* b, int n) {
  float s = 0.0f;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}
static float vecsum(const float* a, int n) {
  float s = 0.0f;
  for (int i = 0; i < n; i++) s += a[i];
  return s;
}
/* Summation orders of the pitch tracker.  Kaldi hands these sums to BLAS (cblas_sdot / sgemv), whose
 * order and use of fused multiply-adds depend on the library build, so NO order is "the" Kaldi one;
 * the oracle fixes one that a 16-lane group of a GPU wavefront reproduces bit for bit, and the HIP
 * kernels (csrc/kernels_pitch.hip) implement exactly the same:
 *   - chain_dot:  s = fmaf(a[i], b[i], s) for i ascending, from 0 (lag correlations, FIR taps)
 *   - tree16:     16 partial sums p[0..15] added as a balanced binary tree of neighbours
 *                 ((p0+p1)+(p2+p3)) + ((p4+p5)+(p6+p7)) ... (frame mean, frame energy, norm average),
 *                 where p[l] runs over the elements l, l+16, l+32, ... ascending */
static float chain_dot(const float* a, const float* b, int n) {
  float s = 0.0f;
  for (int i = 0; i < n; i++) s = fmaf(a[i], b[i], s);
  return s;
}
static float tree16(const float* p) {
  float a[8], b[4];
  for (int k = 0; k < 8; k++) a[k] = p[2 * k] + p[2 * k + 1];
  for (int k = 0; k < 4; k++) b[k] = a[2 * k] + a[2 * k + 1];
  return (b[0] + b[1]) + (b[2] + b[3]);
}
static float strided16_sum(const float* x, int n) {
  float p[16];
  for (int l = 0; l < 16; l++) {
    float s = 0.0f;
    for (int i = l; i < n; i += 16) s += x[i];
    p[l] = s;
  }
  return tree16(p);
}
static float strided16_sumsq(const float* x, int n) {
  float p[16];
  for (int l = 0; l < 16; l++) {
    float s = 0.0f;
    for (int i = l; i < n; i += 16) s = fmaf(x[i], x[i], s);
    p[l] = s;
  }
  return tree16(p);
}
/* Whole-signal sums (tens of thousands of terms): Kaldi hands these to BLAS sdot, whose blocked
 * SIMD accumulation is far more accurate than a sequential float loop; a double accumulator
 * rounded to float is the closest portable stand-in. */
static float vecvec_long(const float* a, const float* b, int64_t n) {
  double s = 0.0;
  for (int64_t i = 0; i < n; i++) s += (double)a[i] * (double)b[i];
  return (float)s;
}
static float vecsum_long(const float* a, int64_t n) {
  double s = 0.0;
  for (int64_t i = 0; i < n; i++) s += (double)a[i];
  return (float)s;
}

/* ------------------------------------------------------------------------------------------ */
/* Framing  [KALDI-UPSTREAM feature-window.h/.cc: FrameExtractionOptions, NumFrames,           */
/*           FirstSampleOfFrame]; wrapped by reference frames.py:137, plp.py:218,517           */
/* ------------------------------------------------------------------------------------------ */
ORC_API int32_t orc_window_shift(const snf_frame_options* o) {
  return (int32_t)((double)o->samp_freq * 0.001 * (double)o->frame_shift_ms);
}
ORC_API int32_t orc_window_size(const snf_frame_options* o) {
  return (int32_t)((double)o->samp_freq * 0.001 * (double)o->frame_length_ms);
}
static int32_t round_up_pow2(int32_t n) {
  int32_t p = 1;
  while (p < n) p <<= 1;
  return p;
}
ORC_API int32_t orc_padded_window_size(const snf_frame_options* o) {
  int32_t w = orc_window_size(o);
  return o->round_to_power_of_two ? round_up_pow2(w) : w;
}
ORC_API int64_t orc_num_frames(const snf_frame_options* o, int64_t num_samples) {
  int64_t shift = orc_window_shift(o), len = orc_window_size(o);
  if (shift <= 0) return 0;
  if (o->snip_edges) {
    if (num_samples < len) return 0;
    return 1 + (num_samples - len) / shift;
  }
  return (num_samples + shift / 2) / shift; /* flush == true */
}
ORC_API int64_t orc_first_sample_of_frame(const snf_frame_options* o, int64_t frame) {
  int64_t shift = orc_window_shift(o);
  if (o->snip_edges) return frame * shift;
  int64_t mid = shift * frame + shift / 2;
  return mid - orc_window_size(o) / 2;
}

/* [KALDI-UPSTREAM feature-window.cc: FeatureWindowFunction]; formulas in reference window.py:6-38 */
ORC_API int orc_window_function(const snf_frame_options* o, float* w) {
  int32_t n = orc_window_size(o);
  double a = M_2PI / (n - 1);
  for (int32_t i = 0; i < n; i++) {
    double x = (double)i;
    switch (o->window_type) {
      case SNF_WINDOW_HANNING: w[i] = (float)(0.5 - 0.5 * cos(a * x)); break;
      case SNF_WINDOW_HAMMING: w[i] = (float)(0.54 - 0.46 * cos(a * x)); break;
      case SNF_WINDOW_POVEY: w[i] = (float)pow(0.5 - 0.5 * cos(a * x), 0.85); break;
      case SNF_WINDOW_RECTANGULAR: w[i] = 1.0f; break;
      case SNF_WINDOW_BLACKMAN:
        w[i] = (float)((double)o->blackman_coeff - 0.5 * cos(a * x) +
                       (0.5 - (double)o->blackman_coeff) * cos(2 * a * x));
        break;
      default: return orc_fail("invalid window type");
    }
  }
  return 0;
}

/* [KALDI-UPSTREAM feature-window.cc: ExtractWindow + ProcessWindow]; in-tree restatement at
 * reference plp.py:203-260 (ExtractWindow, reflection loop :242-254) and plp.py:171-200
 * (ProcessWindow order: dither -> DC -> raw log-energy -> pre-emphasis -> window).
 * `eps_energy` is FLT_EPSILON for the Kaldi computers and float64 eps for shennong's PLP
 * (reference plp.py:193).  Dither is not reproducible (C rand()) and must be 0 here. */
static void extract_window(const snf_frame_options* o, const float* wave, int64_t n, int64_t frame,
                           const float* wfn, float* win /* padded */, double* log_energy,
                           double eps_energy, int use_double_log) {
  int32_t len = orc_window_size(o), padded = orc_padded_window_size(o);
  int64_t start = orc_first_sample_of_frame(o, frame);
  if (start >= 0 && start + len <= n) {
    for (int32_t s = 0; s < len; s++) win[s] = wave[start + s];
  } else {
    for (int32_t s = 0; s < len; s++) {
      int64_t k = s + start;
      while (k < 0 || k >= n) {
        if (k < 0) k = -k - 1;
        else k = 2 * n - 1 - k;
      }
      win[s] = wave[k];
    }
  }
  for (int32_t s = len; s < padded; s++) win[s] = 0.0f;
  /* ProcessWindow on win[0:len] */
  if (o->remove_dc_offset) {
    float m = -vecsum(win, len) / (float)len;
    for (int32_t s = 0; s < len; s++) win[s] += m;
  }
  if (log_energy) {
    float e = vecvec(win, win, len);
    if (use_double_log) {
      double ed = (double)e > eps_energy ? (double)e : eps_energy;
      *log_energy = log(ed);
    } else {
      float ef = e > (float)eps_energy ? e : (float)eps_energy;
      *log_energy = (double)logf(ef);
    }
  }
  if (o->preemph_coeff != 0.0f) {
    float c = o->preemph_coeff;
    for (int32_t i = len - 1; i > 0; i--) win[i] -= c * win[i - 1];
    win[0] -= c * win[0];
  }
  for (int32_t s = 0; s < len; s++) win[s] *= wfn[s];
}

/* exported single-frame probe (used by tests to pin ExtractWindow reflection behaviour) */
ORC_API int orc_extract_window(const snf_frame_options* o, const float* wave, int64_t n,
                               int64_t frame, float* out_padded, float* raw_log_energy) {
  int32_t len = orc_window_size(o);
  float* wfn = (float*)malloc(sizeof(float) * (size_t)(len > 0 ? len : 1));
  if (orc_window_function(o, wfn)) { free(wfn); return -2; }
  double le = 0;
  extract_window(o, wave, n, frame, wfn, out_padded, &le, FLT_EPSILON, 0);
  if (raw_log_energy) *raw_log_energy = (float)le;
  free(wfn);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Real FFT, packed Kaldi layout [Re0, Re(N/2), Re1, Im1, ...]                                  */
/* [KALDI-UPSTREAM matrix-functions.cc RealFft / srfft.cc SplitRadixRealFft — same output       */
/*  convention; rounding differs at the 1e-7 level between any two float32 FFT algorithms]      */
/* ------------------------------------------------------------------------------------------ */
static void complex_fft_pow2(float* d /* re,im interleaved */, int m) {
  /* iterative radix-2 DIT, forward sign exp(-i...) */
  for (int i = 1, j = 0; i < m; i++) {
    int bit = m >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      float tr = d[2 * i], ti = d[2 * i + 1];
      d[2 * i] = d[2 * j]; d[2 * i + 1] = d[2 * j + 1];
      d[2 * j] = tr; d[2 * j + 1] = ti;
    }
  }
  for (int len = 2; len <= m; len <<= 1) {
    int half = len >> 1;
    for (int k = 0; k < half; k++) {
      double ang = -M_2PI * (double)k / (double)len;
      float wr = (float)cos(ang), wi = (float)sin(ang);
      for (int s = k; s < m; s += len) {
        int t = s + half;
        float xr = d[2 * t] * wr - d[2 * t + 1] * wi;
        float xi = d[2 * t] * wi + d[2 * t + 1] * wr;
        d[2 * t] = d[2 * s] - xr; d[2 * t + 1] = d[2 * s + 1] - xi;
        d[2 * s] += xr; d[2 * s + 1] += xi;
      }
    }
  }
}

static void real_fft(float* v, int n) {
  if (n >= 2 && (n & (n - 1)) == 0) {
    int m = n / 2;
    if (m > 1) complex_fft_pow2(v, m);
    /* unpack: A_k = C_k + W^k D_k  (see RealFft in Kaldi matrix-functions.cc) */
    for (int k = 1; 2 * k <= m; k++) {
      double ang = -M_2PI * (double)k / (double)n;
      float kr = (float)cos(ang), ki = (float)sin(ang);
      float ck_re = 0.5f * (v[2 * k] + v[n - 2 * k]);
      float ck_im = 0.5f * (v[2 * k + 1] - v[n - 2 * k + 1]);
      float dk_re = 0.5f * (v[2 * k + 1] + v[n - 2 * k + 1]);
      float dk_im = -0.5f * (v[2 * k] - v[n - 2 * k]);
      v[2 * k] = ck_re + (dk_re * kr - dk_im * ki);
      v[2 * k + 1] = ck_im + (dk_re * ki + dk_im * kr);
      int kd = m - k;
      if (kd != k) {
        /* conj(C), conj(D), twiddle (-kr, ki) */
        v[2 * kd] = ck_re + (dk_re * (-kr) - (-dk_im) * ki);
        v[2 * kd + 1] = -ck_im + (dk_re * ki + (-dk_im) * (-kr));
      }
    }
    float z = v[0] + v[1], h = v[0] - v[1];
    v[0] = z; v[1] = h;
  } else {
    /* general even N: direct DFT in double (Kaldi uses a mixed-radix complex FFT here) */
    double* tmp = (double*)malloc(sizeof(double) * (size_t)n);
    for (int k = 0; k <= n / 2; k++) {
      double re = 0, im = 0;
      for (int t = 0; t < n; t++) {
        double ang = -M_2PI * (double)((int64_t)k * t % n) / (double)n;
        re += v[t] * cos(ang);
        im += v[t] * sin(ang);
      }
      if (k == 0) tmp[0] = re;
      else if (k == n / 2) tmp[1] = re;
      else { tmp[2 * k] = re; tmp[2 * k + 1] = im; }
    }
    for (int t = 0; t < n; t++) v[t] = (float)tmp[t];
    free(tmp);
  }
}

/* [KALDI-UPSTREAM feature-functions.cc ComputePowerSpectrum]; wrapped at reference plp.py:575 */
static void power_spectrum(float* v, int n) {
  int half = n / 2;
  float first = v[0] * v[0], last = v[1] * v[1];
  for (int i = 1; i < half; i++) {
    float re = v[2 * i], im = v[2 * i + 1];
    v[i] = re * re + im * im;
  }
  v[0] = first;
  v[half] = last;
}

/* ------------------------------------------------------------------------------------------ */
/* Mel banks [KALDI-UPSTREAM mel-computations.cc: MelBanks::MelBanks, VtlnWarpFreq, Compute]    */
/* wrapped at reference plp.py:491-492, :580                                                    */
/* ------------------------------------------------------------------------------------------ */
static float mel_scale(float f) { return 1127.0f * logf(1.0f + f / 700.0f); }
static float inv_mel_scale(float m) { return 700.0f * (expf(m / 1127.0f) - 1.0f); }

static float vtln_warp_freq(float vtln_low, float vtln_high, float low, float high, float warp,
                            float freq) {
  if (freq < low || freq > high) return freq;
  float one = 1.0f;
  float l = vtln_low * (warp > one ? warp : one);
  float h = vtln_high * (warp < one ? warp : one);
  float scale = 1.0f / warp;
  float Fl = scale * l, Fh = scale * h;
  float scale_left = (Fl - low) / (l - low);
  float scale_right = (high - Fh) / (high - h);
  if (freq < l) return low + scale_left * (freq - low);
  else if (freq < h) return scale * freq;
  else return high + scale_right * (freq - high);
}
static float vtln_warp_mel(float vl, float vh, float low, float high, float warp, float mel) {
  return mel_scale(vtln_warp_freq(vl, vh, low, high, warp, inv_mel_scale(mel)));
}

typedef struct {
  int num_bins, num_fft_bins;
  int* first;
  int* size;
  float* w; /* [num_bins][num_fft_bins] dense, zero outside support */
  float* center;
} melbanks_t;

static void melbanks_free(melbanks_t* m) {
  free(m->first); free(m->size); free(m->w); free(m->center);
  memset(m, 0, sizeof(*m));
}

static int melbanks_init(melbanks_t* m, const snf_mel_options* mo, const snf_frame_options* fo,
                         float warp) {
  memset(m, 0, sizeof(*m));
  int nb = mo->num_bins;
  if (nb < 3) return orc_fail("Must have at least 3 mel bins");
  float sf = fo->samp_freq;
  int padded = orc_padded_window_size(fo);
  if (padded % 2 != 0) return orc_fail("padded window size must be even");
  int nfft = padded / 2;
  float nyq = 0.5f * sf;
  float low = mo->low_freq, high;
  if (mo->high_freq > 0.0f) high = mo->high_freq;
  else high = nyq + mo->high_freq;
  if (low < 0.0f || low >= nyq || high <= 0.0f || high > nyq || high <= low)
    return orc_fail("Bad values in options: low-freq / high-freq vs. nyquist");
  float bin_width = sf / (float)padded;
  float mel_low = mel_scale(low), mel_high = mel_scale(high);
  float delta = (mel_high - mel_low) / (float)(nb + 1);
  float vl = mo->vtln_low, vh = mo->vtln_high;
  if (vh < 0.0f) vh += nyq;
  if (warp != 1.0f &&
      (vl < 0.0f || vl <= low || vl >= high || vh <= 0.0f || vh >= high || vh <= vl))
    return orc_fail("Bad values in options: vtln-low / vtln-high versus low-freq / high-freq");
  m->num_bins = nb; m->num_fft_bins = nfft;
  m->first = (int*)calloc((size_t)nb, sizeof(int));
  m->size = (int*)calloc((size_t)nb, sizeof(int));
  m->w = (float*)calloc((size_t)nb * (size_t)nfft, sizeof(float));
  m->center = (float*)calloc((size_t)nb, sizeof(float));
  for (int b = 0; b < nb; b++) {
    float left = mel_low + (float)b * delta, center = mel_low + (float)(b + 1) * delta,
          right = mel_low + (float)(b + 2) * delta;
    if (warp != 1.0f) {
      left = vtln_warp_mel(vl, vh, low, high, warp, left);
      center = vtln_warp_mel(vl, vh, low, high, warp, center);
      right = vtln_warp_mel(vl, vh, low, high, warp, right);
    }
    m->center[b] = inv_mel_scale(center);
    int first = -1, last = -1;
    for (int i = 0; i < nfft; i++) {
      float freq = bin_width * (float)i;
      float mel = mel_scale(freq);
      if (mel > left && mel < right) {
        float weight;
        if (mel <= center) weight = (mel - left) / (center - left);
        else weight = (right - mel) / (right - center);
        m->w[(size_t)b * nfft + i] = weight;
        if (first == -1) first = i;
        last = i;
      }
    }
    if (first == -1) {
      melbanks_free(m);
      return orc_fail("You may have set num_bins too large (empty mel bin)");
    }
    m->first[b] = first;
    m->size[b] = last + 1 - first;
  }
  return 0;
}

static void melbanks_compute(const melbanks_t* m, const float* ps, float* out) {
  for (int b = 0; b < m->num_bins; b++)
    out[b] = vecvec(m->w + (size_t)b * m->num_fft_bins + m->first[b], ps + m->first[b], m->size[b]);
}

/* exported for tests: dense weights [num_bins, padded/2], first/size, centre frequencies */
ORC_API int orc_mel_banks(const snf_mel_options* mo, const snf_frame_options* fo, float warp,
                          int32_t* first, int32_t* size, float* weights, float* center) {
  melbanks_t m;
  int rc = melbanks_init(&m, mo, fo, warp);
  if (rc) return rc;
  for (int b = 0; b < m.num_bins; b++) {
    if (first) first[b] = m.first[b];
    if (size) size[b] = m.size[b];
    if (center) center[b] = m.center[b];
  }
  if (weights) memcpy(weights, m.w, sizeof(float) * (size_t)m.num_bins * (size_t)m.num_fft_bins);
  melbanks_free(&m);
  return 0;
}

/* [KALDI-UPSTREAM matrix-functions.cc ComputeDctMatrix; mel-computations.cc ComputeLifterCoeffs] */
static void dct_matrix(float* M, int K, int N) {
  float normalizer = (float)sqrt(1.0 / (double)(float)N);
  for (int j = 0; j < N; j++) M[j] = normalizer;
  normalizer = (float)sqrt(2.0 / (double)(float)N);
  for (int k = 1; k < K; k++)
    for (int n = 0; n < N; n++)
      M[(size_t)k * N + n] = (float)((double)normalizer * cos(M_PI / N * (n + 0.5) * k));
}
static void lifter_coeffs(float Q, float* c, int n) {
  for (int i = 0; i < n; i++) c[i] = (float)(1.0 + 0.5 * (double)Q * sin(M_PI * i / (double)Q));
}
ORC_API void orc_dct_matrix(float* M, int K, int N) { dct_matrix(M, K, N); }
ORC_API void orc_lifter_coeffs(float Q, float* c, int n) { lifter_coeffs(Q, c, n); }

/* ------------------------------------------------------------------------------------------ */
/* PLP helpers                                                                                  */
/* ------------------------------------------------------------------------------------------ */
/* [KALDI-UPSTREAM mel-computations.cc GetEqualLoudnessVector]; wrapped at reference plp.py:506 */
static void equal_loudness(const melbanks_t* m, float* out) {
  for (int i = 0; i < m->num_bins; i++) {
    float fsq = m->center[i] * m->center[i];
    float fsub = (float)((double)fsq / ((double)fsq + 1.6e5));
    out[i] = (float)((double)(fsub * fsub) * (((double)fsq + 1.44e6) / ((double)fsq + 9.61e6)));
  }
}
/* [KALDI-UPSTREAM feature-functions.cc InitIdftBases]; wrapped at reference plp.py:473-474 */
static void idft_bases(int n_bases, int dim, float* M) {
  float angle = (float)(M_PI / (double)(float)(dim - 1));
  float scale = (float)(1.0 / (2.0 * (double)(float)(dim - 1)));
  for (int i = 0; i < n_bases; i++) {
    M[(size_t)i * dim] = (float)(1.0 * (double)scale);
    float ifl = (float)i;
    for (int j = 1; j < dim - 1; j++) {
      float jfl = (float)j;
      M[(size_t)i * dim + j] = (float)(2.0 * (double)scale * cos((double)(angle * ifl * jfl)));
    }
    M[(size_t)i * dim + dim - 1] = (float)((double)scale * cos((double)(angle * ifl * (float)(dim - 1))));
  }
}
/* [KALDI-UPSTREAM mel-computations.cc Durbin / ComputeLpc]; wrapped at reference plp.py:601 */
static float compute_lpc(const float* ac, int n, float* lpc, float* tmp) {
  float E = ac[0];
  for (int i = 0; i < n; i++) {
    float ki = ac[i + 1];
    for (int j = 0; j < i; j++) ki += lpc[j] * ac[i - j];
    ki = ki / E;
    float c = 1 - ki * ki;
    if (c < 1.0e-5f) c = 1.0e-5f;
    E *= c;
    tmp[i] = -ki;
    for (int j = 0; j < i; j++) tmp[j] = lpc[j] - ki * lpc[i - j - 1];
    for (int j = 0; j <= i; j++) lpc[j] = tmp[j];
  }
  return (float)(-log(1.0 / (double)E)); /* -Log(1.0 / ans), double overload */
}
/* reference plp.py:149-168 (_lpc2cepstrum): Python-float (double) accumulation, float32 storage */
ORC_API void orc_lpc2cepstrum(int n, const float* lpc, float* cep) {
  for (int i = 0; i < n; i++) {
    double sum = 0.0;
    for (int j = 0; j < i; j++) sum += (double)(i - j) * (double)lpc[j] * (double)cep[i - j - 1];
    cep[i] = (float)(-(double)lpc[i] - sum / (double)(i + 1));
  }
}

/* reference plp.py:64-146 (RastaFilter): numerator -[-2..2]/10, denominator [1,-0.94], scipy
 * lfilter (direct form II transposed) in float64; first four frames output 0 in the log domain
 * and prime the FIR state with zi*x[0].  Input [T,B] float32 linear mel energies, do_log=True. */
static const double RASTA_NUM[5] = {0.2, 0.1, 0.0, -0.1, -0.2};
typedef struct { int size, count; double* z; float* first; } rasta_t;
static void rasta_init(rasta_t* r, int size) {
  r->size = size; r->count = 0;
  r->z = (double*)calloc((size_t)size * 4, sizeof(double));
  r->first = (float*)calloc((size_t)size * 4, sizeof(float));
}
static void rasta_free(rasta_t* r) { free(r->z); free(r->first); }
static void rasta_filter(rasta_t* r, const float* frame, float* out, int do_log) {
  int B = r->size;
  /* scipy.signal.lfilter_zi(num, 1) = [sum b[1:], sum b[2:], sum b[3:], b[4]] */
  /* back-substitution order of numpy.linalg.solve on the (I - A^T) bidiagonal system */
  double zi[4];
  zi[3] = RASTA_NUM[4];
  zi[2] = RASTA_NUM[3] + zi[3];
  zi[1] = RASTA_NUM[2] + zi[2];
  zi[0] = RASTA_NUM[1] + zi[1];
  for (int b = 0; b < B; b++) {
    float x = frame[b];
    if (do_log) x = logf(x + FLT_EPSILON);
    double y = 0.0;
    double* z = r->z + (size_t)b * 4;
    if (r->count < 4) {
      r->first[(size_t)r->count * B + b] = x;
      if (r->count == 3) {
        double x0 = (double)r->first[b];
        for (int k = 0; k < 4; k++) z[k] = zi[k] * x0;
        for (int t = 0; t < 4; t++) { /* FIR priming, a = [1] */
          double xt = (double)r->first[(size_t)t * B + b];
          z[0] = z[1] + xt * RASTA_NUM[1];
          z[1] = z[2] + xt * RASTA_NUM[2];
          z[2] = z[3] + xt * RASTA_NUM[3];
          z[3] = xt * RASTA_NUM[4];
        }
      }
      y = 0.0;
    } else {
      double xd = (double)x;
      y = z[0] + RASTA_NUM[0] * xd;
      z[0] = z[1] + xd * RASTA_NUM[1] - y * (-0.94);
      z[1] = z[2] + xd * RASTA_NUM[2];
      z[2] = z[3] + xd * RASTA_NUM[3];
      z[3] = xd * RASTA_NUM[4];
    }
    float yf = (float)y;
    out[b] = do_log ? expf(yf) : yf;
  }
  r->count++;
}
ORC_API int orc_rasta(const float* in, int64_t T, int B, float* out, int do_log) {
  rasta_t r; rasta_init(&r, B);
  for (int64_t t = 0; t < T; t++) rasta_filter(&r, in + t * B, out + t * B, do_log);
  rasta_free(&r);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Whole-utterance computers                                                                    */
/* [KALDI-UPSTREAM feature-common-inl.h OfflineFeatureTpl::Compute, feature-fbank.cc,           */
/*  feature-mfcc.cc, feature-spectrogram.cc]; reference call sites processor/base.py:429-431,   */
/*  spectrogram.py:138-140; PLP recipe reference plp.py:510-626; energy energy.py:148-186       */
/* ------------------------------------------------------------------------------------------ */
ORC_API int32_t orc_ndims(const snf_options* o) {
  switch (o->kind) {
    case SNF_KIND_SPECTROGRAM: return orc_padded_window_size(&o->frame) / 2 + 1;
    case SNF_KIND_FBANK: return o->mel.num_bins + (o->use_energy ? 1 : 0);
    case SNF_KIND_MFCC: return o->num_ceps;
    case SNF_KIND_PLP: return o->num_ceps;
    case SNF_KIND_PITCH: return 2;
    case SNF_KIND_ENERGY: return 1;
    default: return -1;
  }
}

ORC_API int orc_compute(const snf_options* o, const int16_t* wave16, int64_t n, float vtln_warp,
                        float* out) {
  const snf_frame_options* fo = &o->frame;
  if (fo->dither != 0.0f) return orc_fail("oracle requires dither == 0 (Kaldi's RandGauss is not reproducible)");
  int kind = o->kind;
  int len = orc_window_size(fo), padded = orc_padded_window_size(fo);
  int64_t T = orc_num_frames(fo, n);
  int D = orc_ndims(o);
  if (len < 2) return orc_fail("window size too small");
  /* [KALDI-UPSTREAM] RealFft / SplitRadixRealFft assert an even transform size (matrix-functions.cc);
     the frame energy (reference processor/energy.py) has no transform */
  if (padded % 2 != 0 && kind != SNF_KIND_ENERGY) return orc_fail("padded window size must be even");
  melbanks_t mb; memset(&mb, 0, sizeof(mb));
  int nb = o->mel.num_bins;
  float *dct = NULL, *lift = NULL, *eql = NULL, *idft = NULL;
  if (kind == SNF_KIND_FBANK || kind == SNF_KIND_MFCC || kind == SNF_KIND_PLP) {
    /* Kaldi constructs the computer (mel banks for warp 1.0, DCT...) before checking frames */
    if (kind == SNF_KIND_MFCC && o->num_ceps > nb)
      return orc_fail("num-ceps cannot be larger than num-mel-bins");
    /* ... the banks of a warp factor other than 1 are built lazily by the first frame
       ([KALDI-UPSTREAM] MfccComputer::GetMelBanks from Compute; reference plp.py:521-522 returns
       before _compute_frame): an utterance without frames never sees their option errors */
    /* ... and the reference's PLP builds ALL its banks that way (plp.py:482-494, called from _compute_frame
       only): no frames, no banks, no option errors; frames with a warp factor: that factor's banks only */
    int rc = 0;
    if (!(kind == SNF_KIND_PLP && T == 0))
      rc = melbanks_init(&mb, &o->mel, fo, kind == SNF_KIND_PLP ? vtln_warp : 1.0f);
    if (rc) return rc;
    if (kind != SNF_KIND_PLP && vtln_warp != 1.0f && T > 0) {
      melbanks_free(&mb);
      rc = melbanks_init(&mb, &o->mel, fo, vtln_warp);
      if (rc) return rc;
    }
  }
  if (kind == SNF_KIND_MFCC) {
    if (o->num_ceps <= 0) { melbanks_free(&mb); return orc_fail("num-ceps must be positive"); }
    dct = (float*)malloc(sizeof(float) * (size_t)nb * nb);
    dct_matrix(dct, nb, nb); /* first num_ceps rows used */
    lift = (float*)malloc(sizeof(float) * (size_t)o->num_ceps);
    if (o->cepstral_lifter != 0.0f) lifter_coeffs(o->cepstral_lifter, lift, o->num_ceps);
  }
  if (kind == SNF_KIND_PLP) {
    eql = (float*)malloc(sizeof(float) * (size_t)nb);
    if (T > 0) equal_loudness(&mb, eql);
    idft = (float*)malloc(sizeof(float) * (size_t)(o->lpc_order + 1) * (size_t)(nb + 2));
    idft_bases(o->lpc_order + 1, nb + 2, idft);
    lift = (float*)malloc(sizeof(float) * (size_t)o->num_ceps);
    if (o->cepstral_lifter != 0.0f) lifter_coeffs(o->cepstral_lifter, lift, o->num_ceps);
  }
  if (T == 0) { melbanks_free(&mb); free(dct); free(lift); free(eql); free(idft); return 0; }

  float* wave = (float*)malloc(sizeof(float) * (size_t)n);
  for (int64_t i = 0; i < n; i++) wave[i] = (float)wave16[i]; /* SubVector(int16) -> float32 */
  float* wfn = (float*)malloc(sizeof(float) * (size_t)len);
  snf_frame_options fo_local = *fo;
  if (kind == SNF_KIND_ENERGY && o->raw_energy) {
    /* reference energy.py:150-154: raw energy = no pre-emphasis, rectangular window */
    fo_local.preemph_coeff = 0.0f;
    fo_local.window_type = SNF_WINDOW_RECTANGULAR;
  }
  orc_window_function(&fo_local, wfn);
  float* win = (float*)malloc(sizeof(float) * (size_t)padded);
  float* mel = (float*)malloc(sizeof(float) * (size_t)(nb + 2 > 1 ? nb + 2 : 1));
  float* ac = NULL, *lpc = NULL, *tmp = NULL, *cep = NULL;
  rasta_t rasta; memset(&rasta, 0, sizeof(rasta));
  if (kind == SNF_KIND_PLP) {
    ac = (float*)malloc(sizeof(float) * (size_t)(o->lpc_order + 1));
    lpc = (float*)malloc(sizeof(float) * (size_t)o->lpc_order);
    tmp = (float*)malloc(sizeof(float) * (size_t)o->lpc_order);
    cep = (float*)malloc(sizeof(float) * (size_t)o->lpc_order);
    if (o->rasta) rasta_init(&rasta, nb);
  }
  int is_plp = (kind == SNF_KIND_PLP);
  double eps_e = is_plp ? DBL_EPSILON : (double)FLT_EPSILON;
  /* log floor: Kaldi computers: logf(energy_floor); shennong PLP: double log (plp.py:476-477) */
  double log_floor = 0.0;
  if (o->energy_floor > 0.0f)
    log_floor = is_plp ? log((double)o->energy_floor) : (double)logf(o->energy_floor);
  int need_raw;
  if (kind == SNF_KIND_SPECTROGRAM) need_raw = o->raw_energy;
  else need_raw = o->use_energy && o->raw_energy;

  for (int64_t t = 0; t < T; t++) {
    double log_energy = 0.0;
    float* row = out + t * D;
    if (kind == SNF_KIND_ENERGY) {
      extract_window(&fo_local, wave, n, t, wfn, win, NULL, eps_e, 0);
      /* reference energy.py:177-183: float64 sum of squares, floor at float64 tiny */
      double s = 0.0;
      for (int i = 0; i < len; i++) s += (double)win[i] * (double)win[i];
      if (s < DBL_MIN) s = DBL_MIN;
      double v = s;
      if (o->compression == SNF_COMPRESS_LOG) v = log(s);
      else if (o->compression == SNF_COMPRESS_SQRT) v = sqrt(s);
      row[0] = (float)v;
      continue;
    }
    extract_window(fo, wave, n, t, wfn, win, need_raw ? &log_energy : NULL, eps_e, is_plp);
    int post_energy = (kind == SNF_KIND_SPECTROGRAM) ? !o->raw_energy
                                                     : (o->use_energy && !o->raw_energy);
    if (post_energy) {
      float e = vecvec(win, win, padded);
      if (is_plp) log_energy = log((double)e > eps_e ? (double)e : eps_e);
      else log_energy = (double)logf(e > FLT_EPSILON ? e : FLT_EPSILON);
    }
    real_fft(win, padded);
    power_spectrum(win, padded);
    int nps = padded / 2 + 1;
    if (kind == SNF_KIND_SPECTROGRAM) {
      for (int i = 0; i < nps; i++) {
        float p = win[i] < FLT_EPSILON ? FLT_EPSILON : win[i];
        row[i] = logf(p);
      }
      if (o->energy_floor > 0.0f && log_energy < log_floor) log_energy = log_floor;
      row[0] = (float)log_energy;
    } else if (kind == SNF_KIND_FBANK) {
      if (!o->use_power)
        for (int i = 0; i < nps; i++) win[i] = powf(win[i], 0.5f);
      int off = (o->use_energy && !o->htk_compat) ? 1 : 0;
      melbanks_compute(&mb, win, row + off);
      if (o->use_log_fbank)
        for (int b = 0; b < nb; b++) {
          float v = row[off + b] < FLT_EPSILON ? FLT_EPSILON : row[off + b];
          row[off + b] = logf(v);
        }
      if (o->use_energy) {
        if (o->energy_floor > 0.0f && log_energy < log_floor) log_energy = log_floor;
        row[o->htk_compat ? nb : 0] = (float)log_energy;
      }
    } else if (kind == SNF_KIND_MFCC) {
      melbanks_compute(&mb, win, mel);
      for (int b = 0; b < nb; b++) {
        float v = mel[b] < FLT_EPSILON ? FLT_EPSILON : mel[b];
        mel[b] = logf(v);
      }
      for (int c = 0; c < o->num_ceps; c++) row[c] = vecvec(dct + (size_t)c * nb, mel, nb);
      if (o->cepstral_lifter != 0.0f)
        for (int c = 0; c < o->num_ceps; c++) row[c] *= lift[c];
      if (o->use_energy) {
        if (o->energy_floor > 0.0f && log_energy < log_floor) log_energy = log_floor;
        row[0] = (float)log_energy;
      }
      if (o->htk_compat) {
        float e = row[0];
        for (int c = 0; c < o->num_ceps - 1; c++) row[c] = row[c + 1];
        if (!o->use_energy) e = (float)((double)e * M_SQRT2);
        row[o->num_ceps - 1] = e;
      }
    } else { /* PLP, reference plp.py:548-626 */
      melbanks_compute(&mb, win, mel + 1);
      if (o->rasta) rasta_filter(&rasta, mel + 1, mel + 1, 1);
      for (int b = 0; b < nb; b++) mel[1 + b] *= eql[b];
      for (int b = 0; b < nb; b++) mel[1 + b] = powf(mel[1 + b], o->compress_factor);
      mel[0] = mel[1];
      mel[nb + 1] = mel[nb];
      for (int i = 0; i <= o->lpc_order; i++) ac[i] = vecvec(idft + (size_t)i * (nb + 2), mel, nb + 2);
      for (int i = 0; i < o->lpc_order; i++) lpc[i] = 0.0f;
      double res = (double)compute_lpc(ac, o->lpc_order, lpc, tmp);
      if (res < DBL_EPSILON) res = DBL_EPSILON; /* plp.py:603: max(., float64 eps) */
      orc_lpc2cepstrum(o->lpc_order, lpc, cep);
      for (int c = 1; c < o->num_ceps; c++) row[c] = cep[c - 1];
      row[0] = (float)res;
      if (o->cepstral_lifter != 0.0f)
        for (int c = 0; c < o->num_ceps; c++) row[c] *= lift[c];
      if (o->cepstral_scale != 1.0f)
        for (int c = 0; c < o->num_ceps; c++) row[c] *= o->cepstral_scale;
      if (o->use_energy) {
        if (o->energy_floor > 0.0f && log_energy < log_floor) log_energy = log_floor;
        row[0] = (float)log_energy;
      }
      if (o->htk_compat) {
        float e = row[0];
        for (int c = 0; c < o->num_ceps - 1; c++) row[c] = row[c + 1];
        row[o->num_ceps - 1] = e;
      }
    }
  }
  if (kind == SNF_KIND_PLP && o->rasta) rasta_free(&rasta);
  free(wave); free(wfn); free(win); free(mel); free(ac); free(lpc); free(tmp); free(cep);
  melbanks_free(&mb); free(dct); free(lift); free(eql); free(idft);
  return 0;
}

/* Batch driver for the CPU baseline: one utterance per task over `nthreads` POSIX threads, the
 * reference's own threading model (joblib threads over utterances, processor/base.py:104-107). */
typedef struct {
  const snf_options* o; const int16_t* wave; const int64_t* soff; const int64_t* foff;
  int64_t n_utts; int ndims; float* out; int tid, nthreads; int rc;
} batch_job_t;
static void* batch_worker(void* arg) {
  batch_job_t* j = (batch_job_t*)arg;
  for (int64_t u = j->tid; u < j->n_utts; u += j->nthreads) {
    int rc = orc_compute(j->o, j->wave + j->soff[u], j->soff[u + 1] - j->soff[u], 1.0f,
                         j->out + j->foff[u] * j->ndims);
    if (rc) j->rc = rc;
  }
  return NULL;
}
ORC_API int orc_compute_batch(const snf_options* o, const int16_t* wave, const int64_t* soff,
                              const int64_t* foff, int64_t n_utts, float* out, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)nthreads);
  batch_job_t* jobs = (batch_job_t*)malloc(sizeof(batch_job_t) * (size_t)nthreads);
  int ndims = orc_ndims(o), rc = 0;
  for (int t = 0; t < nthreads; t++) {
    batch_job_t j = {o, wave, soff, foff, n_utts, ndims, out, t, nthreads, 0};
    jobs[t] = j;
    if (nthreads == 1) batch_worker(&jobs[t]);
    else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
  }
  for (int t = 0; t < nthreads; t++) {
    if (nthreads > 1) pthread_join(th[t], NULL);
    if (jobs[t].rc) rc = jobs[t].rc;
  }
  free(th); free(jobs);
  return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Deltas [KALDI-UPSTREAM feature-functions.cc DeltaFeatures]; reference delta.py:129-131      */
/* ------------------------------------------------------------------------------------------ */
/* scales[i] has 2*i*window+1 entries; returns them concatenated (tests pin order-2 literals) */
ORC_API int orc_delta_scales(int order, int window, float* scales /* sum_i (2*i*window+1) */) {
  if (order < 0 || order >= 1000 || window <= 0 || window >= 1000) return orc_fail("bad delta options");
  float* prev = scales;
  prev[0] = 1.0f;
  int prev_dim = 1;
  for (int i = 1; i <= order; i++) {
    float* cur = prev + prev_dim;
    int cur_dim = prev_dim + 2 * window;
    for (int k = 0; k < cur_dim; k++) cur[k] = 0.0f;
    int prev_offset = (prev_dim - 1) / 2, cur_offset = prev_offset + window;
    float normalizer = 0.0f;
    for (int j = -window; j <= window; j++) {
      normalizer += (float)(j * j);
      for (int k = -prev_offset; k <= prev_offset; k++)
        cur[j + k + cur_offset] += (float)j * prev[k + prev_offset];
    }
    float s = (float)(1.0 / (double)normalizer);
    for (int k = 0; k < cur_dim; k++) cur[k] *= s;
    prev = cur; prev_dim = cur_dim;
  }
  return 0;
}

ORC_API int orc_deltas(int order, int window, const float* in, int64_t T, int D, float* out) {
  int total = 0;
  for (int i = 0; i <= order; i++) total += 2 * i * window + 1;
  float* scales = (float*)malloc(sizeof(float) * (size_t)total);
  int rc = orc_delta_scales(order, window, scales);
  if (rc) { free(scales); return rc; }
  int OD = D * (order + 1);
  for (int64_t t = 0; t < T; t++) {
    float* orow = out + t * OD;
    for (int c = 0; c < OD; c++) orow[c] = 0.0f;
    const float* sc = scales;
    for (int i = 0; i <= order; i++) {
      int dim = 2 * i * window + 1, max_off = (dim - 1) / 2;
      for (int j = -max_off; j <= max_off; j++) {
        int64_t f = t + j;
        if (f < 0) f = 0; else if (f >= T) f = T - 1;
        float s = sc[j + max_off];
        if (s != 0.0f) {
          const float* v = in + f * D;
          for (int c = 0; c < D; c++) orow[i * D + c] += s * v[c];
        }
      }
      sc += dim;
    }
  }
  free(scales);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Kaldi pitch [KALDI-UPSTREAM pitch-functions.cc, resample.cc]; reference call site            */
/* pitch_kaldi.py:296-299 (compute_kaldi_pitch) and :535-537 (process_pitch)                    */
/* ------------------------------------------------------------------------------------------ */
static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

/* resample.cc LinearResample::FilterFunc / ArbitraryResample::FilterFunc (t is BaseFloat) */
static float filter_func(float t, float cutoff, int num_zeros) {
  float window, filter;
  if (fabs((double)t) < (double)num_zeros / (2.0 * (double)cutoff))
    window = (float)(0.5 * (1 + cos(M_2PI * (double)cutoff / (double)num_zeros * (double)t)));
  else
    window = 0.0f;
  if (t != 0.0f) filter = (float)(sin(M_2PI * (double)cutoff * (double)t) / (M_PI * (double)t));
  else filter = (float)(2.0 * (double)cutoff);
  return filter * window;
}

typedef struct {
  int rate_in, rate_out, in_unit, out_unit, num_zeros;
  float cutoff;
  int* first;   /* [out_unit] */
  int* nw;      /* [out_unit] */
  float** w;    /* [out_unit][nw] */
} linres_t;

static void linres_init(linres_t* r, int rate_in, int rate_out, float cutoff, int num_zeros) {
  r->rate_in = rate_in; r->rate_out = rate_out; r->cutoff = cutoff; r->num_zeros = num_zeros;
  int base = gcd_i(rate_in, rate_out);
  r->in_unit = rate_in / base; r->out_unit = rate_out / base;
  r->first = (int*)malloc(sizeof(int) * (size_t)r->out_unit);
  r->nw = (int*)malloc(sizeof(int) * (size_t)r->out_unit);
  r->w = (float**)malloc(sizeof(float*) * (size_t)r->out_unit);
  double window_width = (double)num_zeros / (2.0 * (double)cutoff);
  for (int i = 0; i < r->out_unit; i++) {
    double output_t = (double)i / (double)rate_out;
    double min_t = output_t - window_width, max_t = output_t + window_width;
    int min_idx = (int)ceil(min_t * rate_in), max_idx = (int)floor(max_t * rate_in);
    int num = max_idx - min_idx + 1;
    r->first[i] = min_idx; r->nw[i] = num;
    r->w[i] = (float*)malloc(sizeof(float) * (size_t)num);
    for (int j = 0; j < num; j++) {
      int input_index = min_idx + j;
      double input_t = (double)input_index / (double)rate_in, delta_t = input_t - output_t;
      r->w[i][j] = filter_func((float)delta_t, cutoff, num_zeros) / (float)rate_in;
    }
  }
}
static void linres_free(linres_t* r) {
  for (int i = 0; i < r->out_unit; i++) free(r->w[i]);
  free(r->w); free(r->first); free(r->nw);
}
static int64_t linres_num_out(const linres_t* r, int64_t n_in, int flush) {
  int64_t tick_freq = (int64_t)r->rate_in / gcd_i(r->rate_in, r->rate_out) * r->rate_out; /* lcm */
  int64_t ticks_per_in = tick_freq / r->rate_in;
  int64_t interval = n_in * ticks_per_in;
  if (!flush) {
    /* resample.cc: `BaseFloat window_width = num_zeros_ / (2.0 * filter_cutoff_); int32 window_width_ticks =
       floor(window_width * tick_freq);` - a FLOAT product (float x int32), rounded to float before the floor:
       800 Hz at 16 kHz is 9.99999978 in exact arithmetic and 10.0f as a float product (round 4: this line used a
       double product, one tick short for cutoffs whose width is not a dyadic fraction; the default 1 000 Hz is) */
    float window_width = (float)((double)r->num_zeros / (2.0 * (double)r->cutoff));
    int window_ticks = (int)floorf(window_width * (float)tick_freq);
    interval -= window_ticks;
  }
  if (interval <= 0) return 0;
  int64_t ticks_per_out = tick_freq / r->rate_out;
  int64_t last = interval / ticks_per_out;
  if (last * ticks_per_out == interval) last--;
  return last + 1;
}
/* whole-signal resample: out[k] for k < n_out, input beyond [0,n) treated as zero */
static void linres_apply(const linres_t* r, const float* in, int64_t n, float* out, int64_t n_out) {
  for (int64_t k = 0; k < n_out; k++) {
    int64_t unit = k / r->out_unit;
    int wrapped = (int)(k - unit * r->out_unit);
    int64_t first_in = r->first[wrapped] + unit * r->in_unit;
    const float* w = r->w[wrapped];
    int nw = r->nw[wrapped];
    float s = 0.0f;  /* chain_dot over the taps that fall inside the signal */
    for (int i = 0; i < nw; i++) {
      int64_t idx = first_in + i;
      if (idx >= 0 && idx < n) s = fmaf(w[i], in[idx], s);
    }
    out[k] = s;
  }
}

typedef struct {
  snf_pitch_options o;
  int first_lag, last_lag, num_lags, num_states;
  int win_size, win_shift, full_len;
  float* lags;
  /* ArbitraryResample */
  int* ar_first; int* ar_n; float** ar_w;
} pitchcfg_t;

static int pitch_window_size(const snf_pitch_options* o) {
  return (int)((double)o->resample_freq * (double)o->frame_length_ms / 1000.0);
}
static int pitch_window_shift(const snf_pitch_options* o) {
  return (int)((double)o->resample_freq * (double)o->frame_shift_ms / 1000.0);
}

static int pitchcfg_init(pitchcfg_t* c, const snf_pitch_options* o) {
  memset(c, 0, sizeof(*c));
  c->o = *o;
  if (!(o->samp_freq > 0 && o->resample_freq > 0 && o->lowpass_cutoff > 0 &&
        o->lowpass_cutoff * 2 <= o->samp_freq && o->lowpass_cutoff * 2 <= o->resample_freq &&
        o->lowpass_filter_width > 0 && o->upsample_filter_width > 0 && o->min_f0 > 0 &&
        o->max_f0 > o->min_f0 && o->delta_pitch > 0))
    return orc_fail("bad pitch extraction options");
  double outer_min_lag = 1.0 / (double)o->max_f0 - ((double)o->upsample_filter_width / (2.0 * (double)o->resample_freq));
  double outer_max_lag = 1.0 / (double)o->min_f0 + ((double)o->upsample_filter_width / (2.0 * (double)o->resample_freq));
  c->first_lag = (int)ceil((double)o->resample_freq * outer_min_lag);
  c->last_lag = (int)floor((double)o->resample_freq * outer_max_lag);
  c->num_lags = c->last_lag + 1 - c->first_lag;
  c->win_size = pitch_window_size(o);
  c->win_shift = pitch_window_shift(o);
  c->full_len = c->win_size + c->last_lag;
  if (c->win_shift <= 0 || c->win_size <= 0 || c->num_lags <= 0) return orc_fail("bad pitch frame options");
  /* SelectLags */
  float min_lag = (float)(1.0 / (double)o->max_f0), max_lag = (float)(1.0 / (double)o->min_f0);
  int cnt = 0;
  for (float lag = min_lag; lag <= max_lag; lag = (float)((double)lag * (1.0 + (double)o->delta_pitch))) cnt++;
  c->num_states = cnt;
  c->lags = (float*)malloc(sizeof(float) * (size_t)cnt);
  cnt = 0;
  for (float lag = min_lag; lag <= max_lag; lag = (float)((double)lag * (1.0 + (double)o->delta_pitch))) c->lags[cnt++] = lag;
  /* ArbitraryResample(num_lags, resample_freq, resample_freq*0.5, lags - first_lag/resample_freq, upsample_filter_width) */
  float upsample_cutoff = (float)((double)o->resample_freq * 0.5);
  c->ar_first = (int*)malloc(sizeof(int) * (size_t)cnt);
  c->ar_n = (int*)malloc(sizeof(int) * (size_t)cnt);
  c->ar_w = (float**)malloc(sizeof(float*) * (size_t)cnt);
  float filter_width = (float)((double)o->upsample_filter_width / (2.0 * (double)upsample_cutoff));
  float offset = (float)(-(double)c->first_lag / (double)o->resample_freq);
  for (int i = 0; i < cnt; i++) {
    float t = c->lags[i] + offset, t_min = t - filter_width, t_max = t + filter_width;
    int imin = (int)ceil((double)(o->resample_freq * t_min)), imax = (int)floor((double)(o->resample_freq * t_max));
    if (imin < 0) imin = 0;
    if (imax >= c->num_lags) imax = c->num_lags - 1;
    c->ar_first[i] = imin;
    c->ar_n[i] = imax - imin + 1;
    c->ar_w[i] = (float*)malloc(sizeof(float) * (size_t)(c->ar_n[i] > 0 ? c->ar_n[i] : 1));
    for (int j = 0; j < c->ar_n[i]; j++) {
      float delta_t = t - (float)(imin + j) / o->resample_freq;
      c->ar_w[i][j] = filter_func(delta_t, upsample_cutoff, o->upsample_filter_width) / o->resample_freq;
    }
  }
  return 0;
}
static void pitchcfg_free(pitchcfg_t* c) {
  for (int i = 0; i < c->num_states; i++) free(c->ar_w[i]);
  free(c->ar_w); free(c->ar_first); free(c->ar_n); free(c->lags);
}

/* OnlinePitchFeatureImpl::NumFramesAvailable */
static int64_t pitch_frames_available(const pitchcfg_t* c, int64_t n_down, int input_finished) {
  int64_t shift = c->win_shift, len = c->win_size;
  if (!input_finished) len += c->last_lag;
  if (n_down < len) return 0;
  if (!c->o.snip_edges) {
    if (input_finished) return (int64_t)((float)n_down * 1.0f / (float)shift + 0.5f);
    return (int64_t)((float)(n_down - len / 2) * 1.0f / (float)shift + 0.5f);
  }
  return (n_down - len) / shift + 1;
}

ORC_API int64_t orc_pitch_num_frames(const snf_pitch_options* o, int64_t n) {
  pitchcfg_t c;
  if (pitchcfg_init(&c, o)) return -2;
  linres_t lr;
  linres_init(&lr, (int)o->samp_freq, (int)o->resample_freq, o->lowpass_cutoff, o->lowpass_filter_width);
  int64_t n2 = linres_num_out(&lr, n, 1);
  int64_t T = pitch_frames_available(&c, n2, 1);
  linres_free(&lr); pitchcfg_free(&c);
  return T;
}

/* PitchFrameInfo::ComputeBacktraces — Kaldi's bounded two-sweep exact argmin search */
static void compute_backtraces(const pitchcfg_t* c, const float* nccf_pitch, const float* prev_fwd,
                               int* bp, float* this_fwd, int* lo, int* hi) {
  int S = c->num_states;
  const float delta_pitch_sq = (float)pow((double)logf((float)(1.0 + (double)c->o.delta_pitch)), 2.0);
  const float iff = delta_pitch_sq * c->o.penalty_factor;
  int last_bp = 0;
  for (int i = 0; i < S; i++) {
    int start_j = last_bp;
    float best_cost = (float)((start_j - i) * (start_j - i)) * iff + prev_fwd[start_j];
    int best_j = start_j;
    for (int j = start_j + 1; j < S; j++) {
      float this_cost = (float)((j - i) * (j - i)) * iff + prev_fwd[j];
      if (this_cost < best_cost) { best_cost = this_cost; best_j = j; }
      else break;
    }
    bp[i] = best_j; this_fwd[i] = best_cost;
    lo[i] = best_j; hi[i] = S - 1;
    last_bp = best_j;
  }
  for (int iter = 0; iter < S; iter++) {
    int changed = 0;
    if (iter % 2 == 0) {
      last_bp = S - 1;
      for (int i = S - 1; i >= 0; i--) {
        int lower = lo[i], upper = last_bp < hi[i] ? last_bp : hi[i];
        if (upper == lower) { last_bp = lower; continue; }
        float best_cost = this_fwd[i];
        int best_j = bp[i], initial = best_j;
        if (best_j == upper) { last_bp = best_j; continue; }
        for (int j = upper; j > lower + 1; j--) {
          float this_cost = (float)((j - i) * (j - i)) * iff + prev_fwd[j];
          if (this_cost < best_cost) { best_cost = this_cost; best_j = j; }
          else if (best_j > j) break;
        }
        hi[i] = best_j;
        if (best_j != initial) { this_fwd[i] = best_cost; bp[i] = best_j; changed = 1; }
        last_bp = best_j;
      }
    } else {
      last_bp = 0;
      for (int i = 0; i < S; i++) {
        int lower = last_bp > lo[i] ? last_bp : lo[i], upper = hi[i];
        if (upper == lower) { last_bp = lower; continue; }
        float best_cost = this_fwd[i];
        int best_j = bp[i], initial = best_j;
        if (best_j == lower) { last_bp = best_j; continue; }
        for (int j = lower; j < upper - 1; j++) {
          float this_cost = (float)((j - i) * (j - i)) * iff + prev_fwd[j];
          if (this_cost < best_cost) { best_cost = this_cost; best_j = j; }
          else if (best_j < j) break;
        }
        lo[i] = best_j;
        if (best_j != initial) { this_fwd[i] = best_cost; bp[i] = best_j; changed = 1; }
        last_bp = best_j;
      }
    }
    if (!changed) break;
  }
  /* ComputeLocalCost: 1 - nccf + soft_min_f0 * lag * nccf, added to the forward cost */
  for (int i = 0; i < S; i++) {
    float local = 1.0f;
    local += -1.0f * nccf_pitch[i];
    local += c->o.soft_min_f0 * c->lags[i] * nccf_pitch[i];
    this_fwd[i] += local;
  }
}

/* ComputeKaldiPitch, offline single-chunk call (frames_per_chunk = 0): AcceptWaveform(whole wave)
 * without flush, then InputFinished() flushes the resampler and processes the last frames.
 * The optional debug outputs receive the intermediates the GPU path can be compared with stage by
 * stage: the resampled signal [n_down], the lag-resampled NCCF [T, S], the NCCF without ballast at
 * the integer lags [T, L] and the Viterbi states [T]. */
static int pitch_impl(const snf_pitch_options* o, const int16_t* wave16, int64_t n, float* out,
                      float* dbg_down, float* dbg_res, float* dbg_pov, int32_t* dbg_states) {
  pitchcfg_t c;
  int rc = pitchcfg_init(&c, o);
  if (rc) return rc;
  if (o->preemph_coeff != 0.0f) { pitchcfg_free(&c); return orc_fail("pitch preemph_coeff unsupported"); }
  linres_t lr;
  linres_init(&lr, (int)o->samp_freq, (int)o->resample_freq, o->lowpass_cutoff, o->lowpass_filter_width);
  int64_t n_down_p1 = linres_num_out(&lr, n, 0), n_down = linres_num_out(&lr, n, 1);
  int64_t T1 = pitch_frames_available(&c, n_down_p1, 0);
  int64_t T = pitch_frames_available(&c, n_down, 1);
  if (T1 > T) T1 = T;
  if (T <= 0) { linres_free(&lr); pitchcfg_free(&c); return 0; }
  float* wave = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; i++) wave[i] = (float)wave16[i];
  float* down = (float*)malloc(sizeof(float) * (size_t)n_down);
  linres_apply(&lr, wave, n, down, n_down);
  if (dbg_down) memcpy(dbg_down, down, sizeof(float) * (size_t)n_down);
  /* signal statistics: phase 1 sees down[0:n_down_p1], phase 2 everything (double accumulators
     += float VecVec / Sum of each chunk) */
  double sumsq1 = (double)vecvec_long(down, down, n_down_p1), sum1 = (double)vecsum_long(down, n_down_p1);
  double sumsq2 = sumsq1 + (double)vecvec_long(down + n_down_p1, down + n_down_p1, n_down - n_down_p1);
  double sum2 = sum1 + (double)vecsum_long(down + n_down_p1, n_down - n_down_p1);
  /* (x * x instead of pow(x, 2.0): both are the correctly rounded square) */
  double m1 = n_down_p1 > 0 ? sum1 / (double)n_down_p1 : 0.0, m2 = sum2 / (double)n_down;
  double ms1 = n_down_p1 > 0 ? sumsq1 / (double)n_down_p1 - m1 * m1 : 0.0;
  double ms2 = sumsq2 / (double)n_down - m2 * m2;

  int S = c.num_states, L = c.num_lags, W = c.win_size;
  float* window = (float*)calloc((size_t)c.full_len + 16, sizeof(float));
  float* inner = (float*)malloc(sizeof(float) * (size_t)L);
  float* norm = (float*)malloc(sizeof(float) * (size_t)L);
  float* nccf_pitch = (float*)malloc(sizeof(float) * (size_t)L);
  float* nccf_pov = (float*)malloc(sizeof(float) * (size_t)L);
  float* pitch_res = (float*)malloc(sizeof(float) * (size_t)T * S);
  float* pov_res = (float*)malloc(sizeof(float) * (size_t)T * S);
  float* anp_frame = (float*)malloc(sizeof(float) * (size_t)T);
  int* bp = (int*)malloc(sizeof(int) * (size_t)T * S);
  int* lo = (int*)malloc(sizeof(int) * (size_t)S);
  int* hi = (int*)malloc(sizeof(int) * (size_t)S);
  float* fwd = (float*)calloc((size_t)S, sizeof(float));
  float* nfwd = (float*)calloc((size_t)S, sizeof(float));

  for (int64_t t = 0; t < T; t++) {
    double mean_square = t < T1 ? ms1 : ms2;
    int64_t start;
    if (o->snip_edges) start = t * c.win_shift;
    else start = (int64_t)(((double)t + 0.5) * c.win_shift) - c.full_len / 2;
    for (int i = 0; i < c.full_len; i++) {
      int64_t k = start + i;
      window[i] = (k >= 0 && k < n_down) ? down[k] : 0.0f;
    }
    /* ComputeCorrelation: mean of the first W samples removed from the whole window */
    float m = -strided16_sum(window, W) / (float)W;
    for (int i = 0; i < c.full_len; i++) window[i] += m;
    float e1 = strided16_sumsq(window, W);
    for (int lag = c.first_lag; lag <= c.last_lag; lag++) {
      inner[lag - c.first_lag] = chain_dot(window, window + lag, W);
      norm[lag - c.first_lag] = e1 * chain_dot(window + lag, window + lag, W);
    }
    /* ComputeNccf: sqrtf == (float)pow((double)x, 0.5) up to the rounding of pow */
    float bal = (float)((mean_square * W) * (mean_square * W) * (double)o->nccf_ballast);
    for (int l = 0; l < L; l++) {
      float den = sqrtf(norm[l] + bal);
      nccf_pitch[l] = den != 0.0f ? inner[l] / den : 0.0f;
      float den2 = sqrtf(norm[l]);
      nccf_pov[l] = den2 != 0.0f ? inner[l] / den2 : 0.0f;
    }
    /* avg_norm_prod: lane l of the 16 owns the lags 5 (l + 16 g) .. + 4, g = 0, 1, ... */
    {
      float p[16];
      for (int l16 = 0; l16 < 16; l16++) {
        float sacc = 0.0f;
        for (int g = 0; 5 * (l16 + 16 * g) < L; g++)
          for (int d = 0; d < 5 && 5 * (l16 + 16 * g) + d < L; d++) sacc += norm[5 * (l16 + 16 * g) + d];
        p[l16] = sacc;
      }
      anp_frame[t] = tree16(p) / (float)L;
    }
    if (dbg_pov) memcpy(dbg_pov + t * L, nccf_pov, sizeof(float) * (size_t)L);
    /* ArbitraryResample::Resample (AddMatVec per output column) */
    for (int s = 0; s < S; s++) {
      pitch_res[t * S + s] = chain_dot(nccf_pitch + c.ar_first[s], c.ar_w[s], c.ar_n[s]);
      pov_res[t * S + s] = chain_dot(nccf_pov + c.ar_first[s], c.ar_w[s], c.ar_n[s]);
    }
  }
  if (dbg_res) memcpy(dbg_res, pitch_res, sizeof(float) * (size_t)T * S);
  /* Viterbi forward */
  for (int64_t t = 0; t < T; t++) {
    compute_backtraces(&c, pitch_res + t * S, fwd, bp + t * S, nfwd, lo, hi);
    float* sw = fwd; fwd = nfwd; nfwd = sw;
    float mn = fwd[0];
    for (int s = 1; s < S; s++) if (fwd[s] < mn) mn = fwd[s];
    for (int s = 0; s < S; s++) fwd[s] += -mn;
  }
  /* [KALDI-UPSTREAM] pitch-functions.cc runs RecomputeBacktraces (a) inside AcceptWaveform when frame
   * recompute_frame - 1 has just been processed, over the frames 0 .. recompute_frame - 1, and (b) in
   * InputFinished() when the utterance has fewer frames than that, over all of them.  In the offline
   * call the first AcceptWaveform covers the T1 frames available before the resampler is flushed (all
   * with the statistics ms1) and the flush adds the last T - T1 (2 or 3) frames with ms2:
   *   T1 >= recompute_frame: (a) compares ms1 with ms1 -> nothing to do;
   *   T  <  recompute_frame: (b) with the final ms2;
   *   T1 <  recompute_frame <= T (utterances of 500 - 502 pitch frames): (a) fires in the SECOND call,
   *     with ms2, over frames 0 .. 499 of which T1 carry ms1; the forward costs restart from zero and
   *     frames 500 .. T - 1 follow with their own (ms2) NCCF.  Frames >= T1 rescale by exactly 1, so this
   *     is the same computation as (b) - round 3; it used to be left out on both sides. */
  if (T < o->recompute_frame || T1 < o->recompute_frame) {
    float mean_square = (float)ms2;
    int must = 0;
    for (int64_t t = 0; t < T; t++) {
      /* ApproxEqual(a, b, 0.01): |a-b| <= 0.01 * (|a|+|b|) */
      float a = (float)(t < T1 ? ms1 : ms2), b = mean_square;
      if (!(fabsf(a - b) <= 0.01f * (fabsf(a) + fabsf(b)))) must = 1;
    }
    if (must) {
      float new_ballast = (float)(((double)mean_square * W) * ((double)mean_square * W) * (double)o->nccf_ballast);
      for (int s = 0; s < S; s++) fwd[s] = 0.0f;
      for (int64_t t = 0; t < T; t++) {
        float old_ms = (float)(t < T1 ? ms1 : ms2), anp = anp_frame[t];
        float old_ballast = (float)(((double)old_ms * W) * ((double)old_ms * W) * (double)o->nccf_ballast);
        /* sqrtf == powf(x, 0.5f) up to the rounding of powf */
        float scale = sqrtf((old_ballast + anp) / (new_ballast + anp));
        for (int s = 0; s < S; s++) pitch_res[t * S + s] *= scale;
        compute_backtraces(&c, pitch_res + t * S, fwd, bp + t * S, nfwd, lo, hi);
        float* sw = fwd; fwd = nfwd; nfwd = sw;
        float mn = fwd[0];
        for (int s = 1; s < S; s++) if (fwd[s] < mn) mn = fwd[s];
        for (int s = 0; s < S; s++) fwd[s] += -mn;
      }
    }
  }
  /* traceback */
  int best = 0;
  for (int s = 1; s < S; s++) if (fwd[s] < fwd[best]) best = s;
  for (int64_t t = T - 1; t >= 0; t--) {
    out[t * 2 + 0] = pov_res[t * S + best];
    out[t * 2 + 1] = 1.0f / c.lags[best];
    if (dbg_states) dbg_states[t] = best;
    best = bp[t * S + best];
  }
  free(wave); free(down); free(window); free(inner); free(norm); free(nccf_pitch); free(nccf_pov);
  free(pitch_res); free(pov_res); free(anp_frame); free(bp); free(lo); free(hi);
  free(fwd); free(nfwd);
  linres_free(&lr); pitchcfg_free(&c);
  return 0;
}
ORC_API int orc_pitch(const snf_pitch_options* o, const int16_t* wave16, int64_t n, float* out) {
  return pitch_impl(o, wave16, n, out, NULL, NULL, NULL, NULL);
}
ORC_API int orc_pitch_debug(const snf_pitch_options* o, const int16_t* wave16, int64_t n, float* out,
                            float* down, float* res, float* pov, int32_t* states) {
  return pitch_impl(o, wave16, n, out, down, res, pov, states);
}

/* exported for tests: lags table, resampled signal */
ORC_API int orc_pitch_lags(const snf_pitch_options* o, float* lags, int32_t* first_lag, int32_t* last_lag) {
  pitchcfg_t c;
  int rc = pitchcfg_init(&c, o);
  if (rc) return rc;
  int S = c.num_states;
  if (lags) memcpy(lags, c.lags, sizeof(float) * (size_t)S);
  if (first_lag) *first_lag = c.first_lag;
  if (last_lag) *last_lag = c.last_lag;
  pitchcfg_free(&c);
  return S;
}
ORC_API int64_t orc_linear_resample(int rate_in, int rate_out, float cutoff, int num_zeros,
                                    const float* in, int64_t n, float* out, int flush) {
  linres_t lr;
  linres_init(&lr, rate_in, rate_out, cutoff, num_zeros);
  int64_t n_out = linres_num_out(&lr, n, flush);
  if (out) linres_apply(&lr, in, n, out, n_out);
  linres_free(&lr);
  return n_out;
}

/* ProcessPitch [KALDI-UPSTREAM pitch-functions.cc OnlineProcessPitch]; noise term must be 0 */
static float nccf_to_pov_feature(float n) {
  if (n > 1.0f) n = 1.0f; else if (n < -1.0f) n = -1.0f;
  return (float)(pow(1.0001 - (double)n, 0.15) - 1.0);
}
static float nccf_to_pov(float n) {
  float ndash = fabsf(n);
  if (ndash > 1.0f) ndash = 1.0f;
  /* Exp() is called with double arguments here, so Kaldi takes the double overload */
  float r = (float)(-5.2 + 5.4 * exp(7.5 * ((double)ndash - 1.0)) + 4.8 * (double)ndash -
                    2.0 * exp(-10.0 * (double)ndash) + 4.2 * exp(20.0 * ((double)ndash - 1.0)));
  return (float)(1.0 / (1.0 + exp(-1.0 * (double)r)));
}

ORC_API int32_t orc_process_pitch_ndims(const snf_pitch_post_options* o) {
  return (o->add_pov_feature ? 1 : 0) + (o->add_normalized_log_pitch ? 1 : 0) +
         (o->add_delta_pitch ? 1 : 0) + (o->add_raw_log_pitch ? 1 : 0);
}

ORC_API int orc_process_pitch(const snf_pitch_post_options* o, const float* in, int64_t T, float* out) {
  int D = orc_process_pitch_ndims(o);
  if (D <= 0) return orc_fail("At least one of the pitch features should be chosen");
  if (o->delta_pitch_noise_stddev != 0.0f)
    return orc_fail("oracle requires delta_pitch_noise_stddev == 0 (RandGauss not reproducible)");
  if (o->delay != 0) return orc_fail("delay != 0 changes the number of output rows; unsupported");
  for (int64_t frame = 0; frame < T; frame++) {
    int64_t t = frame < o->delay ? 0 : frame - o->delay;
    int idx = 0;
    float* row = out + frame * D;
    if (o->add_pov_feature) row[idx++] = o->pov_scale * nccf_to_pov_feature(in[t * 2]) + o->pov_offset;
    if (o->add_normalized_log_pitch) {
      int64_t b = t - o->normalization_left_context; if (b < 0) b = 0;
      int64_t e = t + o->normalization_right_context + 1; if (e > T) e = T;
      double sum_pov = 0.0, sum_lp = 0.0;
      for (int64_t f = b; f < e; f++) {
        float pov = nccf_to_pov(in[f * 2]), lp = logf(in[f * 2 + 1]);
        sum_pov += (double)pov;
        sum_lp += (double)(pov * lp);
      }
      float log_pitch = logf(in[t * 2 + 1]);
      float avg = (float)(sum_lp / sum_pov);
      row[idx++] = (log_pitch - avg) * o->pitch_scale;
    }
    if (o->add_delta_pitch) {
      int ctx = o->delta_window;
      int64_t s = t - ctx; if (s < 0) s = 0;
      int64_t e = t + ctx + 1; if (e > T) e = T;
      int nw = (int)(e - s);
      float feats[2048], dl[4096];
      if (nw > 2048) return orc_fail("delta_window too large");
      for (int64_t f = s; f < e; f++) feats[f - s] = logf(in[f * 2 + 1]);
      int rc = orc_deltas(1, ctx, feats, nw, 1, dl);
      if (rc) return rc;
      row[idx++] = (dl[(t - s) * 2 + 1] + 0.0f) * o->delta_pitch_scale;
    }
    if (o->add_raw_log_pitch) row[idx++] = logf(in[t * 2 + 1]);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* VAD and CMVN                                                          */
/* ------------------------------------------------------------------------------------------ */
/* [KALDI-UPSTREAM ivector/voice-activity-detection.cc ComputeVadEnergy]; reference call site
 * postprocessor/vad.py:182-185.  `feats` is [T, D] float32, column 0 is the log-energy. */
ORC_API int orc_vad_energy(float energy_threshold, float energy_mean_scale, int frames_context,
                           float proportion_threshold, const float* feats, int64_t T, int D,
                           float* out) {
  if (T == 0) return 0;
  float thr = energy_threshold;
  if (energy_mean_scale != 0.0f) {
    double dsum = 0.0; /* VectorBase<float>::Sum() accumulates in double */
    for (int64_t t = 0; t < T; t++) dsum += feats[t * D];
    float sum = (float)dsum;
    thr += energy_mean_scale * sum / (float)T;
  }
  for (int64_t t = 0; t < T; t++) {
    int num = 0, den = 0;
    for (int64_t t2 = t - frames_context; t2 <= t + frames_context; t2++) {
      if (t2 >= 0 && t2 < T) {
        den++;
        if (feats[t2 * D] > thr) num++;
      }
    }
    out[t] = ((float)num >= (float)den * proportion_threshold) ? 1.0f : 0.0f;
  }
  return 0;
}

/* [KALDI-UPSTREAM transform/cmvn.cc AccCmvnStats]; reference call site postprocessor/cmvn.py:216-219.
 * stats is [2, D+1] double, accumulated into. */
ORC_API int orc_cmvn_accumulate(const float* feats, int64_t T, int D, const float* weights,
                                double* stats) {
  double* mean = stats;
  double* var = stats + (D + 1);
  for (int64_t t = 0; t < T; t++) {
    float w = weights ? weights[t] : 1.0f;
    if (w == 0.0f) continue;
    mean[D] += (double)w;
    for (int d = 0; d < D; d++) {
      float x = feats[t * D + d];
      mean[d] += (double)(x * w);
      var[d] += (double)(x * x * w);
    }
  }
  return 0;
}

/* [KALDI-UPSTREAM transform/cmvn.cc ApplyCmvn / ApplyCmvnReverse]; reference call site
 * postprocessor/cmvn.py:277-278.  In place on feats [T, D]. */
ORC_API int orc_cmvn_apply(const double* stats, int D, int norm_vars, int reverse, float* feats,
                           int64_t T) {
  double count = stats[D];
  if (count < 1.0) return orc_fail("Insufficient stats for cepstral mean and variance normalization");
  for (int d = 0; d < D; d++) {
    double mean = stats[d] / count, scale = 1.0, offset;
    if (!reverse) {
      if (!norm_vars) {
        /* offset.AddVec(-1.0 / count, mean_stats): AddVec's alpha is a BaseFloat */
        float alpha = (float)(-1.0 / count);
        float off = (float)((double)alpha * stats[d]);
        for (int64_t t = 0; t < T; t++) feats[t * D + d] += off;
        continue;
      }
      double var = stats[(D + 1) + d] / count - mean * mean;
      if (var < 1.0e-20) var = 1.0e-20;
      scale = 1.0 / sqrt(var);
      offset = -(mean * scale);
    } else {
      offset = mean;
      if (norm_vars) {
        double var = stats[(D + 1) + d] / count - mean * mean;
        if (var < 1.0e-20) var = 1.0e-20;
        scale = sqrt(var);
      }
    }
    float fs = (float)scale, fo = (float)offset;
    for (int64_t t = 0; t < T; t++) {
      float x = feats[t * D + d];
      if (norm_vars) x = x * fs;      /* MulColsVec */
      feats[t * D + d] = x + fo;      /* AddVecToRows */
    }
  }
  return 0;
}

/* [KALDI-UPSTREAM feat/feature-functions.cc SlidingWindowCmnInternal] (double arithmetic, incremental
 * window sums); reference call site postprocessor/cmvn.py:493-495. */
ORC_API int orc_sliding_cmn(int center, int cmn_window, int min_window, int normalize_variance,
                            const float* in, int64_t T, int D, float* out) {
  if (cmn_window <= 0 || min_window <= 0) return orc_fail("bad sliding window options");
  double* cur_sum = (double*)calloc((size_t)D, sizeof(double));
  double* cur_sumsq = (double*)calloc((size_t)D, sizeof(double));
  int64_t last_start = -1, last_end = -1;
  for (int64_t t = 0; t < T; t++) {
    int64_t ws, we;
    if (center) { ws = t - (cmn_window / 2); we = ws + cmn_window; }
    else { ws = t - cmn_window; we = t + 1; }
    if (ws < 0) { we -= ws; ws = 0; }
    if (!center) {
      if (we > t) we = (t + 1 > min_window) ? t + 1 : min_window;
    }
    if (we > T) { ws -= (we - T); we = T; if (ws < 0) ws = 0; }
    if (last_start == -1) {
      for (int d = 0; d < D; d++) {
        double s = 0.0, q = 0.0;
        for (int64_t r = ws; r < we; r++) { double x = in[r * D + d]; s += x; q += x * x; }
        cur_sum[d] = s; cur_sumsq[d] = q;
      }
    } else {
      if (ws > last_start)
        for (int d = 0; d < D; d++) {
          double x = in[last_start * D + d];
          cur_sum[d] += -1.0 * x;
          if (normalize_variance) cur_sumsq[d] += -1.0 * x * x;
        }
      if (we > last_end)
        for (int d = 0; d < D; d++) {
          double x = in[last_end * D + d];
          cur_sum[d] += 1.0 * x;
          if (normalize_variance) cur_sumsq[d] += 1.0 * x * x;
        }
    }
    int64_t wf = we - ws;
    last_start = ws; last_end = we;
    for (int d = 0; d < D; d++) {
      double o = (double)in[t * D + d] + (-1.0 / (double)wf) * cur_sum[d];
      if (normalize_variance) {
        if (wf == 1) o = 0.0;
        else {
          double v = cur_sumsq[d] * (1.0 / (double)wf);
          v += (-1.0 / ((double)wf * (double)wf)) * cur_sum[d] * cur_sum[d];
          if (v < 1.0e-10) v = 1.0e-10;
          o *= pow(v, -0.5);
        }
      }
      out[t * D + d] = (float)o;
    }
  }
  free(cur_sum); free(cur_sumsq);
  return 0;
}
