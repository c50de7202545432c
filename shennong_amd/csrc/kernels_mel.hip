// Generic fused speech-feature kernel for gfx950 (MI355X).
//
// One wavefront (64 lanes) owns one frame at a time; its whole pipeline
//   ExtractWindow (+reflection) -> dither -> DC removal -> raw log-energy -> pre-emphasis -> window
//   -> zero-pad -> real FFT (packed complex radix-2 in LDS) -> power spectrum
//   -> {log spectrogram | mel filterbank (+log) | MFCC DCT+lifter | raw mel for PLP}
// runs out of a wave-private LDS region, so a frame makes exactly one trip from HBM (int16 samples
// in, float32 features out) and no workgroup barrier is ever needed.  This kernel handles every
// option combination (any window length / FFT size / window type / snip_edges / VTLN warp);
// kernels_fbank512.hip specialises the headline 25 ms / 16 kHz (512-point) configuration.
//
// Restates [KALDI-UPSTREAM] feature-window.cc (ExtractWindow / ProcessWindow), feature-fbank.cc,
// feature-mfcc.cc, feature-spectrogram.cc, mel-computations.cc (MelBanks::Compute), which the
// reference reaches through pykaldi at shennong/processor/base.py:429-431 and
// spectrogram.py:138-140; the in-tree restatements of the per-frame recipe are plp.py:171-260.
#include <float.h>

#include "snf_internal.h"

namespace snf {

namespace {

constexpr int kWaves = 4;  // wavefronts per workgroup (independent of each other)

__device__ __forceinline__ void wave_lds_sync() {
  // make this wave's LDS writes visible to its own later reads (all 64 lanes run in lock-step;
  // only compiler reordering and outstanding lgkm counters have to be fenced)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// sum over the 64 lanes, same bits in every lane.  __shfl_xor lowers to ds_bpermute_b32 (an LDS round
// trip per step); DPP keeps the reduction on the VALU: quad_perm [1,0,3,2] / [2,3,0,1], row_ror:4 / :8
// inside the 16-lane rows, then the four row sums are read with v_readlane and added uniformly.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf,
                                                               0xf, false));
}
__device__ __forceinline__ float lane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x124>(v);
  v += dpp_f<0x128>(v);
  return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}

// largest u with offsets[u] <= g (offsets[n] > g): the utterance that owns global row g
__device__ __forceinline__ int64_t find_utt(const int64_t* __restrict__ offsets, int64_t n,
                                            int64_t g) {
  int64_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// counter-based N(0,1): statistical stand-in for Kaldi's RandGauss() dither (not reproducible
// against C rand(); parity tests run with dither = 0 exactly like the reference's own tests)
__device__ __forceinline__ float gauss(uint64_t seed, uint64_t frame, uint32_t i) {
  const uint64_t h = mix64(seed ^ mix64(frame * 0x100000001B3ull + i));
  const float u1 = (static_cast<float>((h >> 40) & 0xFFFFFF) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = static_cast<float>((h >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__device__ __forceinline__ int bit_reverse(int v, int bits) {
  return bits == 0 ? 0 : static_cast<int>(__brev(static_cast<unsigned>(v)) >> (32 - bits));
}

}  // namespace

__global__ __launch_bounds__(kWaves * 64, 8) void mel_features_generic_kernel(
    const MelParams p, const BatchArgs b, float* __restrict__ out, const int out_cols,
    double* __restrict__ energy_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = threadIdx.x >> 6;
  const int N = p.padded, M = p.half, L = p.win_len;
  // wave-private LDS: xs (samples, later power spectrum), zs[N] (complex FFT data), mel[nb]
  // xs holds the L samples of the frame, later the M + 1 power bins: the larger of the two (zero
  // padding to N happens on the way into zs), rounded to keep zs 8-byte aligned
  const int xs_floats = ((L > M + 1 ? L : M + 1) + 1) & ~1;
  const int mel_floats = (p.num_bins + 1) & ~1;
  const size_t per_wave = static_cast<size_t>(xs_floats + ((N + 1) & ~1) + mel_floats);
  float* xs = reinterpret_cast<float*>(smem) + per_wave * wid;
  float* zsf = xs + xs_floats;
  float2* zs = reinterpret_cast<float2*>(zsf);
  float* melbuf = zsf + ((N + 1) & ~1);
  float* ps = xs;

  const int64_t stride = static_cast<int64_t>(gridDim.x) * kWaves;
  for (int64_t g = static_cast<int64_t>(blockIdx.x) * kWaves + wid; g < b.total_frames;
       g += stride) {
    // ---- which utterance / which frame ---------------------------------------------------------
    // (frame -> utterance table when the host built one: the binary search is a chain of ~14
    // dependent loads per frame)
    const int64_t u = b.frame_utt ? b.frame_utt[g] : find_utt(b.frame_offsets, b.n_utts, g);
    if (b.utt_mask && !b.utt_mask[u]) continue;  // (a wave-uniform choice: one frame per wave)
    const int64_t f = g - b.frame_offsets[u];
    const int64_t s0 = b.sample_offsets[u];
    const int64_t n = b.sample_offsets[u + 1] - s0;
    const int warp_id = b.utt_warp ? b.utt_warp[u] : 0;
    const int64_t start = p.snip_edges
                              ? f * p.win_shift
                              : f * p.win_shift + p.win_shift / 2 - p.win_len / 2;
    const int16_t* __restrict__ w = b.wave + s0;

    const uint64_t noise_id = p.dither != 0.0f ? wave_noise_id(b, u, f) : 0;
    // ---- ExtractWindow: copy L samples, reflecting at the utterance edges ------------------------
    float part = 0.0f;
    for (int i = lane; i < L; i += 64) {
      int64_t k = start + i;
      while (k < 0 || k >= n) k = k < 0 ? -k - 1 : 2 * n - 1 - k;
      float v = static_cast<float>(w[k]);
      if (p.dither != 0.0f) v += p.dither * gauss(p.seed, noise_id, i);
      xs[i] = v;
      part += v;
    }
    // ---- ProcessWindow: DC removal, raw energy ---------------------------------------------------
    float neg_mean = 0.0f;
    if (p.remove_dc) neg_mean = -wave_sum(part) / static_cast<float>(L);
    float e_part = 0.0f;
    for (int i = lane; i < L; i += 64) {
      const float v = xs[i] + neg_mean;
      xs[i] = v;
      e_part += v * v;
    }
    float raw_energy = 0.0f;
    if (p.need_raw) raw_energy = wave_sum(e_part);
    wave_lds_sync();
    // ---- pre-emphasis (needs the left neighbour), window, zero-pad, bit-reversed packing ---------
    float e2_part = 0.0f;
    for (int i = lane; i < N; i += 64) {
      float y = 0.0f;
      if (i < L) {
        const float x = xs[i];
        const float xm = xs[i > 0 ? i - 1 : 0];
        y = (x - p.preemph * xm) * p.window[i];
      }
      e2_part += y * y;
      if (p.pow2) zsf[2 * bit_reverse(i >> 1, p.log2_half) + (i & 1)] = y;
      else zsf[i] = y;
    }
    float post_energy = 0.0f;
    if (p.need_post) post_energy = wave_sum(e2_part);
    if (p.kind == SNF_KIND_ENERGY) {
      // EnergyProcessor (reference processor/energy.py:173-183): float64 sum of squares of the
      // processed window, floored at the smallest double, then compressed
      wave_lds_sync();
      double de = 0.0;
      for (int i = lane; i < L; i += 64) {
        const double y = p.pow2 ? zsf[2 * bit_reverse(i >> 1, p.log2_half) + (i & 1)] : zsf[i];
        de += y * y;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) de += __shfl_xor(de, off, 64);
      de = fmax(de, DBL_MIN);
      double v = de;
      if (p.compression == SNF_COMPRESS_LOG) v = log(de);
      else if (p.compression == SNF_COMPRESS_SQRT) v = sqrt(de);
      if (lane == 0) out[g * static_cast<int64_t>(out_cols)] = static_cast<float>(v);
      wave_lds_sync();
      continue;
    }

    if (p.pow2) {
      // ---- complex FFT of size M = N/2 on the packed frame (DIT on bit-reversed data, in LDS) -------
      // Two radix-2 stages per pass over the data (radix-4 butterflies, half the LDS traffic); one
      // plain radix-2 stage first when log2(M) is odd.  Same butterflies in the same order as the
      // stage-by-stage form: the results are bit-identical to it.
      int s = 0;
      if (p.log2_half & 1) {
        wave_lds_sync();
        for (int j = lane; j < (M >> 1); j += 64) {  // stage 0: half = 1, twiddle 1
          const float2 a = zs[2 * j], c = zs[2 * j + 1];
          zs[2 * j] = make_float2(a.x + c.x, a.y + c.y);
          zs[2 * j + 1] = make_float2(a.x - c.x, a.y - c.y);
        }
        s = 1;
      }
      for (; s + 2 <= p.log2_half; s += 2) {
        wave_lds_sync();
        const int half = 1 << s;
        const int stride_a = M >> (s + 1), stride_b = M >> (s + 2);
        for (int j = lane; j < (M >> 2); j += 64) {
          const int k = j & (half - 1);
          const int i0 = ((j >> s) << (s + 2)) + k;
          const float2 x0 = zs[i0], x1 = zs[i0 + half], x2 = zs[i0 + 2 * half], x3 = zs[i0 + 3 * half];
          const float2 ta = p.tw_fft[k * stride_a];  // W_{2 half}^k
          const float2 tb = p.tw_fft[k * stride_b];  // W_{4 half}^k
          // stage s on (x0, x1) and (x2, x3)
          const float ar = x1.x * ta.x - x1.y * ta.y, ai = x1.x * ta.y + x1.y * ta.x;
          const float br = x3.x * ta.x - x3.y * ta.y, bi = x3.x * ta.y + x3.y * ta.x;
          const float2 y0 = make_float2(x0.x + ar, x0.y + ai), y1 = make_float2(x0.x - ar, x0.y - ai);
          const float2 y2 = make_float2(x2.x + br, x2.y + bi), y3 = make_float2(x2.x - br, x2.y - bi);
          // stage s + 1 on (y0, y2) with W^k and (y1, y3) with W^(k + half) = -i W^k... the table
          // holds that twiddle too: read it so that the products match the radix-2 form bit for bit
          const float2 tc = p.tw_fft[(k + half) * stride_b];
          const float cr = y2.x * tb.x - y2.y * tb.y, ci = y2.x * tb.y + y2.y * tb.x;
          const float dr = y3.x * tc.x - y3.y * tc.y, di = y3.x * tc.y + y3.y * tc.x;
          zs[i0] = make_float2(y0.x + cr, y0.y + ci);
          zs[i0 + 2 * half] = make_float2(y0.x - cr, y0.y - ci);
          zs[i0 + half] = make_float2(y1.x + dr, y1.y + di);
          zs[i0 + 3 * half] = make_float2(y1.x - dr, y1.y - di);
        }
      }
      wave_lds_sync();
      // ---- real-FFT unpack + ComputePowerSpectrum (Nyquist bin kept like Kaldi) -------------------
      if (lane == 0) {
        const float2 z0 = zs[0];
        const float dc = z0.x + z0.y, ny = z0.x - z0.y;
        ps[0] = dc * dc;
        ps[M] = ny * ny;
      }
      for (int k = 1 + lane; 2 * k <= M; k += 64) {
        const float2 zk = zs[k], zm = zs[M - k];
        const float2 t = p.tw_unpack[k];
        const float c_re = 0.5f * (zk.x + zm.x), c_im = 0.5f * (zk.y - zm.y);
        const float d_re = 0.5f * (zk.y + zm.y), d_im = -0.5f * (zk.x - zm.x);
        const float t_re = d_re * t.x - d_im * t.y, t_im = d_re * t.y + d_im * t.x;
        const float a_re = c_re + t_re, a_im = c_im + t_im;
        ps[k] = a_re * a_re + a_im * a_im;
        if (M - k != k) {
          const float b_re = c_re - t_re, b_im = t_im - c_im;
          ps[M - k] = b_re * b_re + b_im * b_im;
        }
      }
    } else {
      // ---- direct DFT for non power-of-two frames (round_to_power_of_two = False) -----------------
      wave_lds_sync();
      for (int k = lane; k <= M; k += 64) {
        float re = 0.0f, im = 0.0f;
        int idx = 0;
        for (int t = 0; t < N; ++t) {
          const float2 wv = p.tw_dft[idx];
          const float x = zsf[t];
          re += x * wv.x;
          im += x * wv.y;
          idx += k;
          if (idx >= N) idx -= N;
        }
        ps[k] = (k == 0 || k == M) ? re * re : re * re + im * im;
      }
    }
    wave_lds_sync();

    // ---- log energy column -----------------------------------------------------------------------
    const float e_lin = p.need_raw ? raw_energy : post_energy;
    float log_energy = 0.0f;
    if (p.kind == SNF_KIND_PLP) {
      // shennong's PLP floors with float64 eps and takes a double log (reference plp.py:191-193)
      if ((p.need_raw || p.need_post) && lane == 0)
        energy_out[g] = static_cast<double>(e_lin);  // (plp_tail_kernel takes the double log)
    } else if (p.need_raw || p.need_post) {
      log_energy = logf(fmaxf(e_lin, FLT_EPSILON));
      if (p.has_floor && log_energy < p.log_energy_floor) log_energy = p.log_energy_floor;
    }

    float* __restrict__ row = out + g * static_cast<int64_t>(out_cols);
    if (p.kind == SNF_KIND_SPECTROGRAM) {
      for (int k = lane; k <= M; k += 64) {
        float v = logf(fmaxf(ps[k], FLT_EPSILON));
        if (k == 0) v = log_energy;
        row[k] = v;
      }
    } else {
      const int nb = p.num_bins;
      const int* __restrict__ mfirst = p.mel_first + warp_id * nb;
      const int* __restrict__ msize = p.mel_size + warp_id * nb;
      const int* __restrict__ moff = p.mel_offset + warp_id * nb;
      if (p.kind == SNF_KIND_FBANK && !p.use_power) {
        for (int k = lane; k <= M; k += 64) ps[k] = sqrtf(ps[k]);
        wave_lds_sync();
      }
      const int mel_col = (p.kind == SNF_KIND_FBANK && p.use_energy && !p.htk_compat) ? 1 : 0;
      for (int m = lane; m < nb; m += 64) {
        const int first = mfirst[m], size = msize[m];
        const float* __restrict__ wt = p.mel_w + moff[m];
        float acc = 0.0f;
        for (int j = 0; j < size; ++j) acc += wt[j] * ps[first + j];
        if (p.kind == SNF_KIND_FBANK) {
          row[mel_col + m] = p.use_log ? logf(fmaxf(acc, FLT_EPSILON)) : acc;
        } else if (p.kind == SNF_KIND_MFCC) {
          melbuf[m] = logf(fmaxf(acc, FLT_EPSILON));
        } else {  // PLP: linear mel energies, the recipe continues in plp_tail_kernel
          row[m] = acc;
        }
      }
      if (p.kind == SNF_KIND_FBANK && p.use_energy && lane == 0)
        row[p.htk_compat ? nb : 0] = log_energy;
      if (p.kind == SNF_KIND_MFCC) {
        wave_lds_sync();
        for (int c = lane; c < p.num_ceps; c += 64) {
          const float* __restrict__ d = p.dct + c * nb;
          float v = 0.0f;
          for (int m = 0; m < nb; ++m) v += d[m] * melbuf[m];
          if (p.lifter) v *= p.lifter[c];
          if (c == 0 && p.use_energy) v = log_energy;
          int oc = c;
          if (p.htk_compat) {
            oc = c == 0 ? p.num_ceps - 1 : c - 1;
            if (c == 0 && !p.use_energy)
              v = static_cast<float>(static_cast<double>(v) * 1.4142135623730950488016887);
          }
          row[oc] = v;
        }
      }
    }
    wave_lds_sync();  // the next frame reuses xs/zs
  }
}

int launch_mel_features(const MelParams& p, const BatchArgs& b, float* out, int out_cols,
                        double* energy_out, hipStream_t stream) {
  if (b.total_frames <= 0) return SNF_OK;
  const int xs_need = p.win_len > p.half + 1 ? p.win_len : p.half + 1;
  const int xs_floats = (xs_need + 1) & ~1;
  const int mel_floats = (p.num_bins + 1) & ~1;
  const size_t lds = sizeof(float) * kWaves *
                     static_cast<size_t>(xs_floats + ((p.padded + 1) & ~1) + mel_floats);
  if (lds > 160 * 1024)
    return set_error(SNF_E_RUNTIME, "frame too long for the LDS-resident FFT (padded window > 4096)");
  if (lds > 64 * 1024)
    SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mel_features_generic_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      static_cast<int>(lds)));
  int64_t blocks = (b.total_frames + kWaves - 1) / kWaves;
  const int64_t max_blocks = 256 * 16;  // 256 CUs x resident workgroups of 4 waves x depth
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL(mel_features_generic_kernel, dim3(static_cast<unsigned>(blocks)),
                     dim3(kWaves * 64), lds, stream, p, b, out, out_cols, energy_out);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

}  // namespace snf
