// Register-resident spectrogram / filterbank / MFCC / PLP-mel kernel for frames that pad to 2048 samples
// (25 ms windows at 44.1 and 48 kHz: the reference tests MFCC at 44.1 kHz, test/processor/test_mfcc.py:
// 129-137) and, zero-extended, to 1024 samples (32 kHz), on gfx950.  Same per-frame recipe as
// kernels_mel.hip ([KALDI-UPSTREAM] feature-window.cc ProcessWindow order, feature-fbank.cc,
// feature-mfcc.cc, MelBanks::Compute; reached by the reference at shennong/processor/base.py:429-431),
// with the LDS radix-4 FFT of that kernel replaced by register passes:
//
//   wave64 = ONE frame; complex packing z[n] = x[2n] + i x[2n+1], n < 1024 = 16 x 16 x 4
//   A  lane L loads z[L + 64 j], j < 16 (256 contiguous bytes per wave instruction: SGPR base + lane offset
//      + immediate; the next frame's samples are requested when this frame's transform is done and land
//      during its mel phase); DC removal over the wave, pre-emphasis (left neighbour through 16
//      ds_bpermute issued together), window
//   B  16-point FFT over j in registers, twiddle W1024^(L k1), transpose through the wave's 8.5 KB LDS
//      buffer (row pitches 68 and 17 complex: every access is bank-conflict free AND a lane-constant base
//      plus an immediate offset - six address registers serve all ~140 LDS accesses of a frame)
//   C  lane (k1, bq): four 4-point DFTs over the 16-lane rows a, twiddle W64^(b c), transpose
//   D  lane (k1, c): 16-point FFT over b -> lane holds Z[k1 + 16 c + 64 d], d < 16
//   E  real-FFT unpack + power: bins k < 512 pair with 1024 - k, whose spectrum values come from the
//      partner lane through LDS; power spectrum (1025 bins) to LDS
//   F  epilogue: mel filterbank by teams of 8 lanes per bin, 8 bins per round: a lane owns 4 consecutive
//      taps of every 32-tap slice of the filter (16-byte weight loads from a per-plan table - one per warp
//      factor for VTLN - in which the filters start at multiples of 4 bins, are zero-padded to whole
//      slices and rotated by the team index: no per-tap test, conflict-free LDS reads); DCT-II by teams
//      of 4 lanes per coefficient; log / lifter / energy conventions as in the generic kernel
// The index maps were checked lane by lane against numpy.fft, and every LDS access against the bank model
// of MI355X_MICROARCH.md, before the first GPU run (tools/model_fbank2048.py, tests/test_fbank2048_model.py).
// Frames that pad to 1024 samples run as the 2048-point transform of the zero-extended frame:
// X2048[2 k] = X1024[k]; their spectrum (the even bins) is kept compactly.
#include <float.h>

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "snf_internal.h"
#include "device_fft.h"

namespace snf {

namespace {

constexpr int kLongWaves = 16;                 // one workgroup per CU: 16 frames in flight
constexpr int kLongBufBytes = 1088 * 8;        // wave-private LDS: 16 rows x (64 + 4) complex = 64 rows x 17
// table blob (float2 units): window pairs [64][18] | W1024^(L k1) [64][18] | W2048^(kappa + 64 d) [64][10]
// | W64^(b c) [4][16 + 2]; rows padded so that ds_read_b128 is conflict-free (the four rows of the last table
// are broadcast to 16 lanes each: 128-byte rows would put all four on the same banks)
constexpr int kOffWin = 0, kOffTw1 = 64 * 18, kOffTwU = 2 * 64 * 18, kOffTw2 = 2 * 64 * 18 + 64 * 10;
constexpr int kLongTableFloat2 = kOffTw2 + 4 * 18;
constexpr int kLongTableBytes = kLongTableFloat2 * 8;

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// sum over the 64 lanes, the same value in every lane
__device__ __forceinline__ float wave_sum64(float v) {
  v = row_sum16(v);
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
// a wave-uniform 64-bit value as a scalar (keeps the base of the sample loads in SGPRs)
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
  return static_cast<int64_t>((static_cast<unsigned long long>(hi) << 32) | lo);
}
// value of `v` in lane (lane - 1) mod 64: a DPP move with wave_ror:1 (gfx9 keeps the whole-wave rotations;
// checked on the device with tools/ubench_wave_ror.hip), not a trip through the LDS crossbar (ds_bpermute)
__device__ __forceinline__ float from_left_lane(float v, int) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false));
}

}  // namespace

// NJ: element rows a lane can hold inside the window, ceil(win_len / 128): 9 covers 25 ms at 44.1 kHz (and
// every shorter frame), 10 the same at 48 kHz, 16 any window up to 2048 samples.  KIND: the plan's kind
// (the epilogue of one kind per instantiation keeps the scalar register file free of the others' flags)
template <int NJ, int KIND, bool DITHER, bool SNIP>
__global__ __launch_bounds__(kLongWaves * 64) void fbank2048_kernel(
    const MelParams p, const BatchArgs b, const float2* __restrict__ gtab, const int bin_step,
    float* __restrict__ out, const int out_cols, double* __restrict__ energy_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* tab = reinterpret_cast<float2*>(smem);
  for (int i = threadIdx.x; i < kLongTableFloat2; i += blockDim.x) tab[i] = gtab[i];
  __syncthreads();  // the only workgroup-wide barrier: the waves are independent from here on
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float2* buf = reinterpret_cast<float2*>(smem + kLongTableBytes + wid * kLongBufBytes);
  float* ps = reinterpret_cast<float*>(buf);   // power spectrum [1025] (aliases the buffer)
  float* melbuf = ps + 1032;                    // log-mel energies of the frame (MFCC), <= 128
  // The mel phase multiplies a few floats beyond the power spectrum by zero weights, and the padding slots
  // of the transposes (complex index 68 k + 67) are never written: LDS keeps whatever the previous
  // kernel left there, and 0 * NaN = NaN.  Clear the wave's buffer once.
  for (int i = lane; i < kLongBufBytes / 8; i += 64) buf[i] = make_float2(0.0f, 0.0f);
  const int L = p.win_len;
  const float win_len_f = static_cast<float>(L);
  const int left_lane_bytes = ((lane + 63) & 63) * 4;
  typedef int __attribute__((aligned(2))) int_a2;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kLongWaves;
  int64_t g = static_cast<int64_t>(blockIdx.x) * kLongWaves + wid;
  // element j of a lane is inside the window iff 2 (lane + 64 j) < L iff j < (L / 2 - lane + 63) >> 6 (L is
  // even, L / 2 > 63); `nj_any` = the elements some lane of the wave holds (lane 0 holds the most)
  const int nj_any = (L / 2 + 63) >> 6;
  auto clamp_frame = [&](int64_t gi) -> int64_t {
    return gi < b.total_frames ? gi : b.total_frames - 1;
  };
  // sample loads of one frame: SGPR base + lane offset + immediate 256 j.  Elements no lane needs are
  // skipped (their registers keep finite stale values, the zero window weights cancel them: no load ever
  // reaches beyond the window, i.e. beyond the utterance); a lane outside the window at the boundary j
  // re-reads lane 0's element.
#define SNF_LOAD_FRAME(start_, njl_, lane_off_)                                                        \
  do {                                                                                                 \
    const char* __restrict__ wp_ = reinterpret_cast<const char*>(b.wave + uniform64(start_));          \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                   \
      if (j < nj_any) {                                                                                \
        const unsigned off_ = j < (njl_) ? (lane_off_) : 0u;                                           \
        raw[j] = *reinterpret_cast<const int_a2*>(wp_ + off_ + 256 * j);                               \
      }                                                                                                \
    }                                                                                                  \
  } while (0)
  // The samples of the next frame are requested when the transform of the current one is done (the
  // epilogue needs few registers) and converted at the top of the next iteration: their latency hides
  // behind the mel phase without holding 16 registers through the FFT passes.
  int raw[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) raw[j] = 0;
  int64_t start_next = 0;
  int utt_next = 0, edge_next = 0;   // utterance / edge mark of the frame whose samples are in flight
  if (g < b.total_frames) {
    SNF_LOAD_FRAME(b.frame_start[g], (L / 2 - lane + 63) >> 6, 4u * lane);
    utt_next = b.frame_utt[g];
    if (!SNIP) edge_next = b.frame_edge[g];
    start_next = b.frame_start[clamp_frame(g + stride)];
  }
  for (; g < b.total_frames; g += stride) {
    // lane-derived values of the sample phase are recomputed per frame from an opaque copy of the lane
    // index: hoisted out of the loop they would occupy (and spill) dozens of registers
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
    const int njl = (L / 2 - lane_v + 63) >> 6;
    const int kq = lane_v >> 2, bq = lane_v & 3;     // pass C: (k1, quarter of b); pass D: (k1, c)
    const float2* __restrict__ t_win = tab + kOffWin + lane_v * 18;
    const float2* __restrict__ t_tw1 = tab + kOffTw1 + lane_v * 18;
    const float2* __restrict__ t_twu = tab + kOffTwU + lane_v * 10;
    const float2* __restrict__ t_tw2 = tab + kOffTw2 + bq * 18;
    // lane-constant LDS bases (float2 index into the wave's buffer); every access adds a compile-time offset
    const float2* __restrict__ base_lane = buf + lane_v;               // transpose 1 write, exchange write
    float2* __restrict__ base_quad = buf + 68 * kq + bq;             // transpose 1 read, transpose 2 write
    const float2* __restrict__ base_row = buf + 17 * lane_v;           // transpose 2 read
    const float2* __restrict__ base_part =
        buf + (kq == 0 ? ((4 - bq) & 3) : 4 * (16 - kq) + (3 - bq)) + (lane_v == 0 ? 64 : 0);  // partner lane
    const int kappa = kq + 16 * bq;              // lane holds Z[kappa + 64 d] after pass D
    float* __restrict__ ps_lo = ps + kappa;      // P[kappa + 64 d]
    float* __restrict__ ps_hi = ps + (576 - kappa);  // P[1024 - kappa - 64 d] = ps_hi[448 - 64 d]

    auto in_window = [&](int j) -> bool { return j < njl; };
    const int64_t u = utt_next;
    const int edge = edge_next;
    const int warp_v = b.utt_warp ? b.utt_warp[u] : 0;  // (needed by the epilogue only: no wait here)

    // ---- A: samples -> float, DC removal, pre-emphasis, window ---------------------------------------
    float xe[NJ], xo[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      xe[j] = static_cast<float>(static_cast<short>(raw[j] & 0xffff));
      xo[j] = static_cast<float>(raw[j] >> 16);
    }
    if (!SNIP && edge != 0) {
      // [KALDI-UPSTREAM] ExtractWindow, snip_edges = false: samples outside the utterance are reflected
      // (-k - 1 below the start, 2 n - 1 - k beyond the end); only the first and last frames of an
      // utterance take this path, their prefetched samples came from a clamped window
      const int64_t s0 = b.sample_offsets[u], n = b.sample_offsets[u + 1] - s0;
      const int64_t rel = (g - b.frame_offsets[u]) * p.win_shift + p.win_shift / 2 - p.win_len / 2;
      const int16_t* __restrict__ w0 = b.wave + s0;
      float* xf = reinterpret_cast<float*>(buf);  // staged through the wave's buffer: a compact loop
      for (int i = lane_v; i < L; i += 64) {      // instead of 32 unrolled 64-bit reflections
        int64_t k = rel + i;
        while (k < 0 || k >= n) k = k < 0 ? -k - 1 : 2 * n - 1 - k;
        xf[i] = static_cast<float>(w0[k]);
      }
      wave_lds_sync();
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (in_window(j)) {
          const float2 v = base_lane[64 * j];
          xe[j] = v.x;
          xo[j] = v.y;
        }
      }
      lds_wait();
      wave_lds_sync();
    }
    if (DITHER) {  // Kaldi dithers before the DC removal
      const unsigned long long k =
          wave_noise_id(b, u, g - b.frame_offsets[u]) ^ p.seed;  // (batch-independent)
      const unsigned dkey_lo = fmix32(static_cast<unsigned>(k));
      const unsigned dkey_hi = fmix32(static_cast<unsigned>(k >> 32) ^ dkey_lo);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        add_dither_pair(dkey_lo, dkey_hi, static_cast<unsigned>(lane_v + 64 * j), dither_scale(p.dither), xe[j], xo[j]);
      }
    }
    float part = 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) part += in_window(j) ? xe[j] + xo[j] : 0.0f;
    float neg_mean = 0.0f;
    if (p.remove_dc) neg_mean = -wave_sum64(part) / win_len_f;
    float2 z[16];
    float e_lin = 0.0f;
    if (p.need_raw) {  // raw energy: before pre-emphasis and window
      float e_raw = 0.0f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float ae = xe[j] + neg_mean, ao = xo[j] + neg_mean;
        e_raw += in_window(j) ? ae * ae + ao * ao : 0.0f;
      }
      e_lin = wave_sum64(e_raw);
    }
    // left neighbour x[2n-1] of the even sample of element n: the odd sample of element n-1 = lane L-1
    // (same j), lane 63 of j-1 for lane 0.  All 16 exchanges are issued before the first one is used.
    float rot[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) rot[j] = from_left_lane(xo[j] + neg_mean, left_lane_bytes);
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // (window weights in two halves: 16 live registers instead of 32)
      if (8 * h < NJ) {
        float4 win4[4];
        read_quads<4>(t_win + 8 * h, win4);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int j = 8 * h + jj;
          if (j < NJ) {
            const float ae = xe[j] + neg_mean, ao = xo[j] + neg_mean;
            // lane 0, j = 0: x[-1] := x[0] (Kaldi Preemphasize)
            const float ap = lane == 0 ? (j == 0 ? ae : rot[j > 0 ? j - 1 : 0]) : rot[j];
            const float2 w = (jj & 1) ? make_float2(win4[jj >> 1].z, win4[jj >> 1].w)
                                      : make_float2(win4[jj >> 1].x, win4[jj >> 1].y);
            // (elements outside the window hold finite duplicates: their zero window weights make them 0)
            z[j] = make_float2((ae - p.preemph * ap) * w.x, (ao - p.preemph * ae) * w.y);
          }
        }
      }
    }
#pragma unroll
    for (int j = NJ; j < 16; ++j) z[j] = make_float2(0.0f, 0.0f);
    if (p.need_post && !p.need_raw) {
      float e_post = 0.0f;
#pragma unroll
      for (int j = 0; j < 16; ++j) e_post += z[j].x * z[j].x + z[j].y * z[j].y;
      e_lin = wave_sum64(e_post);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- B: pass 1 (FFT over j), twiddle W1024^(L k1), transpose ---------------------------------------
    // (rows j >= NJ are the zero padding: the first layer skips them; butterflies with folded twiddles)
    fft16_lf_head<NJ>(z);
    float4 tw4[8];
    read_quads_whole<8>(t_tw1, tw4);
    lds_wait();
#pragma unroll
    for (int k1 = 1; k1 < 16; ++k1)
      z[k1] = cmul(z[k1], (k1 & 1) ? make_float2(tw4[k1 >> 1].z, tw4[k1 >> 1].w)
                                   : make_float2(tw4[k1 >> 1].x, tw4[k1 >> 1].y));
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) const_cast<float2*>(base_lane)[k1 * 68] = z[k1];
    wave_lds_sync();
    // ---- C: lane (kq, bq): 4-point DFTs over the rows a for b = bq + 4 i, twiddle W64^(b c) --------------
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int a = 0; a < 4; ++a) z[4 * i + a] = base_quad[16 * a + 4 * i];
    float4 tw2q[8];
    read_quads_whole<8>(t_tw2, tw2q);
    lds_wait();
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float2 o0, o1, o2, o3;
      dft4(z[4 * i], z[4 * i + 1], z[4 * i + 2], z[4 * i + 3], o0, o1, o2, o3);
      z[4 * i] = o0;
      z[4 * i + 1] = cmul(o1, make_float2(tw2q[2 * i].z, tw2q[2 * i].w));
      z[4 * i + 2] = cmul(o2, make_float2(tw2q[2 * i + 1].x, tw2q[2 * i + 1].y));
      z[4 * i + 3] = cmul(o3, make_float2(tw2q[2 * i + 1].z, tw2q[2 * i + 1].w));
    }
    // transpose: row r = 4 kq + c (pitch 17) holds b = 0..15
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) base_quad[17 * c + 4 * i] = z[4 * i + c];
    wave_lds_sync();
    read16_b64(base_row, z);
    lds_wait();
    wave_lds_sync();
    // ---- D: pass 3 (FFT over b): z[d] = Z[kappa + 64 d] ---------------------------------------------------
    fft16_lf(z);
    __builtin_amdgcn_sched_barrier(0);
    // ---- E: real-FFT unpack + power.  Partner of k = kappa + 64 d (d < 8) is 1024 - k: register 15 - d of
    // the lane with kappa' = 64 - kappa (own register 16 - d for kappa = 0) ---------------------------------
#pragma unroll
    for (int d = 8; d < 16; ++d) const_cast<float2*>(base_lane)[(d - 8) * 64] = z[d];
    wave_lds_sync();
    float2 zpart[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) zpart[d] = base_part[(7 - d) * 64];
    float4 twuq[4];
    read_quads_whole<4>(t_twu, twuq);
    lds_wait();
    wave_lds_sync();
    float pk[8], pm[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float2 zk = z[d], zp = zpart[d];
      const float2 w = (d & 1) ? make_float2(twuq[d >> 1].z, twuq[d >> 1].w)
                               : make_float2(twuq[d >> 1].x, twuq[d >> 1].y);
      const float c_re = zk.x + zp.x, c_im = zk.y - zp.y;
      const float d_re = zk.y + zp.y, d_im = zp.x - zk.x;
      const float t_re = d_re * w.x - d_im * w.y, t_im = d_re * w.y + d_im * w.x;
      const float a_re = c_re + t_re, a_im = c_im + t_im;
      const float b_re = c_re - t_re, b_im = t_im - c_im;
      pk[d] = 0.25f * (a_re * a_re + a_im * a_im);
      pm[d] = 0.25f * (b_re * b_re + b_im * b_im);
    }
    if (lane == 0) {  // k = 0: DC and Nyquist (Kaldi keeps both)
      const float dc = z[0].x + z[0].y, ny = z[0].x - z[0].y;
      pk[0] = dc * dc;
      pm[0] = ny * ny;
    }
    const float p512 = z[8].x * z[8].x + z[8].y * z[8].y;  // self-paired bin 512: lane 0, register 8
    if (bin_step == 1) {
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        ps_lo[64 * d] = pk[d];
        ps_hi[448 - 64 * d] = pm[d];
      }
      if (lane == 0) ps[512] = p512;
    } else {
      // zero-extended 1024-sample frame: its spectrum is the even bins, kept compactly (bin k at k / 2)
      if ((kappa & 1) == 0) {
        float* __restrict__ pe_lo = ps + (kappa >> 1);
        float* __restrict__ pe_hi = ps + (288 - (kappa >> 1));  // (1024 - kappa - 64 d) / 2 = pe_hi[224 - 32 d]
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          pe_lo[32 * d] = pk[d];
          pe_hi[224 - 32 * d] = pm[d];
        }
      }
      if (lane == 0) ps[256] = p512;
    }
    wave_lds_sync();
    if (KIND == SNF_KIND_FBANK && !p.use_power) {  // magnitude spectrum
      for (int k = lane; k <= p.half; k += 64) ps[k] = sqrtf(ps[k]);
      wave_lds_sync();
    }

    // next frame: samples (converted at the top of the next iteration), start offset of the one after
    SNF_LOAD_FRAME(start_next, njl, 4u * lane_v);
    {
      const int64_t gn = clamp_frame(g + stride);
      utt_next = b.frame_utt[gn];
      if (!SNIP) edge_next = b.frame_edge[gn];
    }
    start_next = b.frame_start[clamp_frame(g + 2 * stride)];
    // ---- F: epilogue (same conventions as mel_features_generic_kernel) -----------------------------------
    float log_energy = 0.0f;
    if (KIND == SNF_KIND_PLP) {
      // shennong's PLP floors with float64 eps and takes a double log (reference plp.py:191-193)
      if ((p.need_raw || p.need_post) && lane == 0)
        energy_out[g] = static_cast<double>(e_lin);  // (plp_tail_kernel takes the double log)
    } else if (p.need_raw || p.need_post) {
      log_energy = fast_log(floor_eps(e_lin));
      if (p.has_floor && log_energy < p.log_energy_floor) log_energy = p.log_energy_floor;
    }
    float* __restrict__ row = out + g * static_cast<int64_t>(out_cols);
    if (KIND == SNF_KIND_SPECTROGRAM) {
      for (int k = lane; k <= p.half; k += 64) {
        float v = fast_log(floor_eps(ps[k]));
        if (k == 0) v = log_energy;
        row[k] = v;
      }
    } else {
      const int nb = p.num_bins;
      const int warp_id = __builtin_amdgcn_readfirstlane(warp_v);
      const int* __restrict__ mfirst = p.mel_first + warp_id * nb;
      const int* __restrict__ msize = p.mel_size + warp_id * nb;
      const int mel_col = (KIND == SNF_KIND_FBANK && p.use_energy && !p.htk_compat) ? 1 : 0;
      // teams of 8 lanes per mel bin, 8 bins per round; the weights come as 16-byte vectors from the plan's
      // ordinary device tables (L1 / L2 resident; the table carries 4 floats of padding), the power spectrum
      // from LDS.  The bin indices of the next round are requested a round ahead.
      const int* __restrict__ moff32 = p.mel_off32 + warp_id * nb;
      const int team = lane >> 3, tl = lane & 7;
      bool active = team < nb;
      int first = active ? mfirst[team] : 0, size = active ? msize[team] : 0, woff = active ? moff32[team] : 0;
      for (int m0 = 0; m0 < nb; m0 += 8) {
        const int m = m0 + team, mn = m + 8;
        const bool active_n = mn < nb;
        const int first_n = active_n ? mfirst[mn] : 0, size_n = active_n ? msize[mn] : 0,
                  woff_n = active_n ? moff32[mn] : 0;
        // The team reads the filter in slices of 32 taps: lane tl owns taps 32 e + 4 tl + c (c < 4), one
        // 16-byte weight load per slice (128 contiguous bytes per team) from a table in which every filter
        // starts at a multiple of 4 bins and is zero-padded to whole slices: no per-tap test.  Inside a
        // group of 4 taps the table is rotated by the team index, and so is the order of the reads: the
        // four teams of a 32-lane LDS group then sit on the four residues modulo 4 of the banks, the 8
        // lanes of a team on 8 different banks of their residue - every read is conflict-free.  A lane
        // whose filter has fewer slices than the widest of the round multiplies finite buffer contents
        // by the zero slice.
        const int lead = first & 3, slices = (size + lead + 31) >> 5;
        const float* __restrict__ wt = p.mel_w32 + woff + 4 * tl;
        const float* __restrict__ wz = p.mel_w32 + 4 * tl;
        const float* __restrict__ pb0 = ps + (first - lead) + 4 * tl;
        const int rot = team & 3;
        const float* __restrict__ pb[4] = {pb0 + rot, pb0 + ((rot + 1) & 3), pb0 + ((rot + 2) & 3),
                                           pb0 + ((rot + 3) & 3)};
        float acc = 0.0f;
        for (int e0 = 0; __any(e0 < slices); e0 += 4) {
          f32x4_a4 w[4];
          float pv[16];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            w[i] = *reinterpret_cast<const f32x4_a4*>(e0 + i < slices ? wt + 32 * (e0 + i) : wz);
#pragma unroll
          for (int e = 0; e < 16; ++e) pv[e] = pb[e & 3][32 * (e0 + (e >> 2))];
          lds_wait();
#pragma unroll
          for (int e = 0; e < 16; ++e) acc += w[e >> 2][e & 3] * pv[e];
        }
        acc += dpp_row_ror<0xB1>(acc);   // quad_perm [1,0,3,2]
        acc += dpp_row_ror<0x4E>(acc);   // quad_perm [2,3,0,1]
        acc += dpp_row_ror<0x141>(acc);  // row_half_mirror: the other quad of the team
        if (active && tl == 0) {
          if (KIND == SNF_KIND_FBANK) {
            row[mel_col + m] = p.use_log ? fast_log(floor_eps(acc)) : acc;
          } else if (KIND == SNF_KIND_MFCC) {
            melbuf[m] = fast_log(floor_eps(acc));
          } else {  // PLP: linear mel energies, the recipe continues in plp_tail_kernel
            row[m] = acc;
          }
        }
        active = active_n;
        first = first_n;
        size = size_n;
        woff = woff_n;
      }
      if (KIND == SNF_KIND_FBANK && p.use_energy && lane == 0)
        row[p.htk_compat ? nb : 0] = log_energy;
      if (KIND == SNF_KIND_MFCC) {
        wave_lds_sync();
        // DCT-II: teams of 4 lanes per cepstral coefficient, 16 coefficients per round
        const int ct = lane >> 2, cl = lane & 3;
        for (int c0 = 0; c0 < p.num_ceps; c0 += 16) {
          const int c = c0 + ct;
          const bool ca = c < p.num_ceps;
          const float* __restrict__ dm = p.dct + (ca ? c : 0) * nb;
          float v = 0.0f;
          for (int m0 = 0; m0 < nb; m0 += 32) {  // 8 coefficients per lane in flight
            float dv[8], mv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int m = m0 + cl + 4 * e;
              dv[e] = dm[m < nb ? m : 0];
              mv[e] = melbuf[m < nb ? m : 0];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v += (m0 + cl + 4 * e < nb) ? dv[e] * mv[e] : 0.0f;
          }
          v += dpp_row_ror<0xB1>(v);
          v += dpp_row_ror<0x4E>(v);
          if (ca && cl == 0) {
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
    }
    wave_lds_sync();  // the next frame reuses the buffer
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool fbank2048_eligible(const MelParams& mp) {
  if (getenv("SNF_DISABLE_FAST2048")) return false;
  if (!mp.pow2 || (mp.padded != 2048 && mp.padded != 1024)) return false;
  if (mp.win_len & 1) return false;
  if (mp.kind != SNF_KIND_FBANK && mp.kind != SNF_KIND_MFCC && mp.kind != SNF_KIND_PLP &&
      mp.kind != SNF_KIND_SPECTROGRAM)
    return false;
  if (mp.kind != SNF_KIND_SPECTROGRAM && mp.num_bins > 128) return false;
  return true;
}

// Window pairs and twiddles of the kernel, laid out per lane (float2 units, see kOff*)
void fbank2048_tables(const MelParams& mp, const std::vector<float>& window, std::vector<float>* blob) {
  constexpr double kTwoPi = 6.283185307179586476925286766559005;
  blob->assign(static_cast<size_t>(kLongTableFloat2) * 2, 0.0f);
  float* t = blob->data();
  auto put = [&](int index, double re, double im) {
    t[2 * index] = static_cast<float>(re);
    t[2 * index + 1] = static_cast<float>(im);
  };
  for (int lane = 0; lane < 64; ++lane) {
    for (int j = 0; j < 16; ++j) {
      const int n = lane + 64 * j;
      const double w0 = 2 * n < mp.win_len ? window[2 * n] : 0.0;
      const double w1 = 2 * n + 1 < mp.win_len ? window[2 * n + 1] : 0.0;
      put(kOffWin + lane * 18 + j, w0, w1);
      const double a1 = -kTwoPi * ((lane * j) % 1024) / 1024.0;  // W1024^(lane k1), k1 = j
      put(kOffTw1 + lane * 18 + j, std::cos(a1), std::sin(a1));
    }
    const int kappa = (lane >> 2) + 16 * (lane & 3);
    for (int d = 0; d < 8; ++d) {
      const double a = -kTwoPi * (kappa + 64 * d) / 2048.0;
      put(kOffTwU + lane * 10 + d, std::cos(a), std::sin(a));
    }
  }
  for (int bq = 0; bq < 4; ++bq)
    for (int i = 0; i < 4; ++i)
      for (int c = 0; c < 4; ++c) {
        const double a = -kTwoPi * (((bq + 4 * i) * c) % 64) / 64.0;
        put(kOffTw2 + bq * 18 + i * 4 + c, std::cos(a), std::sin(a));
      }
}

int launch_fbank2048(const MelParams& p, const BatchArgs& b, const float* tables, float* out, int out_cols,
                     double* energy_out, hipStream_t stream) {
  if (b.total_frames <= 0) return SNF_OK;
  const int lds = kLongTableBytes + kLongWaves * kLongBufBytes;
  int64_t blocks = (b.total_frames + kLongWaves - 1) / kLongWaves;
  if (blocks > 256) blocks = 256;  // one persistent workgroup per CU, grid-stride over the frames
  const int rows = (p.win_len + 127) / 128;
#define SNF_LONG4(NJ_, KIND_, DI_, SN_)                                                                             \
  do {                                                                                                    \
    SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fbank2048_kernel<NJ_, KIND_, DI_, SN_>),       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));                 \
    hipLaunchKernelGGL((fbank2048_kernel<NJ_, KIND_, DI_, SN_>), dim3(static_cast<unsigned>(blocks)),               \
                       dim3(kLongWaves * 64), lds, stream, p, b, reinterpret_cast<const float2*>(tables), \
                       2048 / p.padded, out, out_cols, energy_out);                                       \
  } while (0)
#define SNF_LONG3(NJ_, KIND_, DI_)                                                                         \
  do {                                                                                                    \
    if (p.snip_edges) SNF_LONG4(NJ_, KIND_, DI_, true);                                                   \
    else SNF_LONG4(NJ_, KIND_, DI_, false);                                                               \
  } while (0)
#define SNF_LONG2(NJ_, KIND_)                                                                              \
  do {                                                                                                    \
    if (p.dither != 0.0f) SNF_LONG3(NJ_, KIND_, true);                                                    \
    else SNF_LONG3(NJ_, KIND_, false);                                                                    \
  } while (0)
#define SNF_LONG(NJ_)                                                                                      \
  do {                                                                                                    \
    if (p.kind == SNF_KIND_FBANK) SNF_LONG2(NJ_, SNF_KIND_FBANK);                                         \
    else if (p.kind == SNF_KIND_MFCC) SNF_LONG2(NJ_, SNF_KIND_MFCC);                                      \
    else if (p.kind == SNF_KIND_PLP) SNF_LONG2(NJ_, SNF_KIND_PLP);                                        \
    else SNF_LONG2(NJ_, SNF_KIND_SPECTROGRAM);                                                            \
  } while (0)
  if (rows <= 9) SNF_LONG(9);
  else if (rows <= 10) SNF_LONG(10);
  else SNF_LONG(16);
#undef SNF_LONG2
#undef SNF_LONG3
#undef SNF_LONG4
#undef SNF_LONG
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

}  // namespace snf
