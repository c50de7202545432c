// Register-resident spectrogram / filterbank / MFCC / PLP-mel kernel for frames that pad to 1024 samples
// (25 ms windows at 22.05 and 32 kHz; the reference resamples nothing, it frames whatever rate the file
// has: shennong/processor/base.py:408-436), on gfx950.  Round 6: these frames used to run as the
// 2048-point transform of the zero-extended frame (kernels_fbank2048.hip: twice the arithmetic a frame
// needs, and odd window lengths - 551 samples at 22.05 kHz - fell to the generic kernel).
//
//   wave64 = TWO frames of one utterance, packed as one complex signal z[n] = x_a[n] + i x_b[n], n < 1024.
//   The three register passes of the long-frame kernel (16 x 4 x 16 around two LDS transposes) transform
//   it; the spectra separate without a twiddle:
//       X_a[k] = (Z[k] + conj Z[1024 - k]) / 2,   X_b[k] = (Z[k] - conj Z[1024 - k]) / 2i,   k <= 512.
//   A  lane L loads x[L + 64 j] of both frames (16-bit loads, 128 contiguous bytes per wave instruction;
//      the next pair's samples are requested when this pair's transform is done); per frame: DC removal
//      over the wave, pre-emphasis (left neighbour through ds_bpermute), window
//   B-D as kernels_fbank2048.hip (same tables, same index maps: tools/model_fbank2048.py)
//   E  the upper half of Z meets its partner through LDS; two power spectra (513 bins each) to LDS
//   F  the epilogue of the long-frame kernel, once per frame
// Pairs are formed INSIDE an utterance (PairRec: frames 2 m and 2 m + 1; an odd last frame is transformed
// alone, its partner is zero), so the features of an utterance do not depend on what else is in the batch.
// The two real transforms share their roundings: the error floor of both spectra is set by the louder
// frame.  A pair whose windowed energies differ by more than `kSplitRatio` (a quiet frame beside an onset;
// digital silence beside anything) is therefore transformed as two single frames - two trips through
// B-E, decided per wave, the same arithmetic as a frame paired with zero.
#include <float.h>

#include <cmath>
#include <cstdlib>
#include <vector>

#include "snf_internal.h"
#include "device_fft.h"

namespace snf {

namespace {

constexpr int kPairWaves = 16;                 // one workgroup per CU: 16 pairs in flight
constexpr int kPairBufBytes = 1088 * 8;        // wave-private LDS: 16 rows x (64 + 4) complex = 64 rows x 17
// table blob (float2 units): window (w, w) [64][18] | W1024^(L k1) [64][18] | W64^(b c) [4][16 + 2]
constexpr int kOffWin = 0, kOffTw1 = 64 * 18, kOffTw2 = 2 * 64 * 18;
constexpr int kPairTableFloat2 = kOffTw2 + 4 * 18;
constexpr int kPairTableBytes = kPairTableFloat2 * 8;
constexpr int kSpecB = 544;                    // float offset of frame b's power spectrum in the wave's buffer
constexpr int kMelBuf = 1152;                  // ... of the log-mel energies (MFCC), 128 per frame
constexpr float kSplitRatio = 8.0f;            // windowed-energy ratio beyond which a pair is split

__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum64(float v) {
  v = row_sum16(v);
  return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
  return static_cast<int64_t>((static_cast<unsigned long long>(hi) << 32) | lo);
}
// value of `v` in lane (lane - 1) mod 64 (a DPP move with wave_ror:1: kernels_fbank2048.hip)
__device__ __forceinline__ float from_left_lane(float v, int) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x13C, 0xf, 0xf, false));
}

}  // namespace

// NJ: element rows a lane can hold inside the window, ceil(win_len / 64): 9 covers 25 ms at 22.05 kHz,
// 13 the same at 32 kHz, 16 any window up to 1024 samples.
template <int NJ, int KIND, bool DITHER, bool SNIP>
__global__ __launch_bounds__(kPairWaves * 64) void fbank1024x2_kernel(
    const MelParams p, const BatchArgs b, const float2* __restrict__ gtab, const float split_ratio,
    float* __restrict__ out, const int out_cols, double* __restrict__ energy_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* tab = reinterpret_cast<float2*>(smem);
  for (int i = threadIdx.x; i < kPairTableFloat2; i += blockDim.x) tab[i] = gtab[i];
  __syncthreads();  // the only workgroup-wide barrier: the waves are independent from here on
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float2* buf = reinterpret_cast<float2*>(smem + kPairTableBytes + wid * kPairBufBytes);
  float* ps = reinterpret_cast<float*>(buf);   // power spectra: frame a at 0, frame b at kSpecB
  // (the mel phase multiplies a few floats beyond a spectrum by zero weights, and the padding slots of the
  // transposes are never written: clear the wave's buffer once - 0 * NaN = NaN)
  for (int i = lane; i < kPairBufBytes / 8; i += 64) buf[i] = make_float2(0.0f, 0.0f);
  const int L = p.win_len;
  const float win_len_f = static_cast<float>(L);
  const int left_lane_bytes = ((lane + 63) & 63) * 4;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * kPairWaves;
  int64_t g = static_cast<int64_t>(blockIdx.x) * kPairWaves + wid;
  const int64_t last_pair = b.n_pairs - 1;
  auto rec_of = [&](int64_t pi) -> const PairRec* { return b.pair_tab + (pi < last_pair ? pi : last_pair); };
  // Sample ingest: a frame's samples come in as 16-byte pieces, lane l fetching pieces l and l + 64 (two wave
  // instructions per frame: a 16-bit load per element would be 13 - the texture addresser takes a wave
  // instruction per 64 addresses whatever their width, 0.26 of 1.11 ms by ablation), are laid down in the wave's
  // LDS buffer at the top of the next trip (frame a at byte 0, frame b at byte 2048: the transposes need the
  // buffer later) and read back element by element, lane L taking n = L + 64 j and its left neighbour n - 1.
  // The pieces that would reach beyond the frame are not fetched whole: the last fb mod 16 bytes of a frame
  // come through 16-bit loads of the first lanes - no load ever reaches beyond the window, i.e. beyond the
  // utterance; lanes without a piece re-read piece 0.
  typedef int int4_a2 __attribute__((ext_vector_type(4), aligned(2)));
  const int n_full = (2 * L) >> 4;            // whole 16-byte pieces of a frame (<= 128)
  const int n_tail = ((2 * L) & 15) >> 1;     // samples behind them (< 8)
  int4_a2 qa0 = {0, 0, 0, 0}, qa1 = {0, 0, 0, 0}, qb0 = {0, 0, 0, 0}, qb1 = {0, 0, 0, 0};
  int tail_ab = 0;                            // tail samples: frame a in the low half, frame b in the high half
#define SNF_LOAD_FRAMES(start_a_, start_b_, lane_)                                                           \
  do {                                                                                                       \
    const char* __restrict__ wa_ = reinterpret_cast<const char*>(b.wave + uniform64(start_a_));              \
    const char* __restrict__ wb_ = reinterpret_cast<const char*>(b.wave + uniform64(start_b_));              \
    const unsigned o0_ = (lane_) < n_full ? 16u * (lane_) : 0u;                                              \
    const unsigned o1_ = (lane_) + 64 < n_full ? 16u * ((lane_) + 64) : 0u;                                  \
    qa0 = *reinterpret_cast<const int4_a2*>(wa_ + o0_);                                                      \
    qb0 = *reinterpret_cast<const int4_a2*>(wb_ + o0_);                                                      \
    if (n_full > 64) {                                                                                       \
      qa1 = *reinterpret_cast<const int4_a2*>(wa_ + o1_);                                                    \
      qb1 = *reinterpret_cast<const int4_a2*>(wb_ + o1_);                                                    \
    }                                                                                                        \
    if (n_tail != 0) {                                                                                       \
      const unsigned ot_ = 16u * n_full + ((lane_) < n_tail ? 2u * (lane_) : 0u);                            \
      const unsigned lo_ = *reinterpret_cast<const unsigned short*>(wa_ + ot_);                              \
      const unsigned hi_ = *reinterpret_cast<const unsigned short*>(wb_ + ot_);                              \
      tail_ab = static_cast<int>(lo_ | (hi_ << 16));                                                         \
    }                                                                                                        \
  } while (0)
  int64_t frame_next = 0;
  int utt1_next = 1, flags_next = 0;
  longlong2 starts_now = make_longlong2(0, 0), starts_after = make_longlong2(0, 0);
  if (g <= last_pair) {
    const PairRec* r = rec_of(g);
    const longlong2 st = reinterpret_cast<const longlong2*>(r)[0];
    SNF_LOAD_FRAMES(st.x, st.y, lane);
    starts_now = st;
    frame_next = r->frame_a;
    utt1_next = r->utt1;
    flags_next = r->flags;
    starts_after = reinterpret_cast<const longlong2*>(rec_of(g + stride))[0];
  }
  for (; g <= last_pair;) {
    // lane-derived values are recomputed per pair from an opaque copy of the lane index: hoisted out of the
    // loop they would occupy (and spill) dozens of registers
    int lane_v = lane;
    asm volatile("" : "+v"(lane_v));
    const int njl = (L - lane_v + 63) >> 6;
    const int kq = lane_v >> 2, bq = lane_v & 3;     // pass C: (k1, quarter of b); pass D: (k1, c)
    const float2* __restrict__ t_win = tab + kOffWin + lane_v * 18;
    const float2* __restrict__ t_tw1 = tab + kOffTw1 + lane_v * 18;
    const float2* __restrict__ t_tw2 = tab + kOffTw2 + bq * 18;
    const float2* __restrict__ base_lane = buf + lane_v;               // transpose 1 write, exchange write
    float2* __restrict__ base_quad = buf + 68 * kq + bq;             // transpose 1 read, transpose 2 write
    const float2* __restrict__ base_row = buf + 17 * lane_v;           // transpose 2 read
    const float2* __restrict__ base_part =
        buf + (kq == 0 ? ((4 - bq) & 3) : 4 * (16 - kq) + (3 - bq)) + (lane_v == 0 ? 64 : 0);  // partner lane
    const int kappa = kq + 16 * bq;              // lane holds Z[kappa + 64 d] after pass D

    auto in_window = [&](int j) -> bool { return j < njl; };
    // the samples of this trip's two frames: registers -> LDS (whole pieces, then the tail)
    {
      char* sb_ = reinterpret_cast<char*>(buf);
      if (lane_v < n_full) {
        *reinterpret_cast<int4*>(sb_ + 16 * lane_v) = make_int4(qa0.x, qa0.y, qa0.z, qa0.w);
        *reinterpret_cast<int4*>(sb_ + 2048 + 16 * lane_v) = make_int4(qb0.x, qb0.y, qb0.z, qb0.w);
      }
      if (lane_v + 64 < n_full) {
        *reinterpret_cast<int4*>(sb_ + 16 * (lane_v + 64)) = make_int4(qa1.x, qa1.y, qa1.z, qa1.w);
        *reinterpret_cast<int4*>(sb_ + 2048 + 16 * (lane_v + 64)) = make_int4(qb1.x, qb1.y, qb1.z, qb1.w);
      }
      if (lane_v < n_tail) {
        *reinterpret_cast<short*>(sb_ + 16 * n_full + 2 * lane_v) = static_cast<short>(tail_ab & 0xffff);
        *reinterpret_cast<short*>(sb_ + 2048 + 16 * n_full + 2 * lane_v) = static_cast<short>(tail_ab >> 16);
      }
      wave_lds_sync();
    }
    const int64_t g_a = frame_next;
    const int64_t start_b = starts_now.y;
    const int64_t u = utt1_next - 1;
    const int flags = flags_next;
    const bool has_b = (flags & 4) != 0;         // (wave-uniform)
    const int warp_v = b.utt_warp ? b.utt_warp[u] : 0;  // (needed by the epilogue only: no wait here)

    // ---- A: samples -> float, DC removal, pre-emphasis, window; one frame after the other ---------------
    // (frame f of the pair from its sample registers -> windowed frame, its energy as the options name it,
    // its windowed energy)
    auto prepare = [&](const int f, float (&wf)[NJ], float& e_lin_f, float& e_win_f) __attribute__((always_inline)) {
      short* __restrict__ sf = reinterpret_cast<short*>(reinterpret_cast<char*>(buf) + 2048 * f);
      if (!SNIP && (flags & (1 << f)) != 0) {
        // [KALDI-UPSTREAM] ExtractWindow, snip_edges = false: samples outside the utterance are reflected
        // (-k - 1 below the start, 2 n - 1 - k beyond the end); only the first and last frames of an
        // utterance take this path (their prefetched samples came from a clamped window): the frame's place
        // in LDS is written again
        const int64_t s0 = b.sample_offsets[u], n = b.sample_offsets[u + 1] - s0;
        const int64_t gf = g_a + f;
        const int64_t rel = (gf - b.frame_offsets[u]) * p.win_shift + p.win_shift / 2 - p.win_len / 2;
        const int16_t* __restrict__ w0 = b.wave + s0;
        wave_lds_sync();
        for (int i = lane_v; i < L; i += 64) {
          int64_t k = rel + i;
          while (k < 0 || k >= n) k = k < 0 ? -k - 1 : 2 * n - 1 - k;
          sf[i] = w0[k];
        }
        wave_lds_sync();
      }
      // element j of the lane: sample n = lane + 64 j, and (no dither) its left neighbour n - 1 (n = 0: Kaldi's
      // Preemphasize takes x[0] for x[-1])
      float x[NJ], xl[NJ];
      const short* __restrict__ sl = sf + (lane_v == 0 ? 0 : lane_v - 1);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        x[j] = static_cast<float>(sf[lane_v + 64 * j]);
        if (!DITHER) xl[j] = static_cast<float>(j == 0 ? sl[0] : (sf - 1)[lane_v + 64 * j]);
      }
      lds_wait();
      if (DITHER) {  // Kaldi dithers before the DC removal
        const unsigned long long k = wave_noise_id(b, u, g_a + f - b.frame_offsets[u]) ^ p.seed;
        const unsigned dkey_lo = fmix32(static_cast<unsigned>(k));
        const unsigned dkey_hi = fmix32(static_cast<unsigned>(k >> 32) ^ dkey_lo);
#pragma unroll
        for (int j = 0; j < NJ; j += 2) {
          float spare = 0.0f;
          add_dither_pair(dkey_lo, dkey_hi, static_cast<unsigned>(lane_v + 64 * (j >> 1)), dither_scale(p.dither),
                          x[j], j + 1 < NJ ? x[j + 1] : spare);
        }
      }
      float part = 0.0f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) part += in_window(j) ? x[j] : 0.0f;
      float neg_mean = 0.0f;
      if (p.remove_dc) neg_mean = -wave_sum64(part) / win_len_f;
      if (p.need_raw) {  // raw energy: before pre-emphasis and window
        float e_raw = 0.0f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float a = x[j] + neg_mean;
          e_raw += in_window(j) ? a * a : 0.0f;
        }
        e_lin_f = wave_sum64(e_raw);
      }
      // left neighbour x[n - 1] of a dithered frame (the noise of a sample is drawn once): the same row of lane
      // L - 1, row j - 1 of lane 63 for lane 0
      float rot[NJ];
      if (DITHER) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) rot[j] = from_left_lane(x[j] + neg_mean, left_lane_bytes);
      }
      float e_post = 0.0f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // (window weights in two halves)
        if (8 * h < NJ) {
          float4 win4[4];
          read_quads<4>(t_win + 8 * h, win4);
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const int j = 8 * h + jj;
            if (j < NJ) {
              const float a = x[j] + neg_mean;
              // lane 0, j = 0: x[-1] := x[0] (Kaldi Preemphasize)
              const float ap = DITHER ? (lane == 0 ? (j == 0 ? a : rot[j > 0 ? j - 1 : 0]) : rot[j])
                                      : xl[j] + neg_mean;
              const float w = (jj & 1) ? win4[jj >> 1].z : win4[jj >> 1].x;
              // (elements outside the window hold finite duplicates: their zero window weights make them 0)
              const float v = (a - p.preemph * ap) * w;
              e_post += v * v;
              wf[j] = v;
            }
          }
        }
      }
      e_win_f = wave_sum64(e_post);
      if (p.need_post && !p.need_raw) e_lin_f = e_win_f;
    };
    float wa[NJ], wb[NJ];          // windowed frames
    float e_lin[2] = {0.0f, 0.0f}; // frame energies (raw or windowed, whichever the options name)
    float e_win[2] = {0.0f, 0.0f}; // windowed energies (the pairing decision)
    prepare(0, wa, e_lin[0], e_win[0]);
    prepare(1, wb, e_lin[1], e_win[1]);
    // an odd last frame is transformed alone; a pair of very different energies as two single frames: this trip
    // takes frame a with a zero partner, the NEXT trip of the loop takes frame b the same way (its samples come
    // in again: the pair's record is replayed with frame b in the first seat, see the prefetch below)
    bool split = false;
    if (has_b) {
      const float lo = fminf(e_win[0], e_win[1]), hi = fmaxf(e_win[0], e_win[1]);
      split = hi > split_ratio * lo;
    }
    split = __builtin_amdgcn_readfirstlane(split ? 1 : 0) != 0;
    const bool two = has_b && !split;    // (wave-uniform) both spectra come out of this transform
    __builtin_amdgcn_sched_barrier(0);
    {
      float2 z[16];
#pragma unroll
      for (int j = 0; j < NJ; ++j) z[j] = make_float2(wa[j], two ? wb[j] : 0.0f);
#pragma unroll
      for (int j = NJ; j < 16; ++j) z[j] = make_float2(0.0f, 0.0f);
      // ---- B: pass 1 (FFT over j), twiddle W1024^(L k1), transpose ---------------------------------------
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
      // ---- C: lane (kq, bq): 4-point DFTs over the rows a for b = bq + 4 i, twiddle W64^(b c) -------------
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
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) base_quad[17 * c + 4 * i] = z[4 * i + c];
      wave_lds_sync();
      read16_b64(base_row, z);
      lds_wait();
      wave_lds_sync();
      // ---- D: pass 3 (FFT over b): z[d] = Z[kappa + 64 d] -------------------------------------------------
      fft16_lf(z);
      __builtin_amdgcn_sched_barrier(0);
      // ---- E: the two spectra.  Partner of k = kappa + 64 d (d < 8) is 1024 - k: register 15 - d of the lane
      // with kappa' = 64 - kappa (own register 16 - d for kappa = 0) ---------------------------------------
#pragma unroll
      for (int d = 8; d < 16; ++d) const_cast<float2*>(base_lane)[(d - 8) * 64] = z[d];
      wave_lds_sync();
      float2 zpart[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) zpart[d] = base_part[(7 - d) * 64];
      lds_wait();
      wave_lds_sync();
      float pa[8], pb[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const float2 zk = z[d], zp = zpart[d];
        const float a_re = zk.x + zp.x, a_im = zk.y - zp.y;   // 2 X_a[k]
        const float b_re = zk.y + zp.y, b_im = zp.x - zk.x;   // 2 X_b[k]
        pa[d] = 0.25f * (a_re * a_re + a_im * a_im);
        pb[d] = 0.25f * (b_re * b_re + b_im * b_im);
      }
      if (lane == 0) {  // k = 0: Z[0] = sum x_a + i sum x_b
        pa[0] = z[0].x * z[0].x;
        pb[0] = z[0].y * z[0].y;
      }
      // self-paired bin 512 (lane 0, register 8): X_a[512] = Re Z[512], X_b[512] = Im Z[512]
      const float pa512 = z[8].x * z[8].x, pb512 = z[8].y * z[8].y;
      float* __restrict__ ps_a = ps + kappa;
#pragma unroll
      for (int d = 0; d < 8; ++d) ps_a[64 * d] = pa[d];
      if (lane == 0) ps_a[512] = pa512;
      if (two) {
        float* __restrict__ ps_b = ps + kSpecB + kappa;
#pragma unroll
        for (int d = 0; d < 8; ++d) ps_b[64 * d] = pb[d];
        if (lane == 0) ps_b[512] = pb512;
      }
      wave_lds_sync();
    }
    if (KIND == SNF_KIND_FBANK && !p.use_power) {  // magnitude spectrum
      for (int k = lane; k <= p.half; k += 64) {
        ps[k] = sqrtf(ps[k]);
        if (two) ps[kSpecB + k] = sqrtf(ps[kSpecB + k]);
      }
      wave_lds_sync();
    }

    // next trip: the next pair - or, after a split, frame b of this one in the first seat of a record without a
    // second frame (bit 3: the pair index does not advance).  Samples now (converted at the top of the next
    // trip), the record of the pair after that
    {
      const int64_t ns_a = split ? start_b : starts_after.x, ns_b = split ? start_b : starts_after.y;
      SNF_LOAD_FRAMES(ns_a, ns_b, lane_v);
      if (split) {
        frame_next = g_a + 1;
        flags_next = ((flags >> 1) & 1) | 8;
        starts_now = make_longlong2(start_b, start_b);
      } else {
        const PairRec* rn = rec_of(g + stride);
        const int4 meta = reinterpret_cast<const int4*>(rn)[1];   // {frame_a lo, frame_a hi, utt1, flags}
        frame_next = (static_cast<int64_t>(meta.y) << 32) | static_cast<unsigned>(meta.x);
        utt1_next = meta.z;
        flags_next = meta.w;
        starts_now = starts_after;
        starts_after = reinterpret_cast<const longlong2*>(rec_of(g + 2 * stride))[0];
      }
    }

    // ---- F: epilogue (same conventions as mel_features_generic_kernel).  The two frames of a pair belong to one
    // utterance - one warp factor, one set of filters: the weights are fetched once per round for both -----------
    float log_energy[2] = {0.0f, 0.0f};
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      if (f == 1 && !two) break;
      if (KIND == SNF_KIND_PLP) {
        // shennong's PLP floors with float64 eps and takes a double log (reference plp.py:191-193)
        if ((p.need_raw || p.need_post) && lane == 0)
          energy_out[g_a + f] = static_cast<double>(e_lin[f]);  // (plp_tail_kernel takes the double log)
      } else if (p.need_raw || p.need_post) {
        log_energy[f] = fast_log(floor_eps(e_lin[f]));
        if (p.has_floor && log_energy[f] < p.log_energy_floor) log_energy[f] = p.log_energy_floor;
      }
    }
    float* __restrict__ row_a = out + g_a * static_cast<int64_t>(out_cols);
    float* __restrict__ row_b = row_a + out_cols;
    if (KIND == SNF_KIND_SPECTROGRAM) {
      for (int k = lane; k <= p.half; k += 64) {
        float va = fast_log(floor_eps(ps[k]));
        if (k == 0) va = log_energy[0];
        row_a[k] = va;
        if (two) {
          float vb = fast_log(floor_eps(ps[kSpecB + k]));
          if (k == 0) vb = log_energy[1];
          row_b[k] = vb;
        }
      }
    } else {
      const int nb = p.num_bins;
      const int warp_id = __builtin_amdgcn_readfirstlane(warp_v);
      const int* __restrict__ mfirst = p.mel_first + warp_id * nb;
      const int* __restrict__ msize = p.mel_size + warp_id * nb;
      const int mel_col = (KIND == SNF_KIND_FBANK && p.use_energy && !p.htk_compat) ? 1 : 0;
      float* __restrict__ melbuf = ps + kMelBuf;   // (MFCC) log-mel energies: frame a at 0, frame b at 128
      // teams of 8 lanes per mel bin, 8 bins per round; weights as 16-byte vectors from the plan's device
      // tables (32-tap slices, zero-padded, rotated by the team index: kernels_fbank2048.hip), the power
      // spectra from LDS, two slices per trip (most filters of a 513-bin spectrum fit one or two).  The bin
      // indices of the next round are requested a round ahead.
      const int* __restrict__ moff32 = p.mel_off32 + warp_id * nb;
      const int team = lane >> 3, tl = lane & 7;
      const float* __restrict__ wz = p.mel_w32 + 4 * tl;
      // the first two slices of a filter's weights: zero slices beyond its end
      auto weights_of = [&](int first_, int size_, int woff_, f32x4_a4 (&w_)[2]) __attribute__((always_inline)) {
        const int slices_ = (size_ + (first_ & 3) + 31) >> 5;
        const float* __restrict__ wt_ = p.mel_w32 + woff_ + 4 * tl;
#pragma unroll
        for (int i = 0; i < 2; ++i) w_[i] = *reinterpret_cast<const f32x4_a4*>(i < slices_ ? wt_ + 32 * i : wz);
      };
      // (bin indices two rounds ahead, weights one round ahead: a round waits for neither)
      bool active = team < nb;
      int first = active ? mfirst[team] : 0, size = active ? msize[team] : 0, woff = active ? moff32[team] : 0;
      const bool active_1 = team + 8 < nb;
      int first_n = active_1 ? mfirst[team + 8] : 0, size_n = active_1 ? msize[team + 8] : 0,
          woff_n = active_1 ? moff32[team + 8] : 0;
      f32x4_a4 w[2];
      weights_of(first, size, woff, w);
      for (int m0 = 0; m0 < nb; m0 += 8) {
        const int m = m0 + team, mnn = m + 16;
        const bool active_n = m + 8 < nb, active_nn = mnn < nb;
        const int first_nn = active_nn ? mfirst[mnn] : 0, size_nn = active_nn ? msize[mnn] : 0,
                  woff_nn = active_nn ? moff32[mnn] : 0;
        f32x4_a4 w_n[2];
        weights_of(first_n, size_n, woff_n, w_n);
        const int lead = first & 3, slices = (size + lead + 31) >> 5;
        const float* __restrict__ wt = p.mel_w32 + woff + 4 * tl;
        const float* __restrict__ pb0 = ps + (first - lead) + 4 * tl;
        const int rot = team & 3;
        const float* __restrict__ pbr[4] = {pb0 + rot, pb0 + ((rot + 1) & 3), pb0 + ((rot + 2) & 3),
                                            pb0 + ((rot + 3) & 3)};
        float acc_a = 0.0f, acc_b = 0.0f;
        for (int e0 = 0;;) {
          float pva[8], pvb[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) pva[e] = pbr[e & 3][32 * (e0 + (e >> 2))];
          if (two) {
#pragma unroll
            for (int e = 0; e < 8; ++e) pvb[e] = pbr[e & 3][kSpecB + 32 * (e0 + (e >> 2))];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) pvb[e] = 0.0f;
          }
          lds_wait();
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            acc_a += w[e >> 2][e & 3] * pva[e];
            acc_b += w[e >> 2][e & 3] * pvb[e];
          }
          e0 += 2;
          if (!__any(e0 < slices)) break;
#pragma unroll
          for (int i = 0; i < 2; ++i)   // (a filter wider than two slices: its next two)
            w[i] = *reinterpret_cast<const f32x4_a4*>(e0 + i < slices ? wt + 32 * (e0 + i) : wz);
        }
        acc_a += dpp_row_ror<0xB1>(acc_a);   // quad_perm [1,0,3,2]
        acc_b += dpp_row_ror<0xB1>(acc_b);
        acc_a += dpp_row_ror<0x4E>(acc_a);   // quad_perm [2,3,0,1]
        acc_b += dpp_row_ror<0x4E>(acc_b);
        acc_a += dpp_row_ror<0x141>(acc_a);  // row_half_mirror: the other quad of the team
        acc_b += dpp_row_ror<0x141>(acc_b);
        if (active && tl == 0) {
          if (KIND == SNF_KIND_FBANK) {
            row_a[mel_col + m] = p.use_log ? fast_log(floor_eps(acc_a)) : acc_a;
            if (two) row_b[mel_col + m] = p.use_log ? fast_log(floor_eps(acc_b)) : acc_b;
          } else if (KIND == SNF_KIND_MFCC) {
            melbuf[m] = fast_log(floor_eps(acc_a));
            melbuf[128 + m] = fast_log(floor_eps(acc_b));
          } else {  // PLP: linear mel energies, the recipe continues in plp_tail_kernel
            row_a[m] = acc_a;
            if (two) row_b[m] = acc_b;
          }
        }
        active = active_n;
        first = first_n;
        size = size_n;
        woff = woff_n;
        first_n = first_nn;
        size_n = size_nn;
        woff_n = woff_nn;
        w[0] = w_n[0];
        w[1] = w_n[1];
      }
      if (KIND == SNF_KIND_FBANK && p.use_energy && lane == 0) {
        row_a[p.htk_compat ? nb : 0] = log_energy[0];
        if (two) row_b[p.htk_compat ? nb : 0] = log_energy[1];
      }
      if (KIND == SNF_KIND_MFCC) {
        wave_lds_sync();
        // DCT-II: teams of 4 lanes per cepstral coefficient, 16 coefficients per round, both frames
        const int ct = lane >> 2, cl = lane & 3;
        for (int c0 = 0; c0 < p.num_ceps; c0 += 16) {
          const int c = c0 + ct;
          const bool ca = c < p.num_ceps;
          const float* __restrict__ dm = p.dct + (ca ? c : 0) * nb;
          float va = 0.0f, vb = 0.0f;
          for (int m0 = 0; m0 < nb; m0 += 32) {  // 8 coefficients per lane in flight
            float dv[8], mva[8], mvb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int m = m0 + cl + 4 * e;
              dv[e] = dm[m < nb ? m : 0];
              mva[e] = melbuf[m < nb ? m : 0];
              mvb[e] = melbuf[128 + (m < nb ? m : 0)];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              va += (m0 + cl + 4 * e < nb) ? dv[e] * mva[e] : 0.0f;
              vb += (m0 + cl + 4 * e < nb) ? dv[e] * mvb[e] : 0.0f;
            }
          }
          va += dpp_row_ror<0xB1>(va);
          vb += dpp_row_ror<0xB1>(vb);
          va += dpp_row_ror<0x4E>(va);
          vb += dpp_row_ror<0x4E>(vb);
          if (ca && cl == 0) {
            if (p.lifter) {
              va *= p.lifter[c];
              vb *= p.lifter[c];
            }
            if (c == 0 && p.use_energy) {
              va = log_energy[0];
              vb = log_energy[1];
            }
            int oc = c;
            if (p.htk_compat) {
              oc = c == 0 ? p.num_ceps - 1 : c - 1;
              if (c == 0 && !p.use_energy) {
                va = static_cast<float>(static_cast<double>(va) * 1.4142135623730950488016887);
                vb = static_cast<float>(static_cast<double>(vb) * 1.4142135623730950488016887);
              }
            }
            row_a[oc] = va;
            if (two) row_b[oc] = vb;
          }
        }
      }
    }
    wave_lds_sync();  // the next trip reuses the buffer
    if (!split) g += stride;
  }
#undef SNF_LOAD_FRAMES
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool fbank1024x2_eligible(const MelParams& mp) {
  if (getenv("SNF_DISABLE_PAIR1024")) return false;
  if (!mp.pow2 || mp.padded != 1024) return false;
  if (mp.win_len < 65) return false;   // (lane 0 holds at least two rows: the pre-emphasis neighbour of row 1)
  if (mp.kind != SNF_KIND_FBANK && mp.kind != SNF_KIND_MFCC && mp.kind != SNF_KIND_PLP &&
      mp.kind != SNF_KIND_SPECTROGRAM)
    return false;
  if (mp.kind != SNF_KIND_SPECTROGRAM && mp.num_bins > 128) return false;
  return true;
}

// Window values and twiddles of the kernel, laid out per lane (float2 units, see kOff*)
void fbank1024x2_tables(const MelParams& mp, const std::vector<float>& window, std::vector<float>* blob) {
  constexpr double kTwoPi = 6.283185307179586476925286766559005;
  blob->assign(static_cast<size_t>(kPairTableFloat2) * 2, 0.0f);
  float* t = blob->data();
  auto put = [&](int index, double re, double im) {
    t[2 * index] = static_cast<float>(re);
    t[2 * index + 1] = static_cast<float>(im);
  };
  for (int lane = 0; lane < 64; ++lane) {
    for (int j = 0; j < 16; ++j) {
      const int n = lane + 64 * j;
      const double w = n < mp.win_len ? window[n] : 0.0;
      put(kOffWin + lane * 18 + j, w, w);
      const double a1 = -kTwoPi * ((lane * j) % 1024) / 1024.0;  // W1024^(lane k1), k1 = j
      put(kOffTw1 + lane * 18 + j, std::cos(a1), std::sin(a1));
    }
  }
  for (int bq = 0; bq < 4; ++bq)
    for (int i = 0; i < 4; ++i)
      for (int c = 0; c < 4; ++c) {
        const double a = -kTwoPi * (((bq + 4 * i) * c) % 64) / 64.0;
        put(kOffTw2 + bq * 18 + i * 4 + c, std::cos(a), std::sin(a));
      }
}

int launch_fbank1024x2(const MelParams& p, const BatchArgs& b, const float* tables, float* out, int out_cols,
                       double* energy_out, hipStream_t stream) {
  if (b.n_pairs <= 0) return SNF_OK;
  const int lds = kPairTableBytes + kPairWaves * kPairBufBytes;
  int64_t blocks = (b.n_pairs + kPairWaves - 1) / kPairWaves;
  if (blocks > 256) blocks = 256;  // one persistent workgroup per CU, grid-stride over the pairs
  const int rows = (p.win_len + 63) / 64;
  static const float split_ratio = [] {
    const char* e = getenv("SNF_PAIR_SPLIT_RATIO");   // (experiments: 0 splits every pair, inf none)
    return e ? static_cast<float>(atof(e)) : kSplitRatio;
  }();
#define SNF_PAIR4(NJ_, KIND_, DI_, SN_)                                                                          \
  do {                                                                                                           \
    SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fbank1024x2_kernel<NJ_, KIND_, DI_, SN_>),  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds));                        \
    hipLaunchKernelGGL((fbank1024x2_kernel<NJ_, KIND_, DI_, SN_>), dim3(static_cast<unsigned>(blocks)),         \
                       dim3(kPairWaves * 64), lds, stream, p, b, reinterpret_cast<const float2*>(tables),        \
                       split_ratio, out, out_cols, energy_out);                                                  \
  } while (0)
#define SNF_PAIR3(NJ_, KIND_, DI_)                                                                        \
  do {                                                                                                    \
    if (p.snip_edges) SNF_PAIR4(NJ_, KIND_, DI_, true);                                                   \
    else SNF_PAIR4(NJ_, KIND_, DI_, false);                                                               \
  } while (0)
#define SNF_PAIR2(NJ_, KIND_)                                                                             \
  do {                                                                                                    \
    if (p.dither != 0.0f) SNF_PAIR3(NJ_, KIND_, true);                                                    \
    else SNF_PAIR3(NJ_, KIND_, false);                                                                    \
  } while (0)
#define SNF_PAIR(NJ_)                                                                                     \
  do {                                                                                                    \
    if (p.kind == SNF_KIND_FBANK) SNF_PAIR2(NJ_, SNF_KIND_FBANK);                                         \
    else if (p.kind == SNF_KIND_MFCC) SNF_PAIR2(NJ_, SNF_KIND_MFCC);                                      \
    else if (p.kind == SNF_KIND_PLP) SNF_PAIR2(NJ_, SNF_KIND_PLP);                                        \
    else SNF_PAIR2(NJ_, SNF_KIND_SPECTROGRAM);                                                            \
  } while (0)
  if (rows <= 9) SNF_PAIR(9);
  else if (rows <= 13) SNF_PAIR(13);
  else SNF_PAIR(16);
#undef SNF_PAIR2
#undef SNF_PAIR3
#undef SNF_PAIR4
#undef SNF_PAIR
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

}  // namespace snf
