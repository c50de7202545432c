// Register-resident fused spectrogram / filterbank / MFCC / PLP-mel kernel for every configuration
// whose frames pad to 512 samples (16-32 ms windows at 16 kHz; the 25 ms headline shape has its own
// instantiation), both snip_edges modes, dither, per-utterance VTLN warps, on gfx950.
//
// Mapping (wave64 = 4 frames x 16 lanes; lane l of a frame, complex packing z[n] = x[2n] + i x[2n+1]):
//   A  load: lane l reads samples of z[l + 16 j], j < NJ straight from HBM/L2 (int16 pairs, one dword
//      per element + the left neighbour for pre-emphasis); DC removal via a 16-lane DPP all-reduce;
//      pre-emphasis and window in registers (Kaldi op order).
//   B  pass 1: 16-point FFT over j in registers (radix-4 x radix-4, compile-time twiddles), inter-pass
//      twiddle W256^(l k2), 16x16 transpose through a padded (conflict-free) wave-private LDS tile.
//   C  pass 2: 16-point FFT over n1 in registers -> lane l holds Z[l + 16 k1].
//   D  real-FFT unpack + power: bins k and 256-k are paired; the partner Z[256-k] comes through LDS
//      (half a tile), twiddles W512^k from an LDS table.  Scaled by 4 (the 1/2 factors of the unpack
//      are folded into the mel weights as an exact power of two).
//   E  power spectrum to a wave-private LDS tile; F: sparse mel filterbank (one (round, lane) slot per
//      bin, the widest bins split over two neighbouring lanes), log, row segments to HBM.  MFCC adds
//      the 13x23 DCT-II and lifter from LDS tables.
// Nothing but the int16 samples and the float32 features ever touches HBM; no workgroup barrier.
//
// The dense contractions (mel x frame, DCT-II) are NOT mapped to MFMA here: the mel matrix is 95 %
// zeros (2 non-zeros per FFT bin), f32 MFMA peaks at 1.4x the measured f32 VALU rate, and a 16-frame MFMA
// tile would cost more LDS traffic than the sparse form saves in VALU.
//
// Restates the same [KALDI-UPSTREAM] per-frame recipe as kernels_mel.hip (feature-window.cc
// ProcessWindow order, feature-fbank.cc, feature-mfcc.cc, MelBanks::Compute), reached by the
// reference at shennong/processor/base.py:429-431.
#include <float.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "snf_internal.h"

namespace snf {

namespace {

// ln(x) for x >= FLT_EPSILON via the hardware log2 (1 ulp): 2 instructions instead of ~15
__device__ __forceinline__ float fast_log(float x) {
  return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
}

// On gfx950 ds_read2_b64 runs at half the bandwidth of ds_read_b64 / ds_read_b128 (MI355X_MICROARCH
// LDS table), and hipcc merges adjacent 8-byte LDS loads into it.  The helpers below issue single reads
// through inline asm.  A batch of reads AND the s_waitcnt that completes them form ONE asm statement:
// the compiler treats an asm output as valid the moment the statement ends, so with the wait in a
// later statement it is free to copy (v_mov) an output register before its data has landed - the
// upper lanes of a wave are served last by the LDS pipe, which made exactly the fourth frame of a
// wave read stale values on boxes where the timing lined up.  Outputs are early-clobber: the address
// register is still needed by the later reads of the batch.
typedef __attribute__((address_space(3))) const void* lds_cptr;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_cptr)p));
}
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// dst[i] = the float2 at byte offset 8 i from `base`, i < 16 (one row of the transpose tile)
__device__ __forceinline__ void read16_b64(const void* base, float2 (&d)[16]) {
  asm volatile(
      "ds_read_b64 %0, %16\n ds_read_b64 %1, %16 offset:8\n ds_read_b64 %2, %16 offset:16\n"
      "ds_read_b64 %3, %16 offset:24\n ds_read_b64 %4, %16 offset:32\n ds_read_b64 %5, %16 offset:40\n"
      "ds_read_b64 %6, %16 offset:48\n ds_read_b64 %7, %16 offset:56\n ds_read_b64 %8, %16 offset:64\n"
      "ds_read_b64 %9, %16 offset:72\n ds_read_b64 %10, %16 offset:80\n ds_read_b64 %11, %16 offset:88\n"
      "ds_read_b64 %12, %16 offset:96\n ds_read_b64 %13, %16 offset:104\n ds_read_b64 %14, %16 offset:112\n"
      "ds_read_b64 %15, %16 offset:120\n s_waitcnt lgkmcnt(0)"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]),
        "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11]), "=&v"(d[12]), "=&v"(d[13]),
        "=&v"(d[14]), "=&v"(d[15])
      : "v"(lds_addr(base))
      : "memory");
}
// dst[i] = the float2 at byte offset 128 (7 - i) from `base`, i < 8 (partner rows, reversed)
__device__ __forceinline__ void read8_b64_rev128(const void* base, float2 (&d)[8]) {
  asm volatile(
      "ds_read_b64 %0, %8 offset:896\n ds_read_b64 %1, %8 offset:768\n ds_read_b64 %2, %8 offset:640\n"
      "ds_read_b64 %3, %8 offset:512\n ds_read_b64 %4, %8 offset:384\n ds_read_b64 %5, %8 offset:256\n"
      "ds_read_b64 %6, %8 offset:128\n ds_read_b64 %7, %8\n s_waitcnt lgkmcnt(0)"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]),
        "=&v"(d[7])
      : "v"(lds_addr(base))
      : "memory");
}
// N float4 (= 2 N complex) contiguous from `base`: plain 16-byte LDS loads (the compiler emits
// ds_read_b128 - there is no slower merged form for 128-bit reads - and places the waits itself)
template <int N>
__device__ __forceinline__ void read_quads(const void* base, float4 (&dst)[N]) {
  const float4* __restrict__ q = reinterpret_cast<const float4*>(base);
#pragma unroll
  for (int i = 0; i < N; ++i) dst[i] = q[i];
}
// acc += sum over `ngroups` (a multiple of 2) groups of 4 taps of w * x; weights [group][16 lanes]
// float4 at `wbase`, data contiguous at `pbase`.  Four groups (8 x ds_read_b128) or two are in flight
// per wait; reads and wait are one asm statement (see above).
__device__ __forceinline__ void fma4(const float4& w, const float4& x, float& acc) {
  acc += w.x * x.x;
  acc += w.y * x.y;
  acc += w.z * x.z;
  acc += w.w * x.w;
}
template <int I>
__device__ __forceinline__ void read_groups4(const void* wbase, const void* pbase, float4 (&w)[4],
                                             float4 (&x)[4]) {
  asm volatile(
      "ds_read_b128 %0, %8 offset:%10\n ds_read_b128 %4, %9 offset:%14\n"
      "ds_read_b128 %1, %8 offset:%11\n ds_read_b128 %5, %9 offset:%15\n"
      "ds_read_b128 %2, %8 offset:%12\n ds_read_b128 %6, %9 offset:%16\n"
      "ds_read_b128 %3, %8 offset:%13\n ds_read_b128 %7, %9 offset:%17\n s_waitcnt lgkmcnt(0)"
      : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]),
        "=&v"(x[3])
      : "v"(lds_addr(wbase)), "v"(lds_addr(pbase)), "n"((I + 0) * 256), "n"((I + 1) * 256),
        "n"((I + 2) * 256), "n"((I + 3) * 256), "n"((I + 0) * 16), "n"((I + 1) * 16), "n"((I + 2) * 16),
        "n"((I + 3) * 16)
      : "memory");
}
template <int I>
__device__ __forceinline__ void read_groups2(const void* wbase, const void* pbase, float4 (&w)[2],
                                             float4 (&x)[2]) {
  asm volatile(
      "ds_read_b128 %0, %4 offset:%6\n ds_read_b128 %2, %5 offset:%8\n"
      "ds_read_b128 %1, %4 offset:%7\n ds_read_b128 %3, %5 offset:%9\n s_waitcnt lgkmcnt(0)"
      : "=&v"(w[0]), "=&v"(w[1]), "=&v"(x[0]), "=&v"(x[1])
      : "v"(lds_addr(wbase)), "v"(lds_addr(pbase)), "n"((I + 0) * 256), "n"((I + 1) * 256),
        "n"((I + 0) * 16), "n"((I + 1) * 16)
      : "memory");
}
template <int G, int I = 0>
__device__ __forceinline__ void mel_groups(const void* wbase, const void* pbase, int ngroups,
                                           float& acc) {
  if constexpr (I < G) {
    if (I + 4 <= ngroups) {  // wave-uniform
      float4 w[4], x[4];
      read_groups4<I>(wbase, pbase, w, x);
#pragma unroll
      for (int i = 0; i < 4; ++i) fma4(w[i], x[i], acc);
      mel_groups<G, I + 4>(wbase, pbase, ngroups, acc);
    } else if (I + 2 <= ngroups) {
      float4 w[2], x[2];
      read_groups2<I>(wbase, pbase, w, x);
      fma4(w[0], x[0], acc);
      fma4(w[1], x[1], acc);
    }
  }
}

// counter-based N(0,1) pair for Kaldi's per-frame dither (statistical stand-in for RandGauss(), which
// draws from C rand() and is not reproducible): murmur-style 32-bit finalisers + Box-Muller on the
// hardware log2 / sqrt / sin / cos (v_sin_f32 and v_cos_f32 take revolutions)
__device__ __forceinline__ unsigned fmix32(unsigned h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ float2 gauss_pair(unsigned key_lo, unsigned key_hi, unsigned n) {
  const unsigned h1 = fmix32(key_lo + n * 0x9E3779B1u);
  const unsigned h2 = fmix32(key_hi ^ h1);
  const float u1 = (static_cast<float>(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
  const float u2 = static_cast<float>(h2 >> 8) * (1.0f / 16777216.0f);           // [0, 1)
  const float r = __builtin_amdgcn_sqrtf(-1.38629436111989f * __builtin_amdgcn_logf(u1));
  return make_float2(r * __builtin_amdgcn_cosf(u2), r * __builtin_amdgcn_sinf(u2));
}

constexpr int kMaxGroups = kFast512MaxGroups;  // 4-tap groups per mel round

constexpr int kMaxWaves = 16;             // wavefronts per workgroup: 8 when two workgroups fit the LDS
                                          // of a CU (measured 5 % faster), else one of 16
constexpr int kTileRow = 17;               // complex per transposed row (16 + 1 pad: conflict-free)
constexpr int kFrameTileBytes = 16 * kTileRow * 8;  // wave-private LDS per frame (2176 B)
constexpr int kMaxRounds = kFast512MaxRounds;  // mel bins <= 64
constexpr int kFastHeaderFloats = 16;          // table header: the mel layout of this warp factor
constexpr int kSetsPerBlock = 64;              // PERUTT: frame sets (of 4 frames) per workgroup

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// sum over the 16 lanes of a DPP row (= one frame), result in every lane of the row
template <int CTRL>
__device__ __forceinline__ float dpp_row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                                               0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_row_ror_d(double v) {
  const long long bits = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(bits), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(bits >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) |
                                        static_cast<long long>(static_cast<unsigned>(lo)));
}
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_row_ror<0x128>(v);  // row_ror:8
  v += dpp_row_ror<0x124>(v);  // row_ror:4
  v += dpp_row_ror<0x122>(v);  // row_ror:2
  v += dpp_row_ror<0x121>(v);  // row_ror:1
  return v;
}

__device__ __forceinline__ int64_t find_utt(const int64_t* __restrict__ offsets, int64_t n,
                                            int64_t g) {
  int64_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * (-i)
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// 4-point forward DFT
__device__ __forceinline__ void dft4(float2 a0, float2 a1, float2 a2, float2 a3, float2& o0,
                                     float2& o1, float2& o2, float2& o3) {
  const float2 s0 = cadd(a0, a2), s1 = csub(a0, a2), s2 = cadd(a1, a3), s3 = mul_mi(csub(a1, a3));
  o0 = cadd(s0, s2);
  o1 = cadd(s1, s3);
  o2 = csub(s0, s2);
  o3 = csub(s1, s3);
}

// 16-point forward FFT in registers, natural order in and out (radix-4 DIF x radix-4)
__device__ __forceinline__ void fft16(float2 (&v)[16]) {
  constexpr float c1 = 0.92387953251128675613f;  // cos(pi/8)
  constexpr float s1 = 0.38268343236508977173f;  // sin(pi/8)
  constexpr float r2 = 0.70710678118654752440f;  // sqrt(1/2)
  float2 t[4][4];  // t[m][q]
#pragma unroll
  for (int q = 0; q < 4; ++q) dft4(v[q], v[q + 4], v[q + 8], v[q + 12], t[0][q], t[1][q], t[2][q], t[3][q]);
  // twiddles W16^(q m)
  t[1][1] = cmul(t[1][1], make_float2(c1, -s1));                                   // W^1
  t[1][2] = make_float2((t[1][2].x + t[1][2].y) * r2, (t[1][2].y - t[1][2].x) * r2);  // W^2
  t[1][3] = cmul(t[1][3], make_float2(s1, -c1));                                   // W^3
  t[2][1] = make_float2((t[2][1].x + t[2][1].y) * r2, (t[2][1].y - t[2][1].x) * r2);  // W^2
  t[2][2] = mul_mi(t[2][2]);                                                       // W^4
  t[2][3] = make_float2((t[2][3].y - t[2][3].x) * r2, -(t[2][3].x + t[2][3].y) * r2); // W^6
  t[3][1] = cmul(t[3][1], make_float2(s1, -c1));                                   // W^3
  t[3][2] = make_float2((t[3][2].y - t[3][2].x) * r2, -(t[3][2].x + t[3][2].y) * r2); // W^6
  t[3][3] = cmul(t[3][3], make_float2(-c1, s1));                                   // W^9
#pragma unroll
  for (int m = 0; m < 4; ++m) dft4(t[m][0], t[m][1], t[m][2], t[m][3], v[m], v[4 + m], v[8 + m], v[12 + m]);
}

}  // namespace

// ENERGY: 0 = no log-energy column, 1 = raw (before pre-emphasis/window), 2 = after the window
template <int NJ, int KIND, int ENERGY, bool DITHER, bool SNIP, bool PERUTT>
__global__ __launch_bounds__(kMaxWaves * 64, 4) void fbank512_kernel(const Fast512Params p,
                                                                   const BatchArgs b,
                                                                   float* __restrict__ out,
                                                                   double* __restrict__ energy_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem);
  // PERUTT (utterances with VTLN warps): a workgroup works on a run of frame sets of ONE utterance
  // and stages the tables of that utterance's warp factor; its mel layout comes from the table header
  // instead of the kernel arguments.
  int64_t pu_u = 0, pu_f0 = 0, pu_T = 0, pu_s0 = 0, pu_n = 0;
  int pu_set0 = 0;
  const float* __restrict__ gtab = p.tables;
  int h_rounds = p.rounds, h_off_first = p.off_first, h_off_w = p.off_w, h_off_dct = p.off_dct,
      h_off_lifter = p.off_lifter, h_table_floats = p.table_floats;
  int h_maxcount[kMaxRounds], h_woff[kMaxRounds];
#pragma unroll
  for (int r = 0; r < kMaxRounds; ++r) {
    h_maxcount[r] = p.mel_maxcount[r];
    h_woff[r] = p.mel_woff[r];
  }
  if (PERUTT) {
    pu_u = b.blk_utt[blockIdx.x];
    pu_set0 = b.blk_set0[blockIdx.x];
    pu_f0 = b.frame_offsets[pu_u];
    pu_T = b.frame_offsets[pu_u + 1] - pu_f0;
    pu_s0 = b.sample_offsets[pu_u];
    pu_n = b.sample_offsets[pu_u + 1] - pu_s0;
    gtab = p.tables + static_cast<int64_t>(b.utt_warp ? b.utt_warp[pu_u] : 0) * p.table_stride;
    const int* __restrict__ hdr = reinterpret_cast<const int*>(gtab);
    h_rounds = hdr[0];
#pragma unroll
    for (int r = 0; r < kMaxRounds; ++r) {
      h_maxcount[r] = hdr[1 + r];
      h_woff[r] = hdr[5 + r];
    }
    h_off_first = hdr[9];
    h_off_w = hdr[10];
    h_off_dct = hdr[11];
    h_off_lifter = hdr[12];
    h_table_floats = hdr[13];
  }
  // ---- stage the tables into LDS (the only workgroup-wide barrier of the kernel) -------------------
  for (int i = threadIdx.x; i < h_table_floats; i += blockDim.x) tab[i] = gtab[i];
  __syncthreads();
  // lane-major tables (row = one lane's values, padded so that the 16 lanes of a frame hit 64
  // distinct banks with ds_read_b128): window pairs [16][16 + 2], inter-pass twiddles [16][16 + 2],
  // unpack twiddles [16][8 + 2] complex
  const float2* __restrict__ t_win = reinterpret_cast<const float2*>(tab + kFastHeaderFloats);
  const float2* __restrict__ t_tw16 = t_win + 16 * 18;
  const float2* __restrict__ t_tw512 = t_tw16 + 16 * 18;
  // per mel slot (round, lane): first tap (multiple of 4), output bin (-1: none), split flag
  const int* __restrict__ t_first = reinterpret_cast<const int*>(tab + h_off_first);
  const int* __restrict__ t_bin = t_first + 16 * kMaxRounds;
  const int* __restrict__ t_pair = t_bin + 16 * kMaxRounds;
  const float* __restrict__ t_w = tab + h_off_w;
  const float* __restrict__ t_dct = tab + h_off_dct;
  const float* __restrict__ t_lifter = tab + h_off_lifter;

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l = lane & 15, q = lane >> 4;
  const int tab_bytes = ((PERUTT ? p.table_stride : p.table_floats) * 4 + 255) & ~255;
  char* wave_base = smem + tab_bytes + (wid * 4 + q) * kFrameTileBytes;
  float2* tile = reinterpret_cast<float2*>(wave_base);          // 16 rows x 17 complex
  // power tile aliases the frame tile: 257 floats.  The frame tiles are 544 floats apart (bank offset
  // 0, 32, 0, 32): a skew of 16 q floats puts the 16-lane runs of the four frames of a 4-byte access on
  // four disjoint bank windows
  float* ptile = reinterpret_cast<float*>(wave_base) + q * 16;
  // The padding column of the frame tile (complex index 17 k + 16) is never written by the
  // transposes, but the mel phase reads a few floats past the end of the power tile with ZERO
  // weights (group rounding of the top bin), and LDS keeps whatever the previous workgroup or kernel
  // left there: a NaN bit pattern (e.g. the -1 entries of another plan's slot table) made
  // 0 * NaN = NaN out of the last mel bin of the fourth frame of a wave.  Zero it once.
  tile[l * kTileRow + 16] = make_float2(0.0f, 0.0f);

  const int n_waves = blockDim.x >> 6;
  // flat mode: sets of 4 consecutive global frames, grid-stride.  PERUTT: sets of 4 consecutive frames
  // of the workgroup's utterance, kSetsPerBlock of them per workgroup.
  int64_t n_sets = (b.total_frames + 3) >> 2;
  int64_t set_stride = static_cast<int64_t>(gridDim.x) * n_waves;
  if (PERUTT) {
    const int64_t utt_sets = (pu_T + 3) >> 2;
    n_sets = pu_set0 + kSetsPerBlock < utt_sets ? pu_set0 + kSetsPerBlock : utt_sets;
    set_stride = n_waves;
  }
  typedef int __attribute__((aligned(2))) int_a2;
  const int64_t last_frame = PERUTT ? pu_T - 1 : b.total_frames - 1;  // (local index when PERUTT)
  // first sample / edge mark of (local) frame index gi, clamped to the last frame
  auto start_of = [&](int64_t gi) -> int64_t {
    const int64_t gc = gi < last_frame ? gi : last_frame;
    if (!PERUTT) return b.frame_start[gc];
    if (SNIP) return pu_s0 + gc * p.win_shift;
    int64_t rel = gc * p.win_shift + p.win_shift / 2 - p.win_len / 2;
    if (rel < 0) rel = 0;
    if (rel + p.win_len > pu_n) rel = pu_n - p.win_len;
    return pu_s0 + rel;
  };
  auto edge_of = [&](int64_t gi) -> int {
    const int64_t gc = gi < last_frame ? gi : last_frame;
    if (!PERUTT) return b.frame_edge[gc];
    const int64_t rel = gc * p.win_shift + p.win_shift / 2 - p.win_len / 2;
    return (rel < 0 || rel + p.win_len > pu_n) ? static_cast<int>(pu_u + 1) : 0;
  };
  // Software pipeline over frame sets: the samples of set i+1 and the start offset of set i+2 are
  // requested while set i is being transformed, so no global-memory latency sits on the critical
  // path of a wave.  frame_start[g] (sample index of the first sample of global frame g) is built
  // once per offsets table by build_frame_start_kernel.
  int64_t set = PERUTT ? pu_set0 + wid : static_cast<int64_t>(blockIdx.x) * n_waves + wid;
  // NJ = 13 is the exact shape of the 25 ms / 16 kHz window (only element j = 12 can fall outside the
  // window); NJ = 16 covers every other window length that pads to 512 samples with a per-element test
  const bool in_last = 2 * (l + 16 * (NJ - 1)) < p.win_len;
  auto in_window = [&](int j) -> bool {
    if (NJ == 13) return j < NJ - 1 || in_last;
    return 2 * (l + 16 * j) < p.win_len;
  };
  int raw[NJ];
  int64_t start_next = 0;
  int edge_next = 0;  // snip_edges = false: 0 for an interior frame, utterance + 1 for a frame that
                      // reaches outside its utterance (reloaded with Kaldi's reflection)
  if (set < n_sets) {
    const int64_t g = set * 4 + q;
    const int16_t* __restrict__ wp = b.wave + start_of(g);
    const int16_t* __restrict__ wl = wp + 2 * l;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      raw[j] = *reinterpret_cast<const int_a2*>((NJ == 13 && j < NJ - 1) || in_window(j) ? wl + 32 * j : wp);
    const int64_t gn = (set + set_stride) * 4 + q;
    start_next = start_of(gn);
    if (!SNIP) edge_next = edge_of(g);
  }
  for (; set < n_sets; set += set_stride) {
    const int64_t gl = set * 4 + q;               // frame index (inside the utterance when PERUTT)
    const bool valid = gl <= last_frame;
    const int64_t g = PERUTT ? pu_f0 + gl : gl;   // global output row
    const int edge_cur = edge_next;

    // ---- A: DC removal, pre-emphasis, window ------------------------------------------------------
    // A1: one dword (two int16 samples) per element, requested one iteration ago.  Only the last j
    // can fall outside the window (NJ = ceil(win_len / 32)).
    float4 win4[(NJ + 1) / 2];
    read_quads<(NJ + 1) / 2>(t_win + l * 18, win4);
    unsigned dkey_lo = 0, dkey_hi = 0;
    if (DITHER) {
      const unsigned long long k = (static_cast<unsigned long long>(g) + 1) * 0x9E3779B97F4A7C15ull ^ p.seed;
      dkey_lo = fmix32(static_cast<unsigned>(k));
      dkey_hi = fmix32(static_cast<unsigned>(k >> 32) ^ dkey_lo);
    }
    float xe[NJ], xo[NJ];
    float part = 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      xe[j] = static_cast<float>(static_cast<short>(raw[j] & 0xffff));
      xo[j] = static_cast<float>(raw[j] >> 16);
    }
    if (!SNIP && edge_cur != 0 && valid) {
      // [KALDI-UPSTREAM] ExtractWindow, snip_edges = false: samples outside the utterance are
      // reflected (-k - 1 below the start, 2 n - 1 - k beyond the end).  Only the first and last
      // frames of an utterance take this path; their prefetched samples came from a clamped window.
      const int64_t u = edge_cur - 1;
      const int64_t s0 = b.sample_offsets[u], n = b.sample_offsets[u + 1] - s0;
      const int64_t rel = (g - b.frame_offsets[u]) * p.win_shift + p.win_shift / 2 - p.win_len / 2;
      const int16_t* __restrict__ w0 = b.wave + s0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (in_window(j)) {
          int64_t k0 = rel + 2 * (l + 16 * j), k1 = k0 + 1;
          while (k0 < 0 || k0 >= n) k0 = k0 < 0 ? -k0 - 1 : 2 * n - 1 - k0;
          while (k1 < 0 || k1 >= n) k1 = k1 < 0 ? -k1 - 1 : 2 * n - 1 - k1;
          xe[j] = static_cast<float>(w0[k0]);
          xo[j] = static_cast<float>(w0[k1]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (DITHER) {  // Kaldi dithers before the DC removal
        const float2 nz = gauss_pair(dkey_lo, dkey_hi, static_cast<unsigned>(l + 16 * j));
        xe[j] += p.dither * nz.x;
        xo[j] += p.dither * nz.y;
      }
      const float s2 = xe[j] + xo[j];
      part += in_window(j) ? s2 : 0.0f;
    }
    // prefetch: samples of the next set (its start offset arrived during the previous iteration),
    // start offset of the set after it
    if (set + set_stride < n_sets) {
      const int16_t* __restrict__ wp = b.wave + start_next;
      const int16_t* __restrict__ wl = wp + 2 * l;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        raw[j] = *reinterpret_cast<const int_a2*>((NJ == 13 && j < NJ - 1) || in_window(j) ? wl + 32 * j : wp);
      const int64_t gn = (set + 2 * set_stride) * 4 + q;
      start_next = start_of(gn);
      if (!SNIP) edge_next = edge_of((set + set_stride) * 4 + q);
    }
    float neg_mean = 0.0f;
    if (p.remove_dc) neg_mean = -row_sum16(part) / static_cast<float>(p.win_len);
    lds_wait();
    // A2: the left neighbour x[2n-1] is the odd sample of element n-1 = lane l-1 (same j), or lane 15
    // of j-1 for lane 0: one DPP row rotate of the mean-removed value per element
    float2 z[16];
    float e_raw = 0.0f, e_post = 0.0f;
    float rot_prev = xe[0] + neg_mean;  // lane 0, j = 0: x[-1] := x[0] (Kaldi Preemphasize)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < NJ) {
        const bool in = in_window(j);
        const float ae = xe[j] + neg_mean, ao = xo[j] + neg_mean;
        const float rot = dpp_row_ror<0x121>(ao);  // lane l <- lane (l - 1) mod 16 of its frame
        const float ap = l == 0 ? rot_prev : rot;
        rot_prev = rot;
        const float2 w = (j & 1) ? make_float2(win4[j >> 1].z, win4[j >> 1].w)
                                 : make_float2(win4[j >> 1].x, win4[j >> 1].y);  // zero outside the window
        if (ENERGY == 1 && in) e_raw += ae * ae + ao * ao;
        const float ye = (ae - p.preemph * ap) * w.x;
        const float yo = (ao - p.preemph * ae) * w.y;
        z[j] = make_float2(ye, yo);
        if (ENERGY == 2) e_post += ye * ye + yo * yo;
      } else {
        z[j] = make_float2(0.0f, 0.0f);
      }
    }
    if (KIND == SNF_KIND_ENERGY) {
      // EnergyProcessor (reference processor/energy.py:173-183): float64 sum of squares of the
      // processed float32 window, floored at the smallest double, then compressed; nothing else of
      // the pipeline below is needed
      double de = 0.0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const double a = static_cast<double>(z[j].x), c = static_cast<double>(z[j].y);
        de += a * a + c * c;
      }
      de += dpp_row_ror_d<0x128>(de);
      de += dpp_row_ror_d<0x124>(de);
      de += dpp_row_ror_d<0x122>(de);
      de += dpp_row_ror_d<0x121>(de);
      de = fmax(de, DBL_MIN);
      double v = de;
      if (p.compression == SNF_COMPRESS_LOG) v = log(de);
      else if (p.compression == SNF_COMPRESS_SQRT) v = sqrt(de);
      if (valid && l == 0) out[g * static_cast<int64_t>(p.out_cols)] = static_cast<float>(v);
      continue;
    }
    float e_lin = 0.0f;
    if (ENERGY != 0) e_lin = row_sum16(ENERGY == 1 ? e_raw : e_post);
    __builtin_amdgcn_sched_barrier(0);

    // ---- B: pass 1 (FFT over j), inter-pass twiddle, transpose -------------------------------------
    float4 tw4[8];
    read_quads<8>(t_tw16 + l * 18, tw4);  // W256^(l k2), lands while the butterflies run
    fft16(z);
    lds_wait();
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2)
      z[k2] = cmul(z[k2], (k2 & 1) ? make_float2(tw4[k2 >> 1].z, tw4[k2 >> 1].w)
                                   : make_float2(tw4[k2 >> 1].x, tw4[k2 >> 1].y));
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) tile[k2 * kTileRow + l] = z[k2];
    wave_lds_sync();
    read16_b64(tile + l * kTileRow, z);
    __builtin_amdgcn_sched_barrier(0);
    // ---- C: pass 2 (FFT over n1): z[k1] = Z[l + 16 k1] ---------------------------------------------
    fft16(z);
    __builtin_amdgcn_sched_barrier(0);
    wave_lds_sync();

    // ---- D: real-FFT unpack + power (x4) -----------------------------------------------------------
    // upper half to LDS: xbuf[r][c] = Z[c + 16 (r + 8)], rows 0..7 (+ row 8 scratch for lane 0)
#pragma unroll
    for (int r = 0; r < 8; ++r) tile[r * 16 + l] = z[r + 8];
    wave_lds_sync();
    const float2* __restrict__ partner = tile + (16 - l);  // Z[256 - k]: row 7-k1, column 16-l
    float2 zpart[8];
    float4 w512q[4];
    read_quads<4>(t_tw512 + l * 10, w512q);     // W512^(l + 16 k1)
    read8_b64_rev128(partner, zpart);           // zpart[k1] = Z[256 - l - 16 k1]
    __builtin_amdgcn_sched_barrier(0);
    float pk[8], pm[8];  // 4 P[k], 4 P[256-k] for k = l + 16 k1
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
      const float2 zk = z[k1];
      const float2 zp = zpart[k1];
      const float2 w = (k1 & 1) ? make_float2(w512q[k1 >> 1].z, w512q[k1 >> 1].w)
                                : make_float2(w512q[k1 >> 1].x, w512q[k1 >> 1].y);
      const float c_re = zk.x + zp.x, c_im = zk.y - zp.y;
      const float d_re = zk.y + zp.y, d_im = zp.x - zk.x;
      const float t_re = d_re * w.x - d_im * w.y, t_im = d_re * w.y + d_im * w.x;
      const float a_re = c_re + t_re, a_im = c_im + t_im;
      const float b_re = c_re - t_re, b_im = t_im - c_im;
      pk[k1] = a_re * a_re + a_im * a_im;
      pm[k1] = b_re * b_re + b_im * b_im;
    }
    if (l == 0) {
      // k = 0: DC (and Nyquist, unused by the mel banks); pairs (16 k1, 256 - 16 k1) were computed
      // above with zp = Z[256 - 16 k1] = row (8 - k1); k1 = 0 read scratch -> overwrite
      const float dc = z[0].x + z[0].y;
      pk[0] = 4.0f * dc * dc;
      const float ny = z[0].x - z[0].y;
      pm[0] = 4.0f * ny * ny;
    }
    // k = 128 (self-paired): Z[128] sits in lane 0, register 8
    const float p128 = 4.0f * (z[8].x * z[8].x + z[8].y * z[8].y);
    wave_lds_sync();
    // ---- E: power tile ------------------------------------------------------------------------------
    float* __restrict__ pmirror = ptile + (144 - l);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
      ptile[l + 16 * k1] = pk[k1];
      pmirror[16 * (7 - k1)] = pm[k1];  // index 256 - l - 16 k1
    }
    if (l == 0) ptile[128] = p128;
    wave_lds_sync();

    // ---- log-energy column ---------------------------------------------------------------------------
    float log_energy = 0.0f;
    if (ENERGY != 0) {
      if (KIND == SNF_KIND_PLP) {
        if (valid && l == 0) energy_out[g] = log(fmax(static_cast<double>(e_lin), DBL_EPSILON));
      } else {
        log_energy = logf(fmaxf(e_lin, FLT_EPSILON));
        if (p.has_floor && log_energy < p.log_energy_floor) log_energy = p.log_energy_floor;
      }
    }

    // ---- F: sparse mel filterbank, log, epilogue ------------------------------------------------------
    float* __restrict__ row = out + g * static_cast<int64_t>(p.out_cols);
    if (KIND == SNF_KIND_SPECTROGRAM) {
      // log power spectrum, 257 bins: lane l stores bins l + 16 i (64-byte segments), bin 0 = energy
      if (valid) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float v = fast_log(fmaxf(0.25f * ptile[l + 16 * i], FLT_EPSILON));
          if (i == 0 && l == 0) v = log_energy;
          row[l + 16 * i] = v;
        }
        if (l == 0) row[256] = fast_log(fmaxf(0.25f * ptile[256], FLT_EPSILON));
      }
    }
    const int mel_col = (KIND == SNF_KIND_FBANK && p.use_energy && !p.htk_compat) ? 1 : 0;
    float logmel[kMaxRounds];
    int mbin[kMaxRounds];
#pragma unroll
    for (int r = 0; r < kMaxRounds; ++r) {
      mbin[r] = -1;
      if (r < h_rounds) {
        const int start = t_first[r * 16 + l];  // first tap rounded down to a multiple of 4
        const int m = t_bin[r * 16 + l];        // mel bin stored by this slot, or -1
        const int pair = t_pair[r * 16 + l];    // this slot and its quad neighbour share a wide bin
        // taps outside the slot's range carry zero weights and read finite filler in the tile (every
        // float of the tile is written before it is read: transposes + the zeroed padding column)
        float acc = 0.0f;
        mel_groups<kMaxGroups>(t_w + h_woff[r] + 4 * l, ptile + start, h_maxcount[r], acc);
        // wide bins are split over two neighbouring lanes of the same round (the idle slots of the
        // last round would otherwise dictate the group count of the whole round)
        const float other = dpp_row_ror<0xB1>(acc);  // quad_perm [1,0,3,2]: lane l ^ 1
        if (pair) acc += other;
        if (KIND == SNF_KIND_FBANK) {
          const float v = p.use_log ? fast_log(fmaxf(acc, FLT_EPSILON)) : acc;
          if (valid && m >= 0) row[mel_col + m] = v;
        } else if (KIND == SNF_KIND_MFCC) {
          logmel[r] = fast_log(fmaxf(acc, FLT_EPSILON));
          mbin[r] = m;
        } else {
          if (valid && m >= 0) row[m] = acc;
        }
      }
    }
    if (KIND == SNF_KIND_FBANK) {
      if (p.use_energy && valid && l == 0) row[p.htk_compat ? p.num_bins : 0] = log_energy;
    }
    if (KIND == SNF_KIND_MFCC) {
      // DCT-II + lifter: lane c owns cepstrum c (num_ceps <= 16); log-mel goes through the (now idle)
      // power tile
      wave_lds_sync();
#pragma unroll
      for (int r = 0; r < kMaxRounds; ++r)
        if (r < h_rounds && mbin[r] >= 0) ptile[mbin[r]] = logmel[r];
      wave_lds_sync();
      float v = 0.0f;
      mel_groups<16>(t_dct + 4 * l, ptile, ((p.num_bins + 7) >> 3) << 1, v);  // num_bins <= 64
      v *= t_lifter[l];
      if (l == 0 && p.use_energy) v = log_energy;
      int oc = l;
      if (p.htk_compat) {
        oc = l == 0 ? p.num_ceps - 1 : l - 1;
        if (l == 0 && !p.use_energy)
          v = static_cast<float>(static_cast<double>(v) * 1.4142135623730950488016887);
      }
      if (valid && l < p.num_ceps) row[oc] = v;
    }
    wave_lds_sync();  // the tile is reused by the next frame set
  }
}

// one thread per frame: sample index (into the concatenated wave) of the frame's first sample.
// snip_edges = false: frames are centred (start = f shift + shift / 2 - len / 2, [KALDI-UPSTREAM]
// FirstSampleOfFrame) and may reach outside the utterance; their start is clamped into the utterance
// for the bulk loads and frame_edge marks them for the reflecting reload.
__global__ void build_frame_start_kernel(const int64_t* __restrict__ frame_offsets,
                                         const int64_t* __restrict__ sample_offsets, int64_t n_utts,
                                         int64_t total_frames, int win_shift, int win_len,
                                         int snip_edges, int64_t* __restrict__ frame_start,
                                         int32_t* __restrict__ frame_edge,
                                         int32_t* __restrict__ frame_utt) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= total_frames) return;
  const int64_t u = find_utt(frame_offsets, n_utts, g);
  frame_utt[g] = static_cast<int32_t>(u);
  const int64_t s0 = sample_offsets[u], n = sample_offsets[u + 1] - s0;
  const int64_t f = g - frame_offsets[u];
  if (snip_edges) {
    frame_start[g] = s0 + f * win_shift;
    return;
  }
  const int64_t rel = f * win_shift + win_shift / 2 - win_len / 2;
  const bool edge = rel < 0 || rel + win_len > n;
  int64_t safe = rel < 0 ? 0 : rel;
  if (safe + win_len > n) safe = n - win_len;  // n >= win_len is checked by the host
  frame_start[g] = s0 + safe;
  frame_edge[g] = edge ? static_cast<int32_t>(u + 1) : 0;
}

int launch_build_frame_start(const int64_t* d_frame_offsets, const int64_t* d_sample_offsets,
                             int64_t n_utts, int64_t total_frames, int win_shift, int win_len,
                             int snip_edges, int64_t* d_frame_start, int32_t* d_frame_edge,
                             int32_t* d_frame_utt, hipStream_t stream) {
  if (total_frames <= 0) return SNF_OK;
  hipLaunchKernelGGL(build_frame_start_kernel,
                     dim3(static_cast<unsigned>((total_frames + 255) / 256)), dim3(256), 0, stream,
                     d_frame_offsets, d_sample_offsets, n_utts, total_frames, win_shift, win_len,
                     snip_edges, d_frame_start, d_frame_edge, d_frame_utt);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
bool fast512_eligible(const MelParams& mp, bool any_warp) {
  if (getenv("SNF_DISABLE_FAST512")) return false;
  if (any_warp) return false;
  // Frames that pad to 256 or 128 samples (8 kHz audio, short windows) run as the same 512-point
  // transform of the zero-extended frame: X512[s k] = X_N[k] with s = 512 / N, so the mel taps sit
  // on every s-th bin (fast512_build interleaves zero weights).  Twice the FFT arithmetic the frame
  // needs, still several times faster than the LDS radix-2 kernel.  The spectrogram needs the N/2+1
  // bins themselves and stays on the generic kernel for those sizes.
  if ((mp.padded != 512 && mp.padded != 256 && mp.padded != 128) || !mp.pow2) return false;
  if (mp.win_len & 1) return false;
  if (mp.kind != SNF_KIND_FBANK && mp.kind != SNF_KIND_MFCC && mp.kind != SNF_KIND_PLP &&
      mp.kind != SNF_KIND_SPECTROGRAM && mp.kind != SNF_KIND_ENERGY)
    return false;
  if (mp.kind == SNF_KIND_SPECTROGRAM && mp.padded != 512) return false;
  if (mp.kind == SNF_KIND_FBANK && !mp.use_power) return false;
  if (mp.num_bins > 16 * kMaxRounds) return false;
  if (mp.kind == SNF_KIND_MFCC && mp.num_ceps > 16) return false;
  return true;
}

// Builds the packed LDS table blob from the plan's host tables (warp 1.0 mel banks).
int fast512_build(const MelParams& mp, const std::vector<float>& window, const MelBanksHost& mb_in,
                  const std::vector<float>& dct, const std::vector<float>& lifter,
                  std::vector<float>* blob, Fast512Params* out) {
  constexpr double kTwoPi = 6.283185307179586476925286766559005;
  // frames shorter than 512 samples: spread the taps of every bin over the bins of the 512-point
  // spectrum of the zero-extended frame (see fast512_eligible)
  MelBanksHost spread;
  const int bin_stride = 512 / mp.padded;
  if (bin_stride > 1 && mb_in.num_bins > 0) {
    spread.num_bins = mb_in.num_bins;
    spread.num_fft_bins = mb_in.num_fft_bins * bin_stride;
    spread.center_freqs = mb_in.center_freqs;
    for (int m = 0; m < mb_in.num_bins; ++m) {
      spread.first.push_back(mb_in.first[m] * bin_stride);
      spread.size.push_back(mb_in.size[m] > 0 ? (mb_in.size[m] - 1) * bin_stride + 1 : 0);
      spread.offset.push_back(static_cast<int>(spread.w.size()));
      for (int k = 0; k < mb_in.size[m]; ++k) {
        spread.w.push_back(mb_in.w[mb_in.offset[m] + k]);
        if (k + 1 < mb_in.size[m]) spread.w.insert(spread.w.end(), bin_stride - 1, 0.0f);
      }
    }
  }
  const MelBanksHost& mb = bin_stride > 1 && mb_in.num_bins > 0 ? spread : mb_in;
  Fast512Params p{};
  p.win_len = mp.win_len;
  p.win_shift = mp.win_shift;
  p.remove_dc = mp.remove_dc;
  p.snip_edges = mp.snip_edges;
  p.preemph = mp.preemph;
  p.dither = mp.dither;
  p.seed = mp.seed;
  p.kind = mp.kind;
  p.compression = mp.compression;
  p.use_energy = mp.use_energy;
  p.need_raw = mp.need_raw;
  p.need_post = mp.need_post;
  p.htk_compat = mp.htk_compat;
  p.use_log = mp.use_log;
  p.has_floor = mp.has_floor;
  p.log_energy_floor = mp.log_energy_floor;
  p.num_bins = mp.num_bins;
  p.num_ceps = mp.num_ceps;
  p.rounds = (mp.num_bins + 15) / 16;
  blob->clear();
  blob->resize(kFastHeaderFloats, 0.0f);  // header, filled in at the end
  // window pairs, lane-major: row l = elements l + 16 j (j < 16), 2 complex of padding
  for (int l = 0; l < 16; ++l)
    for (int j = 0; j < 18; ++j) {
      const int n = l + 16 * j;
      blob->push_back(j < 16 && 2 * n < mp.win_len ? window[2 * n] : 0.0f);
      blob->push_back(j < 16 && 2 * n + 1 < mp.win_len ? window[2 * n + 1] : 0.0f);
    }
  // inter-pass twiddles, lane-major: row n1 = exp(-2 pi i n1 k2 / 256), k2 < 16
  for (int n1 = 0; n1 < 16; ++n1)
    for (int k2 = 0; k2 < 18; ++k2) {
      const double a = -kTwoPi * (n1 * (k2 < 16 ? k2 : 0)) / 256.0;
      blob->push_back(static_cast<float>(std::cos(a)));
      blob->push_back(static_cast<float>(std::sin(a)));
    }
  // unpack twiddles, lane-major: row l = exp(-2 pi i (l + 16 k1) / 512), k1 < 8
  for (int l = 0; l < 16; ++l)
    for (int k1 = 0; k1 < 10; ++k1) {
      const double a = -kTwoPi * (l + 16 * (k1 < 8 ? k1 : 0)) / 512.0;
      blob->push_back(static_cast<float>(std::cos(a)));
      blob->push_back(static_cast<float>(std::sin(a)));
    }
  // mel: every (round, lane) slot sums one run of 4-tap groups.  A bin is one slot, or - for the
  // widest bins, as many as there are idle slots - two slots in neighbouring lanes of one round whose
  // partial sums are added through DPP.  Slots are sorted by size so that the rounds are as short as
  // possible (a round costs the group count of its longest slot).
  struct Piece { int bin, start, groups, units, lo; };  // units = 2: pair (two consecutive pieces);
                                                        // lo: first tap the piece is responsible for
  std::vector<Piece> singles, pairs;  // pairs hold the first half; the second half follows it
  {
    std::vector<int> order(mb.num_bins), groups_of(mb.num_bins), start_of(mb.num_bins);
    for (int m = 0; m < mb.num_bins; ++m) {
      order[m] = m;
      start_of[m] = mb.first[m] & ~3;
      groups_of[m] = (mb.first[m] + mb.size[m] - start_of[m] + 3) / 4;
    }
    std::stable_sort(order.begin(), order.end(),
                     [&](int x, int y) { return groups_of[x] > groups_of[y]; });
    int spare = 16 * p.rounds - mb.num_bins;
    std::vector<char> split(mb.num_bins, 0);
    for (int m : order)
      if (spare > 0 && groups_of[m] >= 2) { split[m] = 1; --spare; }
    for (int m = 0; m < mb.num_bins; ++m) {
      if (split[m]) {
        const int ga = (groups_of[m] + 1) / 2;
        pairs.push_back({m, start_of[m], ga, 2, 0});
        pairs.push_back({-1, start_of[m] + 4 * ga, groups_of[m] - ga, 0, start_of[m] + 4 * ga});
      } else {
        singles.push_back({m, start_of[m], groups_of[m], 1, 0});
      }
    }
  }
  // units sorted by decreasing size; pairs (2 slots, even lane first) are placed before the singles
  // of the same round
  std::vector<std::vector<Piece>> round_slots(p.rounds);
  {
    std::vector<std::pair<int, int>> units;  // (groups, index) index < 0: pair -(idx+1), else single
    for (size_t i = 0; i < pairs.size(); i += 2) units.push_back({pairs[i].groups, -static_cast<int>(i) - 1});
    for (size_t i = 0; i < singles.size(); ++i) units.push_back({singles[i].groups, static_cast<int>(i)});
    std::stable_sort(units.begin(), units.end(),
                     [](const std::pair<int, int>& x, const std::pair<int, int>& y) { return x.first > y.first; });
    std::vector<char> used(units.size(), 0);
    for (int r = 0; r < p.rounds; ++r) {
      std::vector<Piece> pr, sg;
      int free_slots = 16;
      for (size_t u = 0; u < units.size() && free_slots > 0; ++u) {
        if (used[u]) continue;
        if (units[u].second < 0) {
          if (free_slots < 2) continue;
          const size_t i = static_cast<size_t>(-units[u].second - 1);
          pr.push_back(pairs[i]);
          pr.push_back(pairs[i + 1]);
          free_slots -= 2;
        } else {
          sg.push_back(singles[units[u].second]);
          free_slots -= 1;
        }
        used[u] = 1;
      }
      round_slots[r] = pr;
      round_slots[r].insert(round_slots[r].end(), sg.begin(), sg.end());
    }
    for (char u : used)
      if (!u) return 1;  // (cannot happen: 16 * rounds slots >= pieces)
  }
  // LDS bank conflicts of the 16-byte tap reads: two lanes of a frame collide on every group when
  // their first taps differ by a multiple of 64 (same 16-byte slot modulo the 64 banks).  A slot that
  // is shorter than its round may start up to (round length - own length) groups early (the extra
  // leading taps carry zero weights), which moves its residue: choose the shifts greedily so that
  // the 16 residues of a round are distinct; idle lanes get one of the free residues.
  std::vector<std::vector<int>> idle_start(p.rounds, std::vector<int>(16, 0));
  for (int r = 0; r < p.rounds; ++r) {
    int round_groups = 0;
    for (const Piece& pc : round_slots[r]) round_groups = pc.groups > round_groups ? pc.groups : round_groups;
    round_groups = (round_groups + 1) & ~1;
    bool taken[16] = {};
    std::vector<int> order(round_slots[r].size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int>(i);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
      return round_slots[r][x].groups > round_slots[r][y].groups;  // least slack first
    });
    for (int i : order) {
      Piece& pc = round_slots[r][i];
      const int slack = round_groups - pc.groups;
      int best = -1;
      for (int k = 0; k <= slack && best < 0; ++k)
        if (pc.start - 4 * k >= 0 && !taken[((pc.start - 4 * k) / 4) & 15]) best = k;
      if (best > 0) {
        pc.start -= 4 * best;
        pc.groups += best;
      }
      taken[(pc.start / 4) & 15] = true;
    }
    for (int l = static_cast<int>(round_slots[r].size()); l < 16; ++l) {
      int res = 0;
      while (res < 15 && taken[res]) ++res;
      taken[res] = true;
      idle_start[r][l] = 4 * res;
    }
  }
  p.off_first = static_cast<int>(blob->size());
  auto push_int = [&](int v) {
    float as_float;
    std::memcpy(&as_float, &v, 4);
    blob->push_back(as_float);
  };
  for (int table = 0; table < 3; ++table)  // first tap, output bin, pair flag: [kMaxRounds][16] each
    for (int r = 0; r < kMaxRounds; ++r)
      for (int l = 0; l < 16; ++l) {
        int v = table == 1 ? -1 : 0;
        if (table == 0 && r < p.rounds) v = idle_start[r][l];
        if (r < p.rounds && l < static_cast<int>(round_slots[r].size())) {
          const Piece& pc = round_slots[r][l];
          const bool second = pc.units == 0;
          if (table == 0) v = pc.start;
          else if (table == 1) v = pc.bin;
          else v = (pc.units == 2 || second) ? 1 : 0;
        }
        push_int(v);
      }
  for (int r = 0; r < p.rounds; ++r) {
    int groups = 0;
    for (const Piece& pc : round_slots[r]) groups = pc.groups > groups ? pc.groups : groups;
    if (groups > kMaxGroups) return 1;  // a slot is too long for the unrolled tap loop: not eligible
    p.mel_maxcount[r] = (groups + 1) & ~1;  // 4-tap groups of this round (even: read in batches)
  }
  while (blob->size() % 4) blob->push_back(0.0f);  // 16-byte alignment of the float4 weights
  p.off_w = static_cast<int>(blob->size());
  int woff = 0;
  for (int r = 0; r < p.rounds; ++r) {
    p.mel_woff[r] = woff;
    for (int g = 0; g < p.mel_maxcount[r]; ++g)
      for (int l = 0; l < 16; ++l)
        for (int i = 0; i < 4; ++i) {
          float w = 0.0f;
          if (l < static_cast<int>(round_slots[r].size())) {
            const Piece& pc = round_slots[r][l];
            // the second half of a split bin takes its bin from the slot before it
            const int m = pc.units == 0 ? round_slots[r][l - 1].bin : pc.bin;
            const int k = pc.start + 4 * g + i;  // FFT bin of this tap
            if (g < pc.groups && k >= pc.lo && k >= mb.first[m] && k < mb.first[m] + mb.size[m])
              w = 0.25f * mb.w[mb.offset[m] + k - mb.first[m]];  // exact power-of-two scaling
          }
          blob->push_back(w);
        }
    woff += p.mel_maxcount[r] * 64;
  }
  p.off_dct = static_cast<int>(blob->size());
  if (mp.kind == SNF_KIND_MFCC) {
    // [group of 4 bins][lane = cepstrum][4]
    for (int g = 0; g < ((mp.num_bins + 7) / 8) * 2; ++g)
      for (int c = 0; c < 16; ++c)
        for (int i = 0; i < 4; ++i) {
          const int m = 4 * g + i;
          blob->push_back(c < mp.num_ceps && m < mp.num_bins
                              ? dct[static_cast<size_t>(c) * mp.num_bins + m] : 0.0f);
        }
  }
  p.off_lifter = static_cast<int>(blob->size());
  for (int c = 0; c < 16; ++c)
    blob->push_back(c < static_cast<int>(lifter.size()) ? lifter[c] : 1.0f);
  p.table_floats = static_cast<int>(blob->size());
  {  // header: what the PERUTT kernel needs to know about THIS warp factor's tables
    int hdr[kFastHeaderFloats] = {};
    hdr[0] = p.rounds;
    for (int r = 0; r < kMaxRounds; ++r) {
      hdr[1 + r] = p.mel_maxcount[r];
      hdr[5 + r] = p.mel_woff[r];
    }
    hdr[9] = p.off_first;
    hdr[10] = p.off_w;
    hdr[11] = p.off_dct;
    hdr[12] = p.off_lifter;
    hdr[13] = p.table_floats;
    std::memcpy(blob->data(), hdr, sizeof(hdr));
  }
  *out = p;
  return SNF_OK;
}

int launch_fbank512(const Fast512Params& p, const BatchArgs& b, float* out, int out_cols,
                    double* energy_out, hipStream_t stream) {
  if (b.total_frames <= 0) return SNF_OK;
  Fast512Params q = p;
  q.out_cols = out_cols;
  const int tab_bytes = ((b.blk_utt ? p.table_stride : p.table_floats) * 4 + 255) & ~255;
  const bool per_utt = b.blk_utt != nullptr;
  int n_waves = 8;
  size_t lds = static_cast<size_t>(tab_bytes) + n_waves * 4 * kFrameTileBytes;
  if (2 * (lds + 512) > 160 * 1024) {  // two 8-wave workgroups do not fit: one of 16 waves
    n_waves = kMaxWaves;
    lds = static_cast<size_t>(tab_bytes) + n_waves * 4 * kFrameTileBytes;
  }
  if (lds > 160 * 1024) return set_error(SNF_E_RUNTIME, "fast512: tables do not fit in LDS");
  const int nj = (p.win_len + 31) / 32 == 13 ? 13 : 16;
  const int64_t n_sets = (b.total_frames + 3) / 4;
  int64_t blocks = (n_sets + n_waves - 1) / n_waves;
  if (per_utt) blocks = b.n_blocks;
  const int64_t max_blocks = 256 * 4 * (kMaxWaves / n_waves);  // resident workgroups x grid-stride depth 4
  if (!per_utt && blocks > max_blocks) blocks = max_blocks;
  const dim3 grid(static_cast<unsigned>(blocks)), block(n_waves * 64);
#define SNF_LAUNCH6(NJ_, KIND_, EN_, DI_, SN_, PU_)                                                  \
  do {                                                                                              \
    if (lds > 64 * 1024)                                                                            \
      SNF_HIP_CHECK(hipFuncSetAttribute(                                                            \
          reinterpret_cast<const void*>(fbank512_kernel<NJ_, KIND_, EN_, DI_, SN_, PU_>),           \
          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                      \
    hipLaunchKernelGGL((fbank512_kernel<NJ_, KIND_, EN_, DI_, SN_, PU_>), grid, block, lds, stream, \
                       q, b, out, energy_out);                                                      \
  } while (0)
#define SNF_LAUNCH5(NJ_, KIND_, EN_, DI_, SN_)                                                       \
  do {                                                                                              \
    if (per_utt && KIND_ != SNF_KIND_SPECTROGRAM && KIND_ != SNF_KIND_ENERGY)                       \
      SNF_LAUNCH6(NJ_, KIND_, EN_, DI_, SN_,                                                        \
                  (KIND_ != SNF_KIND_SPECTROGRAM && KIND_ != SNF_KIND_ENERGY));                     \
    else SNF_LAUNCH6(NJ_, KIND_, EN_, DI_, SN_, false);                                             \
  } while (0)
#define SNF_LAUNCH4(NJ_, KIND_, EN_, DI_)                                                           \
  do {                                                                                              \
    if (p.snip_edges) SNF_LAUNCH5(NJ_, KIND_, EN_, DI_, true);                                      \
    else SNF_LAUNCH5(NJ_, KIND_, EN_, DI_, false);                                                  \
  } while (0)
#define SNF_LAUNCH3(NJ_, KIND_, EN_)                                                                \
  do {                                                                                              \
    if (p.dither != 0.0f) SNF_LAUNCH4(NJ_, KIND_, EN_, true);                                       \
    else SNF_LAUNCH4(NJ_, KIND_, EN_, false);                                                       \
  } while (0)
#define SNF_LAUNCH(NJ_, KIND_)                                                                      \
  do {                                                                                              \
    if (energy == 0) SNF_LAUNCH3(NJ_, KIND_, 0);                                                    \
    else if (energy == 1) SNF_LAUNCH3(NJ_, KIND_, 1);                                               \
    else SNF_LAUNCH3(NJ_, KIND_, 2);                                                                \
  } while (0)
  const int energy = p.need_raw ? 1 : (p.need_post ? 2 : 0);
  if (nj == 13) {
    if (p.kind == SNF_KIND_FBANK) SNF_LAUNCH(13, SNF_KIND_FBANK);
    else if (p.kind == SNF_KIND_MFCC) SNF_LAUNCH(13, SNF_KIND_MFCC);
    else if (p.kind == SNF_KIND_SPECTROGRAM) SNF_LAUNCH(13, SNF_KIND_SPECTROGRAM);
    else if (p.kind == SNF_KIND_ENERGY) SNF_LAUNCH3(13, SNF_KIND_ENERGY, 0);
    else SNF_LAUNCH(13, SNF_KIND_PLP);
  } else {
    if (p.kind == SNF_KIND_FBANK) SNF_LAUNCH(16, SNF_KIND_FBANK);
    else if (p.kind == SNF_KIND_MFCC) SNF_LAUNCH(16, SNF_KIND_MFCC);
    else if (p.kind == SNF_KIND_SPECTROGRAM) SNF_LAUNCH(16, SNF_KIND_SPECTROGRAM);
    else if (p.kind == SNF_KIND_ENERGY) SNF_LAUNCH3(16, SNF_KIND_ENERGY, 0);
    else SNF_LAUNCH(16, SNF_KIND_PLP);
  }
#undef SNF_LAUNCH6
#undef SNF_LAUNCH5
#undef SNF_LAUNCH4
#undef SNF_LAUNCH3
#undef SNF_LAUNCH
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

}  // namespace snf
