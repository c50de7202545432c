// SNAPSHOT of the round-3 experiment build of fbank512b_kernel (not compiled into the library).
// Every variant profiles/NOTEBOOK.md 4.1 quotes a number for is a bit of the template parameter V (1: split real /
// imaginary exchange tile, 2 / 4: late / early request of the next set's samples, 8: one 16_16_16_16
// typed load per element, 16: compiler-scheduled LDS waits, 32 / 64: L2 touch-ahead, 128: plain dword
// loads, 256: MFMA weights from global memory, 512: one buffer store per set), the occupancy
// configurations are SNF_FBANK512B_CFG (4, 48, 6, 68, 54, 8), the phase ablations and the s_memtime stamps
// are ABL.  To re-run: copy over shennong_amd/csrc/kernels_fbank512b.hip, make, tools/ab_fbank512.cpp with
// SNF_FBANK512B_CFG / _V / _ABL in the variant specs (gpurun_out/run*.txt of round 3 are its outputs).
// fbank512b_kernel: the occupancy-first form of the register-resident 512-point kernel (round 3).
//
// Same mapping and the same arithmetic as fbank512_kernel (kernels_fbank512.hip: wave64 = 4 frames x 16
// lanes, two register FFT-16 passes, real-FFT unpack, mel filterbank as a v_mfma_f32_4x4x1 block chain),
// for the flat, undithered, snip_edges batches the benchmark and most callers run.  What changed, and why
// (profiles/r03_*): fbank512_kernel is bound by vector-instruction issue at 4 waves per SIMD - each wave
// is parked in s_waitcnt 40 % of the time and 4 waves cannot cover that.  This form is built to run 6 - 8
// waves per SIMD:
//   * samples arrive through TYPED buffer loads (tbuffer_load_format, 16_16 SSCALED): the texture path
//     converts int16 -> float, and a second one-sample stream delivers x[2n - 1] to the lane that owns
//     x[2n], x[2n + 1].  No conversion, no v_dot2c, no DPP neighbour move, no v_cndmask for lane 0 - the
//     half-rate instruction classes of phase A are gone, and out-of-window / out-of-buffer reads are
//     range-checked by the buffer descriptor (they return 0);
//   * the 16 x 16 exchange between the passes moves the real and the imaginary parts one after the other
//     through a [16][17] float tile: 1088 bytes per frame instead of 2176, 4352 per wave - 24 to 32
//     waves fit in the LDS of a CU beside two copies of the tables;
//   * nothing is prefetched across the transform: the samples of the next set are requested after the
//     unpack, when most registers are dead, and land during the mel phase; table rows are read in
//     pieces where they are used.  <= 80 VGPRs (6 waves per SIMD) / <= 64 (8).
// Phase A produces bit-identical values to fbank512_kernel (same operations on the same operands); the
// transform, unpack and mel chain are the same code.
//
// Restates the [KALDI-UPSTREAM] per-frame recipe (feature-window.cc ProcessWindow, feature-fbank.cc,
// feature-mfcc.cc, MelBanks::Compute), reached by the reference at shennong/processor/base.py:429-431.
#include <float.h>

#include <cstdlib>

#include "snf_internal.h"
#include "device_fft.h"

namespace snf {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// llvm.amdgcn.raw.tbuffer.load: MTBUF loads with the format in the instruction (gfx9: dfmt | nfmt << 4).
// clang has no builtin for them; binding the intrinsic by name lets the compiler schedule the loads and
// place the s_waitcnt itself.
__device__ f32x2 tbuf_load_s16x2(i32x4 rsrc, int voffset, int soffset, int format, int aux) __asm(
    "llvm.amdgcn.raw.tbuffer.load.v2f32");
__device__ float tbuf_load_s16(i32x4 rsrc, int voffset, int soffset, int format, int aux) __asm(
    "llvm.amdgcn.raw.tbuffer.load.f32");
__device__ f32x4 tbuf_load_s16x4(i32x4 rsrc, int voffset, int soffset, int format, int aux) __asm(
    "llvm.amdgcn.raw.tbuffer.load.v4f32");
constexpr int kFmtS16x2 = 5 | (3 << 4);  // BUF_DATA_FORMAT_16_16, BUF_NUM_FORMAT_SSCALED
constexpr int kFmtS16 = 2 | (3 << 4);    // BUF_DATA_FORMAT_16,    BUF_NUM_FORMAT_SSCALED
constexpr int kFmtS16x4 = 12 | (3 << 4); // BUF_DATA_FORMAT_16_16_16_16, BUF_NUM_FORMAT_SSCALED

constexpr int kHeaderFloats = 16;   // (= kFastHeaderFloats of kernels_fbank512.hip: same table blob)
constexpr int kTileSplit = 1088;    // bytes per frame, split exchange: [16][17] floats = 272 floats
constexpr int kTileFull = 2176;     // bytes per frame, complex exchange: [16][17] float2

// 16 floats at byte offsets 4 i from `base` (one row of the split exchange tile)
__device__ __forceinline__ void read16_b32(const void* base, float (&d)[16]) {
  asm volatile(
      "ds_read_b32 %0, %16\n ds_read_b32 %1, %16 offset:4\n ds_read_b32 %2, %16 offset:8\n"
      "ds_read_b32 %3, %16 offset:12\n ds_read_b32 %4, %16 offset:16\n ds_read_b32 %5, %16 offset:20\n"
      "ds_read_b32 %6, %16 offset:24\n ds_read_b32 %7, %16 offset:28\n ds_read_b32 %8, %16 offset:32\n"
      "ds_read_b32 %9, %16 offset:36\n ds_read_b32 %10, %16 offset:40\n ds_read_b32 %11, %16 offset:44\n"
      "ds_read_b32 %12, %16 offset:48\n ds_read_b32 %13, %16 offset:52\n ds_read_b32 %14, %16 offset:56\n"
      "ds_read_b32 %15, %16 offset:60\n s_waitcnt lgkmcnt(0)"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]),
        "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11]), "=&v"(d[12]), "=&v"(d[13]),
        "=&v"(d[14]), "=&v"(d[15])
      : "v"(lds_addr(base))
      : "memory");
}

// Tile traffic as volatile accesses: program order is kept (a wave's LDS instructions execute in order, so
// its own write -> read needs no wait), nothing is merged into the half-rate ds_read2 / ds_write2 forms,
// and the compiler waits for exactly the operands an instruction needs.
typedef __attribute__((address_space(3))) volatile f32x2 lds_vf2;
typedef __attribute__((address_space(3))) volatile float lds_vf1;
typedef __attribute__((address_space(3))) volatile f32x4 lds_vf4;
__device__ __forceinline__ void vst(float2* p, float2 v) { *((lds_vf2*)p) = f32x2{v.x, v.y}; }
__device__ __forceinline__ void vst(float* p, float v) { *((lds_vf1*)p) = v; }
__device__ __forceinline__ void vst(float4* p, float4 v) { *((lds_vf4*)p) = f32x4{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ float2 vld(const float2* p) {
  const f32x2 v = *((lds_vf2*)p);
  return make_float2(v[0], v[1]);
}
__device__ __forceinline__ float vld(const float* p) { return *((lds_vf1*)p); }
__device__ __forceinline__ float4 vld(const float4* p) {
  const f32x4 v = *((lds_vf4*)p);
  return make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace

// ENERGY: 0 = no log-energy column, 1 = raw (before pre-emphasis/window), 2 = after the window
// V (variant bits): 1 = split exchange tile; 2 / 4 = samples of the next set requested late (after the
// unpack) / early (right after phase A: a whole iteration ahead); 8 = one 16_16_16_16 load per element
// (x[2n-2] .. x[2n+1]) instead of a 16_16 and a 16 load; 16 = tile traffic as volatile C++ accesses (the
// compiler places partial waits and overlaps the transfers with the butterflies) instead of asm blocks
// with full waits
template <int NJ, int KIND, int ENERGY, int V, int WAVES, int OCC, int ABL = 0>
__global__ __launch_bounds__(WAVES * 64, OCC) void fbank512b_kernel(const Fast512Params p, const BatchArgs b,
                                                                   float* __restrict__ out,
                                                                   double* __restrict__ energy_out) {
  constexpr bool SPLIT = (V & 1) != 0, LATE = (V & 2) != 0, EARLY = (V & 4) != 0, XYZW = (V & 8) != 0,
                 VIS = (V & 16) != 0;
  // 32 / 64: one load instruction per iteration touches every 128-byte line of the set this wave works on
  // two / three iterations from now (L2 prefetch).  A wave holds only 1.3 KB of NEW samples per set;
  // with 16 waves per CU that is 20 KB in flight per CU whenever all of them are waiting at once -
  // too little to cover the HBM latency at 8 TB/s (the kernel with every phase left out still took 0.70 ms)
  constexpr int TOUCH = (V & 32) ? 2 : ((V & 64) ? 3 : 0);
  // 128: phase A of fbank512_kernel - plain dword loads (two int16 per element), conversion on the vector
  // pipe, the left neighbour through DPP.  Typed buffer loads cost 9-11 ns per instruction and CU on the
  // texture path against 3.8 ns for global_load_dword (tools/ubench_vmem.hip): 26 of them per set are a
  // 0.70 ms floor of their own.
  constexpr bool RAW = (V & 128) != 0;
  // 256: the MFMA weight table (9 KB of the 16.5 KB of tables) is read from global memory (L1 / L2) instead
  // of LDS, which lets two 16-wave workgroups share a CU (8 waves per SIMD)
  constexpr bool GA = (V & 256) != 0;
  // 512: the feature rows leave through ONE unconditional buffer store per set whose descriptor covers the
  // rows that exist (lanes with nothing to store, and frames past the end, fall outside it and are
  // dropped by the range check).  With the stores inside `if`s the compiler cannot count the vector-memory
  // operations issued after the sample loads of the next set, falls back to s_waitcnt vmcnt(0) at the top
  // of every iteration and the wave sits there until the STORES of the previous set are acknowledged
  // (1300 - 2000 clocks of 11 000 per iteration, s_memtime stamps in profiles/r03_*).  The host selects it
  // for layouts without partial groups / energy column (num_bins % 4 == 0).
  constexpr bool BST = (V & 512) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem);
  // ---- stage the tables into LDS (the only workgroup-wide barrier of the kernel) -------------------
  // (GA: the blob without its MFMA weight rows [off_mm_a, off_mm_lane); later offsets move down)
  const int ga_gap = GA ? p.off_mm_lane - p.off_mm_a : 0;
  for (int i = threadIdx.x; i < p.table_floats - ga_gap; i += WAVES * 64)
    tab[i] = p.tables[i < p.off_mm_a ? i : i + ga_gap];
  __syncthreads();
  const float2* __restrict__ t_win = reinterpret_cast<const float2*>(tab + kHeaderFloats);
  const float2* __restrict__ t_tw16 = t_win + 16 * 18;
  const float2* __restrict__ t_tw512 = t_tw16 + 16 * 18;
  const float4* __restrict__ t_mm_a =
      reinterpret_cast<const float4*>(GA ? p.tables + p.off_mm_a : tab + p.off_mm_a);
  const float* __restrict__ t_lifter = tab + p.off_lifter - ga_gap;
  const float4* __restrict__ t_dd_v = reinterpret_cast<const float4*>(tab + p.off_dd_v - ga_gap);

  constexpr int kTile = SPLIT ? kTileSplit : kTileFull;
  // (wid through readfirstlane: the set index and everything derived from it stay in scalar registers)
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l = lane & 15, q = lane >> 4;
  const int tab_bytes = ((p.table_floats - ga_gap) * 4 + 255) & ~255;
  char* wave_base = smem + tab_bytes + (wid * 4 + q) * kTile;
  float2* tile = reinterpret_cast<float2*>(wave_base);   // partner rows (and the complex exchange tile)
  float* ftile = reinterpret_cast<float*>(wave_base);    // split exchange: [16][17] floats
  // power tile [257]: aliases the exchange tile.  Frames sit 272 floats apart (split; 16 mod 64 dwords:
  // the four frames of an MFMA block read four disjoint 16-byte slots) or 544 + 16 q (complex tile)
  float* ptile = reinterpret_cast<float*>(wave_base) + (SPLIT ? 0 : q * 16);
  // the padding column of the complex tile is never written by the transposes and the mel phase reads a
  // few floats past the power tile with zero weights (see fbank512_kernel); the split tile is written
  // densely up to float 270 by every exchange, reads stop at float 259
  if (!SPLIT) tile[l * 17 + 16] = make_float2(0.0f, 0.0f);
  // MFMA view of the wave: lane = 4 b + j, block b, frame j of the set (lane-constant)
  // (where the lane's B operands start, the first of the 4 bins it stores, the 0/1 factors of the partial
  // sums of a split group: read from the LDS table where they are used - they would otherwise hold five
  // registers through the transform)
  const int mj = lane & 3;
  const float* __restrict__ mm_lane = tab + p.off_mm_lane - ga_gap + lane;
  const float* __restrict__ mtile =
      reinterpret_cast<const float*>(smem + tab_bytes + (wid * 4 + mj) * kTile) + (SPLIT ? 0 : mj * 16);

  const float win_len_f = static_cast<float>(p.win_len), inv_win_len = 1.0f / win_len_f;
  const int64_t n_sets = (b.total_frames + 3) >> 2;
  const int64_t set_stride = static_cast<int64_t>(gridDim.x) * WAVES;
  const int64_t last_frame = b.total_frames - 1;
  const int64_t total_samples = b.sample_offsets[b.n_utts];
  int64_t set = static_cast<int64_t>(blockIdx.x) * WAVES + wid;
  if (set >= n_sets) return;

  // NJ = 13: the 25 ms / 16 kHz window (only element j = 12 can fall outside it); NJ = 16: any other
  // window that pads to 512 samples, with a per-element test
  const bool in_last = 2 * (l + 16 * (NJ - 1)) < p.win_len;
  auto in_window = [&](int j) -> bool {
    if (NJ == 13) return j < NJ - 1 || in_last;
    return 2 * (l + 16 * j) < p.win_len;
  };
  auto start_of = [&](int64_t s) -> int64_t {
    const int64_t gi = s * 4 + q;
    return b.frame_start[gi < last_frame ? gi : last_frame];
  };
  // The samples of a frame set through one buffer descriptor: base = the first sample of the set's first
  // frame (wave-uniform), lane offset = its own frame's distance from it + its element.  Element n =
  // l + 16 j holds x[2n], x[2n + 1] (one 16_16 load) and needs x[2n - 1] (one 16 load, two bytes lower;
  // Kaldi's Preemphasize defines the predecessor of x[0] as x[0] itself: lane 0 reads it at j = 0).
  float xe[NJ], xo[NJ], xp[NJ];
  int raw[NJ];
  typedef int __attribute__((aligned(2))) int_a2;
  auto request = [&](int64_t st) {
    if (RAW) {
      const int16_t* __restrict__ wp = b.wave + st;
      const int16_t* __restrict__ wl = wp + 2 * l;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        raw[j] = *reinterpret_cast<const int_a2*>((NJ == 13 && j < NJ - 1) || in_window(j) ? wl + 32 * j : wp);
      return;
    }
    const int lo = __builtin_amdgcn_readfirstlane(static_cast<int>(st));
    const int hi = __builtin_amdgcn_readfirstlane(static_cast<int>(st >> 32));
    const int64_t st0 = (static_cast<int64_t>(hi) << 32) | static_cast<unsigned>(lo);
    const unsigned long long base = reinterpret_cast<unsigned long long>(b.wave + st0);
    const int64_t rem = (total_samples - st0) * 2;
    i32x4 rs;
    rs[0] = static_cast<int>(base);
    rs[1] = static_cast<int>((base >> 32) & 0xffff);
    rs[2] = rem > 0xffffffffll ? -1 : static_cast<int>(rem);
    rs[3] = 0x00020000;
    const int voff = static_cast<int>(st - st0) * 2 + 4 * l;
    if (XYZW) {
      // x[2n - 2], x[2n - 1], x[2n], x[2n + 1]; lane 0 reads x[0] .. x[3] at j = 0 (nothing lies below the
      // first frame of the buffer) and takes its pair from the front
      const int voff_q0 = l == 0 ? voff : voff - 4;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4 v = tbuf_load_s16x4(rs, (j == 0 ? voff_q0 : voff - 4) + 64 * j, 0, kFmtS16x4, 0);
        if (j == 0) {
          xe[j] = l == 0 ? v[0] : v[2];
          xo[j] = l == 0 ? v[1] : v[3];
          xp[j] = l == 0 ? v[0] : v[1];
        } else {
          xe[j] = v[2];
          xo[j] = v[3];
          xp[j] = v[1];
        }
      }
    } else {
      const int voff_p0 = l == 0 ? voff : voff - 2;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x2 c = tbuf_load_s16x2(rs, voff + 64 * j, 0, kFmtS16x2, 0);
        xe[j] = c[0];
        xo[j] = c[1];
        xp[j] = tbuf_load_s16(rs, (j == 0 ? voff_p0 : voff - 2) + 64 * j, 0, kFmtS16, 0);
      }
    }
  };
  constexpr bool AHEAD = LATE || EARLY;
  constexpr int kNext = AHEAD ? 1 : 0;  // the set `request` is called for, counted from the current one
  int64_t start_next = start_of(set + kNext * set_stride);
  if (AHEAD) request(start_of(set));
  if (BST && AHEAD) {
    // one dropped store behind the first request: the loop is entered with the same sequence of vector-
    // memory operations in flight as its back edge carries (loads, then one store), so the waits at the
    // top of the body can be counted instead of draining everything
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000);
    if (KIND == SNF_KIND_MFCC) __builtin_amdgcn_raw_buffer_store_b32(0u, none, -1, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, none, -1, 0, 2);
  }
  // first sample of the set to touch in the coming iteration (fetched one iteration before it is used)
  int64_t far_pending = TOUCH > 0 ? start_of(set + (kNext + TOUCH) * set_stride) : 0;
  int touched = 0;
  const int touch_off = 64 * (l < 6 ? l : 6);  // samples: lines 0 .. 6 of the frame's 800 bytes

  // (ABL & 1024: timing experiment - wave 0 of workgroup 0 stamps s_memtime at the phase boundaries of
  // its 20th iteration and leaves the stamps in the first output row)
  long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int iter = 0;
#define SNF_STAMP(I_)                                                          \
  do {                                                                         \
    if ((ABL & 1024) && iter == 20) {                                          \
      __builtin_amdgcn_sched_barrier(0);                                       \
      stamp[I_] = __builtin_readcyclecounter();                                \
      __builtin_amdgcn_sched_barrier(0);                                       \
    }                                                                          \
  } while (0)
  for (; set < n_sets; set += set_stride, ++iter) {
    SNF_STAMP(0);
    const int64_t g = set * 4 + q;  // global frame = output row
    const bool valid = g <= last_frame;
    if (TOUCH > 0) {
      // (last iteration's touch is consumed here - long landed - so that the compiler's wait for it
      // costs nothing; then the lines of the set TOUCH iterations past the requested one are touched)
      asm volatile("" : : "v"(touched));
      touched = *reinterpret_cast<const volatile int*>(b.wave + far_pending + touch_off);
      far_pending = start_of(set + (kNext + TOUCH + 1) * set_stride);
    }
    if (!AHEAD) {
      request(start_next);
      start_next = start_of(set + set_stride);
    }

    // ---- A: DC removal, pre-emphasis, window (Kaldi's ProcessWindow order) -------------------------
    float part = 0.0f;
    if (RAW) {
      int part_i = 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        xe[j] = static_cast<float>(static_cast<short>(raw[j] & 0xffff));
        xo[j] = static_cast<float>(raw[j] >> 16);
        const int both = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, raw[j]), short2v{1, 1}, part_i, false);
        part_i = in_window(j) ? both : part_i;
      }
      part = static_cast<float>(part_i);
      if (EARLY) {
        // the conversions are the last readers of `raw`: pin them so that the loads of the next set reuse
        // the same registers
#pragma unroll
        for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(xe[j]), "+v"(xo[j]) : : "memory");
        asm volatile("" : "+v"(part) : : "memory");
        request(start_next);
        start_next = start_of(set + 2 * set_stride);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        // the samples are integers and the partial sums stay below 2^24: exact in any order
        const float s2 = xe[j] + xo[j];
        part += in_window(j) ? s2 : 0.0f;
      }
    }
    float neg_mean = 0.0f;
    if (p.remove_dc) {
      const float sum = row_sum16(part);
      // sum / N correctly rounded without the IEEE division sequence (see fbank512_kernel)
      const float qv = sum * inv_win_len;
      neg_mean = -__builtin_fmaf(__builtin_fmaf(-qv, win_len_f, sum), inv_win_len, qv);
    }
    SNF_STAMP(1);
    float2 z[16];
    float e_raw = 0.0f, e_post = 0.0f;
    float rot_prev = xe[0] + neg_mean;  // (RAW) lane 0, j = 0: x[-1] := x[0] (Kaldi Preemphasize)
#pragma unroll
    for (int jj = 0; jj < 16; jj += 2) {
      float4 w4;
      if (jj < NJ)  // (zero outside the window; rows are 16-byte aligned)
        w4 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(t_win + l * 18 + jj, 16));
#pragma unroll
      for (int j = jj; j < jj + 2; ++j) {
        if (j < NJ) {
          if (ABL & 256) {
            z[j] = make_float2(xe[j] + xp[j], xo[j] + neg_mean);
            continue;
          }
          const float ae = xe[j] + neg_mean, ao = xo[j] + neg_mean;
          float ap;
          if (RAW) {
            // the left neighbour x[2n-1] is the odd sample of lane l - 1 (same j), or lane 15 of j - 1
            const float rot = dpp_row_ror<0x121>(ao);
            ap = l == 0 ? rot_prev : rot;
            rot_prev = rot;
          } else {
            ap = xp[j] + neg_mean;
          }
          const float2 w = (j & 1) ? make_float2(w4.z, w4.w) : make_float2(w4.x, w4.y);
          if (ENERGY == 1 && in_window(j)) e_raw += ae * ae + ao * ao;
          const float ye = (ae - p.preemph * ap) * w.x;
          const float yo = (ao - p.preemph * ae) * w.y;
          z[j] = make_float2(ye, yo);
          if (ENERGY == 2) e_post += ye * ye + yo * yo;
        } else {
          z[j] = make_float2(0.0f, 0.0f);
        }
      }
    }
    float e_lin = 0.0f;
    if (ENERGY != 0) e_lin = row_sum16(ENERGY == 1 ? e_raw : e_post);
    if (EARLY && !RAW) {
      // (the values above are the last readers of the sample registers: pin them so that the loads below
      // reuse those registers)
#pragma unroll
      for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(z[j].x), "+v"(z[j].y) : : "memory");
      request(start_next);
      start_next = start_of(set + 2 * set_stride);
    }
    __builtin_amdgcn_sched_barrier(0);

    SNF_STAMP(2);
    // ---- B: pass 1 (FFT over j), inter-pass twiddle W256^(l k2), exchange --------------------------
    // (ABL bits: phases left out in timing experiments - the results are then meaningless)
    if (!(ABL & 1)) fft16(z);
#pragma unroll
    for (int kk = 0; kk < ((ABL & 2) ? 0 : 16); kk += 4) {
      const float4 ta = *reinterpret_cast<const float4*>(__builtin_assume_aligned(t_tw16 + l * 18 + kk, 16));
      const float4 tb = *reinterpret_cast<const float4*>(__builtin_assume_aligned(t_tw16 + l * 18 + kk + 2, 16));
      if (kk != 0) z[kk] = cmul(z[kk], make_float2(ta.x, ta.y));
      z[kk + 1] = cmul(z[kk + 1], make_float2(ta.z, ta.w));
      z[kk + 2] = cmul(z[kk + 2], make_float2(tb.x, tb.y));
      z[kk + 3] = cmul(z[kk + 3], make_float2(tb.z, tb.w));
    }
    SNF_STAMP(3);
    if (ABL & 4) {
    } else if (VIS && SPLIT) {
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) vst(ftile + k2 * 17 + l, z[k2].x);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) z[k2].x = vld(ftile + l * 17 + k2);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) vst(ftile + k2 * 17 + l, z[k2].y);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) z[k2].y = vld(ftile + l * 17 + k2);
    } else if (VIS) {
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) vst(tile + k2 * 17 + l, z[k2]);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) z[k2] = vld(tile + l * 17 + k2);
    } else if (SPLIT) {
      // real parts, then imaginary parts, through the same [16][17] float tile (conflict-free both ways)
      float r[16];
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) ftile[k2 * 17 + l] = z[k2].x;
      wave_lds_sync();
      read16_b32(ftile + l * 17, r);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) z[k2].x = r[k2];
      wave_lds_sync();
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) ftile[k2 * 17 + l] = z[k2].y;
      wave_lds_sync();
      read16_b32(ftile + l * 17, r);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) z[k2].y = r[k2];
    } else {
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) tile[k2 * 17 + l] = z[k2];
      wave_lds_sync();
      read16_b64(tile + l * 17, z);
    }
    // (VIS: the reads above are still in flight here; the barrier only keeps later phases from being
    // hoisted above the transform - the compiler waits for each operand where the butterflies need it)
    __builtin_amdgcn_sched_barrier(0);
    SNF_STAMP(4);
    // ---- C: pass 2 (FFT over n1): z[k1] = Z[l + 16 k1] ---------------------------------------------
    if (!(ABL & 8)) fft16(z);
    SNF_STAMP(5);
    __builtin_amdgcn_sched_barrier(0);
    if (!VIS) wave_lds_sync();

    // ---- D: real-FFT unpack + power (x4): the partner Z[256 - k] of k = l + 16 k1 (k1 < 8) is
    // (16 - l) + 16 (15 - k1): the upper half of the spectrum goes through the tile, rows 0..7 ---------
#pragma unroll
    for (int r = 0; r < ((ABL & 16) ? 0 : 8); ++r) {
      if (VIS) vst(tile + r * 16 + l, z[r + 8]);
      else tile[r * 16 + l] = z[r + 8];
    }
    if (!VIS) wave_lds_sync();
    float pk[8], pm[8];  // 4 P[k], 4 P[256 - k]
    {
      const float2* partner = tile + (16 - l);
      float2 zpart[8];
      if (ABL & 16) {
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) zpart[k1] = z[15 - k1];
      } else if (VIS) {
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) zpart[k1] = vld(partner + 16 * (7 - k1));
      } else {
        read8_b64_rev128(partner, zpart);  // zpart[k1] = Z[256 - l - 16 k1]
      }
      float4 w512q[4];
      read_quads<4>(t_tw512 + l * 10, w512q);  // W512^(l + 16 k1)
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) {
        const float2 zk = z[k1];
        const float2 zp = zpart[k1];
        const float2 w = (k1 & 1) ? make_float2(w512q[k1 >> 1].z, w512q[k1 >> 1].w)
                                  : make_float2(w512q[k1 >> 1].x, w512q[k1 >> 1].y);
        if (ABL & 32) {
          pk[k1] = zk.x + zp.y;
          pm[k1] = zk.y + zp.x;
          continue;
        }
        const float c_re = zk.x + zp.x, c_im = zk.y - zp.y;
        const float d_re = zk.y + zp.y, d_im = zp.x - zk.x;
        const float t_re = d_re * w.x - d_im * w.y, t_im = d_re * w.y + d_im * w.x;
        const float a_re = c_re + t_re, a_im = c_im + t_im;
        const float b_re = c_re - t_re, b_im = t_im - c_im;
        pk[k1] = a_re * a_re + a_im * a_im;
        pm[k1] = b_re * b_re + b_im * b_im;
      }
    }
    if (l == 0) {  // k = 0: DC and Nyquist (lane 0 read scratch for k1 = 0)
      const float dc = z[0].x + z[0].y;
      pk[0] = 4.0f * dc * dc;
      const float ny = z[0].x - z[0].y;
      pm[0] = 4.0f * ny * ny;
    }
    const float p128 = 4.0f * (z[8].x * z[8].x + z[8].y * z[8].y);  // k = 128: lane 0, register 8
    SNF_STAMP(6);
    if (!VIS) wave_lds_sync();
    // ---- E: power tile ------------------------------------------------------------------------------
    {
      float* pmirror = ptile + (144 - l);
#pragma unroll
      for (int k1 = 0; k1 < ((ABL & 64) ? 1 : 8); ++k1) {
        if (VIS) {
          vst(ptile + l + 16 * k1, pk[k1]);
          vst(pmirror + 16 * (7 - k1), pm[k1]);
        } else {
          ptile[l + 16 * k1] = pk[k1];
          pmirror[16 * (7 - k1)] = pm[k1];  // index 256 - l - 16 k1
        }
      }
      if (l == 0) {
        if (VIS) vst(ptile + 128, p128);
        else ptile[128] = p128;
      }
    }
    if (!VIS) wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
    // the samples of this wave's next set: requested now that the transform registers are dead, they
    // land during the mel phase (the final request of a wave re-reads the last frame)
    if (LATE) {
      request(start_next);
      start_next = start_of(set + 2 * set_stride);
      __builtin_amdgcn_sched_barrier(0);
    }

    SNF_STAMP(7);
    // ---- log-energy column ---------------------------------------------------------------------------
    float log_energy = 0.0f;
    if (ENERGY != 0) {
      if (KIND == SNF_KIND_PLP) {
        if (valid && l == 0) energy_out[g] = static_cast<double>(e_lin);
      } else {
        log_energy = fast_log(floor_eps(e_lin));
        if (p.has_floor && log_energy < p.log_energy_floor) log_energy = p.log_energy_floor;
      }
    }
    float* __restrict__ row = out + g * static_cast<int64_t>(p.out_cols);
    if (KIND == SNF_KIND_SPECTROGRAM) {
      if (valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 pw = VIS ? vld(reinterpret_cast<const float4*>(ptile + 4 * l + 64 * i))
                                : *reinterpret_cast<const float4*>(ptile + 4 * l + 64 * i);
          f32x4_a4 v = {fast_log(fmaxf(0.25f * pw.x, FLT_EPSILON)), fast_log(fmaxf(0.25f * pw.y, FLT_EPSILON)),
                        fast_log(fmaxf(0.25f * pw.z, FLT_EPSILON)), fast_log(fmaxf(0.25f * pw.w, FLT_EPSILON))};
          if (i == 0 && l == 0) v[0] = log_energy;
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4_a4*>(row + 4 * l + 64 * i));
        }
        if (l == 0) row[256] = fast_log(fmaxf(0.25f * (VIS ? vld(ptile + 256) : ptile[256]), FLT_EPSILON));
      }
    } else {
      // ---- F: mel filterbank on the matrix pipe (see fbank512_kernel) -------------------------------
      const bool mvalid = set * 4 + mj <= last_frame;
      const int mm_start = reinterpret_cast<const int*>(mm_lane)[0];
      const float4* __restrict__ bsrc = reinterpret_cast<const float4*>(mtile + mm_start);
      const float4* __restrict__ asrc = t_mm_a + lane;
      f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
      // two operand sets in flight; mm_quads is even and the weight table ends with a row of zeros, so
      // the last look-ahead read needs no test (the B side reads finite tile contents)
      float4 a0 = asrc[0], x0 = VIS ? vld(bsrc) : bsrc[0];
      if (ABL & 128) {
        acc0 = f32x4{a0.x + x0.x + pk[0] + pk[3] + pm[2] + pm[5], a0.y + x0.y + pk[1] + pk[4] + pm[1] + pm[6],
                     a0.z + x0.z + pk[2] + pk[5] + pm[0] + pm[7], a0.w + x0.w + pk[6] + pk[7] + pm[3] + pm[4]};
      }
      const int n_quads = (ABL & 128) ? 0 : p.mm_quads;  // wave-uniform
      int t = 0;
      for (; t + 1 < n_quads; t += 2) {
        const float4 a1 = asrc[(t + 1) * 64], x1 = VIS ? vld(bsrc + t + 1) : bsrc[t + 1];
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, x0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, x0.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, x0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, x0.w, acc1, 0, 0, 0);
        a0 = asrc[(t + 2) * 64];  // (behind the table: rows of zeros)
        x0 = VIS ? vld(bsrc + t + 2) : bsrc[t + 2];
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, x1.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, x1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.z, x1.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.w, x1.w, acc1, 0, 0, 0);
      }
      if (t < n_quads) {  // an odd quad on its own
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, x0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, x0.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, x0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, x0.w, acc1, 0, 0, 0);
      }
      float mel[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) mel[i] = acc0[i] + acc1[i];
      SNF_STAMP(8);
      const int mm_out = reinterpret_cast<const int*>(mm_lane)[64];
      if (p.mm_levels > 1) {
        const float mm_f1 = mm_lane[128], mm_f2 = mm_lane[192], mm_f3 = mm_lane[256];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float own = mel[i];
          fmac_row_shl<4>(mel[i], own, mm_f1);
          fmac_row_shl<8>(mel[i], own, mm_f2);
          if (p.mm_levels > 3) fmac_row_shl<12>(mel[i], own, mm_f3);
        }
      }
      const int mel_col = (KIND == SNF_KIND_FBANK && p.use_energy && !p.htk_compat) ? 1 : 0;
      if (KIND == SNF_KIND_FBANK || KIND == SNF_KIND_PLP) {
        if (KIND == SNF_KIND_FBANK && p.use_log) {
#pragma unroll
          for (int i = 0; i < 4; ++i) mel[i] = fast_log(floor_eps(mel[i]));
        }
        if (BST) {
          const int64_t left = b.total_frames - set * 4;
          const int rows = left < 4 ? static_cast<int>(left) : 4;
          const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
              out + set * 4 * static_cast<int64_t>(p.out_cols), 0, rows * p.out_cols * 4, 0x00020000);
          const int ooff = mm_out >= 0 ? (mj * p.out_cols + mel_col + mm_out) * 4 : -1;
          __builtin_amdgcn_raw_buffer_store_b128(
              __builtin_bit_cast(u32x4, f32x4{mel[0], mel[1], mel[2], mel[3]}), orsrc, ooff, 0, 2);
        } else if (mvalid && mm_out >= 0) {
          // output row of the MFMA view: lane 4 b + j -> frame j of the set
          float* __restrict__ dst = out + (set * 4 + mj) * static_cast<int64_t>(p.out_cols) + mel_col + mm_out;
          if (mm_out + 4 <= p.num_bins) {
            __builtin_nontemporal_store(f32x4_a4{mel[0], mel[1], mel[2], mel[3]}, reinterpret_cast<f32x4_a4*>(dst));
          } else {
#pragma unroll
            for (int i = 0; i < 3; ++i)
              if (mm_out + i < p.num_bins) dst[i] = mel[i];
          }
        }
        if (!BST && KIND == SNF_KIND_FBANK && p.use_energy && valid && l == 0)
          row[p.htk_compat ? p.num_bins : 0] = log_energy;
      }
      if (KIND == SNF_KIND_MFCC) {
        // log-mel of frame j back to its (now idle) power tile; DCT-II + lifter on the vector pipe: lane
        // l of a frame's row owns cepstrum l and walks the log-mel 4 bins at a time
        if (!VIS) wave_lds_sync();
        if (mm_out >= 0) {
          const float4 lm = make_float4(fast_log(floor_eps(mel[0])), fast_log(floor_eps(mel[1])),
                                        fast_log(floor_eps(mel[2])), fast_log(floor_eps(mel[3])));
          float4* dstq = reinterpret_cast<float4*>(const_cast<float*>(mtile) + mm_out);
          if (VIS) vst(dstq, lm);
          else *dstq = lm;
        }
        if (!VIS) wave_lds_sync();
        const float4* __restrict__ dw = t_dd_v + l;
        const float4* __restrict__ dx = reinterpret_cast<const float4*>(ptile);
        float v = 0.0f;
#pragma unroll 2
        for (int g4 = 0; g4 < p.dd_groups; ++g4) {
          const float4 w = dw[g4 * 16], x = VIS ? vld(dx + g4) : dx[g4];
          v += w.x * x.x;
          v += w.y * x.y;
          v += w.z * x.z;
          v += w.w * x.w;
        }
        v *= t_lifter[l];
        if (l == 0 && p.use_energy) v = log_energy;
        int oc = l;
        if (p.htk_compat) {
          oc = l == 0 ? p.num_ceps - 1 : l - 1;
          if (l == 0 && !p.use_energy)
            v = static_cast<float>(static_cast<double>(v) * 1.4142135623730950488016887);
        }
        if (BST) {
          const int64_t left = b.total_frames - set * 4;
          const int rows = left < 4 ? static_cast<int>(left) : 4;
          const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
              out + set * 4 * static_cast<int64_t>(p.out_cols), 0, rows * p.out_cols * 4, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc,
                                                l < p.num_ceps ? (q * p.out_cols + oc) * 4 : -1, 0, 0);
        } else if (valid && l < p.num_ceps) {
          row[oc] = v;
        }
      }
    }
    if (!VIS) wave_lds_sync();  // the tile is reused by the next frame set
    SNF_STAMP(9);
    if ((ABL & 1024) && iter == 21 && blockIdx.x == 0 && threadIdx.x == 0) {
      long long* dbg = reinterpret_cast<long long*>(out);
#pragma unroll
      for (int i = 0; i < 10; ++i) dbg[i] = stamp[i];
    }
  }
#undef SNF_STAMP
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// Flat batches (no per-utterance warps, no fused deltas), snip_edges, no dither, vector-pipe DCT.
bool fbank512b_eligible(const Fast512Params& p, const BatchArgs& b) {
  if (const char* knob = getenv("SNF_FBANK512_OLD"))
    if (knob[0] == '1') return false;
  if (p.dual || p.fused_delta || b.blk_utt != nullptr) return false;
  if (!p.snip_edges || p.dither != 0.0f || p.dct_mfma) return false;
  if (p.kind != SNF_KIND_FBANK && p.kind != SNF_KIND_MFCC && p.kind != SNF_KIND_PLP &&
      p.kind != SNF_KIND_SPECTROGRAM)
    return false;
  return true;
}

namespace {

template <int NJ, int KIND, int ENERGY, int V, int WAVES, int OCC, int ABL = 0>
int launch_one(const Fast512Params& q, const BatchArgs& b, float* out, double* energy_out, hipStream_t stream) {
  const int tab_bytes = ((q.table_floats - ((V & 256) ? q.off_mm_lane - q.off_mm_a : 0)) * 4 + 255) & ~255;
  const size_t lds = static_cast<size_t>(tab_bytes) + WAVES * 4 * ((V & 1) ? kTileSplit : kTileFull);
  if (lds > 160 * 1024) return set_error(SNF_E_RUNTIME, "fbank512b: tables do not fit in LDS");
  const int64_t n_sets = (b.total_frames + 3) / 4;
  int64_t blocks = (n_sets + WAVES - 1) / WAVES;
  const int64_t resident = 256 * ((OCC * 4) / WAVES);  // workgroups the chip holds at this occupancy
  if (blocks > resident) blocks = resident;
  auto kern = fbank512b_kernel<NJ, KIND, ENERGY, V, WAVES, OCC, ABL>;
  if (lds > 64 * 1024)
    SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(WAVES * 64), lds, stream, q, b, out,
                     energy_out);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

template <int NJ, int KIND, int ENERGY>
int launch_cfg(const Fast512Params& q, const BatchArgs& b, float* out, double* energy_out, hipStream_t stream) {
  // occupancy configuration: SNF_FBANK512B_CFG = 4 (complex tile, 4 waves per SIMD), 6, 7, 8 (split tile)
  int cfg = 4;
  if (const char* knob = getenv("SNF_FBANK512B_CFG")) cfg = atoi(knob);
  // variant bits (see the kernel): SNF_FBANK512B_V; experiments are instantiated for the 25 ms shape only
  int v = 2;
  if (const char* knob = getenv("SNF_FBANK512B_V")) v = atoi(knob);
  const int tab_bytes = ((q.table_floats - ((v & 256) ? q.off_mm_lane - q.off_mm_a : 0)) * 4 + 255) & ~255;
  auto fits = [&](int waves, int wgs, int tile) {
    return static_cast<size_t>(wgs) * (tab_bytes + waves * 4 * tile + 512) <= 160 * 1024;
  };
#define SNF_V(VV_, WAVES_, OCC_, S_) \
  case VV_: return launch_one<13, kK, kE, 2 * VV_ + S_, WAVES_, OCC_>(q, b, out, energy_out, stream)
#define SNF_B(WAVES_, OCC_, S_)                                                                      \
  do {                                                                                              \
    if (NJ == 13 && ((KIND == SNF_KIND_FBANK && ENERGY == 0) || (KIND == SNF_KIND_MFCC && ENERGY == 1))) { \
      constexpr int kK = (KIND == SNF_KIND_MFCC) ? SNF_KIND_MFCC : SNF_KIND_FBANK;                   \
      constexpr int kE = (KIND == SNF_KIND_MFCC) ? 1 : 0;                                           \
      switch (v >> 1) {                                                                             \
        SNF_V(5, WAVES_, OCC_, S_); \
        SNF_V(64, WAVES_, OCC_, S_); \
        SNF_V(65, WAVES_, OCC_, S_); \
        SNF_V(66, WAVES_, OCC_, S_); \
        SNF_V(320, WAVES_, OCC_, S_); \
        SNF_V(321, WAVES_, OCC_, S_); \
        SNF_V(322, WAVES_, OCC_, S_); \
        default: break;                                                                             \
      }                                                                                             \
    }                                                                                               \
    return launch_one<NJ, KIND, ENERGY, 2 + S_, WAVES_, OCC_>(q, b, out, energy_out, stream);        \
  } while (0)
  if (const char* knob = getenv("SNF_FBANK512B_ABL")) {  // timing experiments: phases left out
    if (NJ == 13 && KIND == SNF_KIND_FBANK && ENERGY == 0) {
      switch (atoi(knob)) {
#define SNF_A(A_) case A_: return (v & 512) ? launch_one<13, SNF_KIND_FBANK, 0, 644, 16, 4, A_>(q, b, out, energy_out, stream) : launch_one<13, SNF_KIND_FBANK, 0, 132, 16, 4, A_>(q, b, out, energy_out, stream)
        SNF_A(511); SNF_A(1024); SNF_A(1535);
#undef SNF_A
        default: break;
      }
    }
  }
  if (cfg == 8 && fits(16, 2, kTileSplit)) SNF_B(16, 8, 1);
  if (cfg == 6 && fits(12, 2, kTileSplit)) SNF_B(12, 6, 1);
  if (cfg == 48 && fits(8, 2, kTileSplit)) SNF_B(8, 4, 1);
  if (cfg == 68 && fits(8, 3, kTileSplit)) SNF_B(8, 6, 1);
  if (cfg == 54 && fits(5, 4, kTileSplit)) SNF_B(5, 5, 1);
  SNF_B(16, 4, 0);
#undef SNF_B
#undef SNF_V
}

template <int NJ, int KIND>
int launch_energy(const Fast512Params& q, const BatchArgs& b, float* out, double* energy_out, hipStream_t stream) {
  const int energy = q.need_raw ? 1 : (q.need_post ? 2 : 0);
  if (energy == 0) return launch_cfg<NJ, KIND, 0>(q, b, out, energy_out, stream);
  if (energy == 1) return launch_cfg<NJ, KIND, 1>(q, b, out, energy_out, stream);
  return launch_cfg<NJ, KIND, 2>(q, b, out, energy_out, stream);
}

template <int NJ>
int launch_kind(const Fast512Params& q, const BatchArgs& b, float* out, double* energy_out, hipStream_t stream) {
  if (q.kind == SNF_KIND_FBANK) return launch_energy<NJ, SNF_KIND_FBANK>(q, b, out, energy_out, stream);
  if (q.kind == SNF_KIND_MFCC) return launch_energy<NJ, SNF_KIND_MFCC>(q, b, out, energy_out, stream);
  if (q.kind == SNF_KIND_PLP) return launch_energy<NJ, SNF_KIND_PLP>(q, b, out, energy_out, stream);
  return launch_energy<NJ, SNF_KIND_SPECTROGRAM>(q, b, out, energy_out, stream);
}

}  // namespace

int launch_fbank512b(const Fast512Params& p, const BatchArgs& b, float* out, int out_cols, double* energy_out,
                     hipStream_t stream) {
  if (b.total_frames <= 0) return SNF_OK;
  Fast512Params q = p;
  q.out_cols = out_cols;
  if ((p.win_len + 31) / 32 == 13) return launch_kind<13>(q, b, out, energy_out, stream);
  return launch_kind<16>(q, b, out, energy_out, stream);
}

}  // namespace snf
