// fbank512b_kernel: the flat-batch form of the register-resident 512-point kernel (round 3).
//
// Same mapping and, bit for bit, the same arithmetic as fbank512_kernel (kernels_fbank512.hip: wave64 = 4
// frames x 16 lanes, two register FFT-16 passes, real-FFT unpack, mel filterbank as a v_mfma_f32_4x4x1
// block chain) for the batches the benchmark and most callers run: flat scheduling (no per-utterance VTLN
// tables, no fused deltas), snip_edges, no dither.  fbank512_kernel keeps every other mode.
//
// What round 3 measured about the 512-point kernel, and what this form does about it (profiles/NOTEBOOK.md 4.1,
// profiles/r03_*, tools/ubench_r3.hip, ubench_ifetch.hip, ubench_vmem.hip, tools/experiments/):
//   * Its time is the SUM of what its instructions cost at issue - vector 0.87 ns (4-byte encodings) /
//     1.06 ns (8-byte VOP3) / 1.8-1.9 ns (DPP, SDWA, v_cndmask with an SGPR mask, v_dot2c, conversions) per
//     wave instruction and SIMD, 6 ns per v_mfma_f32_4x4x1 (which also keeps the vector pipe of its SIMD
//     from issuing), LDS transfers, scalar and wait instructions - leaving phases out removes their share
//     and nothing more, 6 waves per SIMD run as fast as 4, and the kernel with every phase left out is
//     the 0.40 ms its loads and stores take.  Fewer and cheaper instructions are the only lever.
//   * Hence: the set index lives in scalar registers (wid through v_readfirstlane: the 64-bit address
//     arithmetic leaves the vector pipe); the MFMA chain is as long as the widest part of the bank plan
//     (28 instructions for 40 bins, 24 for 23, instead of a padded 32) with every operand read issued
//     before its first instruction; and the rows leave through ONE unconditional buffer store per set:
//     with the stores inside `if`s the compiler cannot count the vector-memory operations behind the
//     prefetched samples, waits with vmcnt(0) at the top of every iteration, and the wave sits there
//     until the stores of the previous set are acknowledged (1300-2000 of 11 000 clocks per iteration).
//   * Measured and dropped (numbers in profiles/NOTEBOOK.md): typed buffer loads that convert int16 -> float in the
//     texture path (9-11 ns per instruction and CU against 3.8 ns for global_load_dword: 26 of them are a
//     0.70 ms floor of their own), the 16 x 16 exchange through MFMA transposes + v_permlane swaps instead
//     of LDS (2.5 x slower), a split real / imaginary exchange tile for 6-8 waves per SIMD (no gain; the
//     compiler spills at 64 registers), L2 touch-ahead loads, compiler-scheduled LDS waits.
//
// Restates the [KALDI-UPSTREAM] per-frame recipe (feature-window.cc ProcessWindow, feature-fbank.cc,
// feature-mfcc.cc, MelBanks::Compute), reached by the reference at shennong/processor/base.py:429-431.
#include <float.h>

#include <cstdlib>

#include "snf_internal.h"
#include "device_fft.h"

namespace snf {

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 16;          // one 16-wave workgroup per CU (4 waves per SIMD)
constexpr int kHeaderFloats = 16;   // (= kFastHeaderFloats of kernels_fbank512.hip: same table blob)
constexpr int kTileRow = 17;        // complex per row of the exchange tile (16 + 1 pad: conflict-free)
constexpr int kTileBytes = 16 * kTileRow * 8;  // wave-private LDS per frame (2176 B)


}  // namespace

// ENERGY: 0 = no log-energy column, 1 = raw (before pre-emphasis / window), 2 = after the window.
// BST: the rows of a set are whole groups of 4 values without an energy column (fbank with num_bins % 4
// == 0, MFCC): one unconditional buffer store per set (see above); otherwise the stores of fbank512_kernel.
// Round 4: the butterflies carry their twiddles (device_fft.h, Linzer-Feig form): the second radix-4 layer of
// both register passes, the inter-pass twiddle (now behind the transpose, in the first butterflies of pass
// 2) and the twiddle of the real-FFT unpack; the tables hold (cos, tan) pairs.  -52 of 722 vector
// instructions per frame set, -1.9 % time (profiles/NOTEBOOK.md 4.1c).  fbank512_kernel uses the same arithmetic.
// DITHER (round 4): Kaldi's per-window dither, N(0, dither^2) added to every sample before the DC removal -
// the reference's default (dither = 1.0, shennong/processor/base.py:122).  The stream is the one of
// fbank512_kernel (same key per frame, same generator: bit-identical features from either kernel); the key
// of a frame comes from a table made once per call (launch_build_frame_noise) and is prefetched with the
// frame's first-sample index.
template <int NJ, int KIND, int ENERGY, bool BST, bool DITHER>
__global__ __launch_bounds__(kWaves * 64, 4) void fbank512b_kernel(const Fast512Params p, const BatchArgs b,
                                                                  float* __restrict__ out,
                                                                  double* __restrict__ energy_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem);
  // ---- stage the tables into LDS (the only workgroup-wide barrier of the kernel) -------------------
  for (int i = threadIdx.x; i < p.table_floats; i += kWaves * 64) tab[i] = p.tables[i];
  __syncthreads();
  const float2* __restrict__ t_win = reinterpret_cast<const float2*>(tab + kHeaderFloats);
  const float2* __restrict__ t_tw16 = t_win + 16 * 18;
  const float2* __restrict__ t_tw512 = t_tw16 + 16 * 18;
  const float4* __restrict__ t_mm_a = reinterpret_cast<const float4*>(tab + p.off_mm_a);
  const float* __restrict__ t_lifter = tab + p.off_lifter;
  const float4* __restrict__ t_dd_v = reinterpret_cast<const float4*>(tab + p.off_dd_v);

  // (wid through readfirstlane: the set index and everything derived from it stay in scalar registers)
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l = lane & 15, q = lane >> 4;
  const int tab_bytes = (p.table_floats * 4 + 255) & ~255;
  char* wave_base = smem + tab_bytes + (wid * 4 + q) * kTileBytes;
  float2* tile = reinterpret_cast<float2*>(wave_base);  // 16 rows x 17 complex
  // power tile [257]: aliases the frame tile.  The frame tiles are 544 floats apart (bank offset 0, 32, 0,
  // 32): a skew of 16 q floats puts the 16-lane runs of the four frames of a 4-byte access on four
  // disjoint bank windows
  float* ptile = reinterpret_cast<float*>(wave_base) + q * 16;
  // the padding column of the frame tile is never written by the transposes and the mel phase reads a few
  // floats past the power tile with zero weights: LDS keeps what the previous kernel left there (see
  // fbank512_kernel).  Zero it once.
  tile[l * kTileRow + 16] = make_float2(0.0f, 0.0f);
  // MFMA view of the wave: lane = 4 b + j, block b, frame j of the set.  Where the lane's B operands start,
  // the first of the 4 bins it stores and the 0 / 1 factors of the partial sums of a split group are read
  // from the table where they are used (five registers less through the transform).
  const int mj = lane & 3;
  const float* __restrict__ mm_lane = tab + p.off_mm_lane + lane;
  const float* __restrict__ mtile =
      reinterpret_cast<const float*>(smem + tab_bytes + (wid * 4 + mj) * kTileBytes) + mj * 16;

  const float win_len_f = static_cast<float>(p.win_len), inv_win_len = 1.0f / win_len_f;
  const int64_t n_sets = (b.total_frames + 3) >> 2;
  const int64_t set_stride = static_cast<int64_t>(gridDim.x) * kWaves;
  const int64_t last_frame = b.total_frames - 1;
  int64_t set = static_cast<int64_t>(blockIdx.x) * kWaves + wid;
  if (set >= n_sets) return;

  // NJ = 13: the 25 ms / 16 kHz window (only element j = 12 can fall outside it); NJ = 16: any other
  // window that pads to 512 samples, with a per-element test
  const bool in_last = 2 * (l + 16 * (NJ - 1)) < p.win_len;
  auto in_window = [&](int j) -> bool {
    if (NJ == 13) return j < NJ - 1 || in_last;
    return 2 * (l + 16 * j) < p.win_len;
  };
  auto start_of = [&](int64_t s) -> int64_t {  // first sample of the lane's frame of set s (clamped)
    const int64_t gi = s * 4 + q;
    return b.frame_start[gi < last_frame ? gi : last_frame];
  };
  auto noise_of = [&](int64_t s) -> unsigned long long {  // ... and its noise key
    const int64_t gi = s * 4 + q;
    return DITHER ? b.frame_noise[gi < last_frame ? gi : last_frame] : 0ull;
  };
  // Software pipeline over frame sets: one dword (two int16 samples) per element, requested a whole
  // iteration before it is converted; the start offset of the set after that arrives meanwhile.
  typedef int __attribute__((aligned(2))) int_a2;
  int raw[NJ];
  auto request = [&](int64_t st) {
    const int16_t* __restrict__ wp = b.wave + st;
    const int16_t* __restrict__ wl = wp + 2 * l;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      raw[j] = *reinterpret_cast<const int_a2*>((NJ == 13 && j < NJ - 1) || in_window(j) ? wl + 32 * j : wp);
  };
  // the unpack twiddles stay in registers for the whole kernel: the kernel uses 103 of the 128 registers
  // 4 waves per SIMD leave it, and LDS bandwidth is what it is short of (4 ds_read_b128 per set less)
  float4 w512q[4];
  read_quads<4>(t_tw512 + l * 10, w512q);
  int64_t start_next = start_of(set + set_stride);
  unsigned long long noise_cur = noise_of(set), noise_next = noise_of(set + set_stride);
  request(start_of(set));
  if (BST) {
    // one dropped store behind the first request: the loop is entered with the same sequence of vector-
    // memory operations in flight as its back edge carries (loads, then one store), so the waits at the
    // top of the body are counted (vmcnt(13) .. vmcnt(1)) instead of draining everything
    const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000);
    if (KIND == SNF_KIND_MFCC) __builtin_amdgcn_raw_buffer_store_b32(0u, none, -1, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, none, -1, 0, 2);
  }

  for (; set < n_sets; set += set_stride) {
    const int64_t g = set * 4 + q;  // global frame = output row
    const bool valid = g <= last_frame;

    // ---- A: DC removal, pre-emphasis, window (Kaldi's ProcessWindow order) -------------------------
    float xe[NJ], xo[NJ];
    // the samples are integers whose sums stay below 2^24: the float32 sum Kaldi forms is exact in any
    // order, so v_dot2c_i32_i16 adds both halves of a dword in one instruction (not with dither: the sum
    // of the dithered samples is formed like fbank512_kernel forms it)
    int part_i = 0;
    float part = 0.0f;
    unsigned dkey_lo = 0, dkey_hi = 0;
    if (DITHER) {
      const unsigned long long k = noise_cur ^ p.seed;
      dkey_lo = fmix32(static_cast<unsigned>(k));
      dkey_hi = fmix32(static_cast<unsigned>(k >> 32) ^ dkey_lo);
      // (finished values: without this the last xor-shift of the keys is redone for every element)
      asm volatile("" : "+v"(dkey_lo), "+v"(dkey_hi));
    }
    const float dscale = DITHER ? dither_scale(p.dither) : 0.0f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      xe[j] = static_cast<float>(static_cast<short>(raw[j] & 0xffff));
      xo[j] = static_cast<float>(raw[j] >> 16);
      if (DITHER) {
        add_dither_pair(dkey_lo, dkey_hi, static_cast<unsigned>(l + 16 * j), dscale, xe[j], xo[j]);
        const float s2 = xe[j] + xo[j];
        part += in_window(j) ? s2 : 0.0f;
      } else {
        const int both = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, raw[j]), short2v{1, 1}, part_i, false);
        part_i = in_window(j) ? both : part_i;
      }
    }
    if (!DITHER) part = static_cast<float>(part_i);
    // the conversions are the last readers of `raw`: pin them so that the loads of the next set reuse the
    // same registers
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(xe[j]), "+v"(xo[j]) : : "memory");
    asm volatile("" : "+v"(part) : : "memory");
    request(start_next);  // (the final request of a wave re-reads the last frame)
    start_next = start_of(set + 2 * set_stride);
    if (DITHER) {
      noise_cur = noise_next;
      noise_next = noise_of(set + 2 * set_stride);
    }

    float neg_mean = 0.0f;
    if (p.remove_dc) {
      const float sum = row_sum16(part);
      // sum / N correctly rounded without the IEEE division sequence (see fbank512_kernel)
      const float qv = sum * inv_win_len;
      neg_mean = -__builtin_fmaf(__builtin_fmaf(-qv, win_len_f, sum), inv_win_len, qv);
      if (DITHER) neg_mean = -sum / win_len_f;  // (not an integer sum: the division, like fbank512_kernel)
    }
    float2 z[16];
    float e_raw = 0.0f, e_post = 0.0f;
    float rot_prev = xe[0] + neg_mean;  // lane 0, j = 0: x[-1] := x[0] (Kaldi Preemphasize)
#pragma unroll
    for (int jj = 0; jj < 16; jj += 2) {
      float4 w4;
      if (jj < NJ)  // (zero outside the window; rows are 16-byte aligned)
        w4 = *reinterpret_cast<const float4*>(__builtin_assume_aligned(t_win + l * 18 + jj, 16));
#pragma unroll
      for (int j = jj; j < jj + 2; ++j) {
        if (j < NJ) {
          const float ae = xe[j] + neg_mean, ao = xo[j] + neg_mean;
          // the left neighbour x[2n-1] is the odd sample of lane l - 1 (same j), or lane 15 of j - 1
          const float rot = dpp_row_ror<0x121>(ao);
          const float ap = l == 0 ? rot_prev : rot;
          rot_prev = rot;
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
    __builtin_amdgcn_sched_barrier(0);

    // ---- B: pass 1 (FFT over j), inter-pass twiddle W256^(l k2), 16 x 16 transpose -----------------
    fft16_lf(z);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) tile[k2 * kTileRow + l] = z[k2];
    wave_lds_order();
    // the inter-pass twiddle W256^(n l) of element n of this lane's row (the table is symmetric in n and
    // l) as (cos, tan) pairs: read while the tile lands
    float2 ct[16];
    {
      float4 tw4[8];
      read_tw8_row16(t_tw16 + l * 18, tile + l * kTileRow, tw4, z);  // (cos, tan) pairs + the transposed row
#pragma unroll
      for (int m = 0; m < 16; m += 2) {
        ct[m] = make_float2(tw4[m >> 1].x, tw4[m >> 1].y);
        ct[m + 1] = make_float2(tw4[m >> 1].z, tw4[m >> 1].w);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- C: pass 2 (twiddle + FFT over n1): z[k1] = Z[l + 16 k1] -----------------------------------
    fft16_twin(z, ct);
    __builtin_amdgcn_sched_barrier(0);
    wave_lds_order();

    // ---- D: real-FFT unpack + power (x4): the partner Z[256 - k] of k = l + 16 k1 (k1 < 8) is
    // (16 - l) + 16 (15 - k1): the upper half of the spectrum goes through the tile, rows 0..7 ---------
#pragma unroll
    for (int r = 0; r < 8; ++r) tile[r * 16 + l] = z[r + 8];
    wave_lds_order();
    float pk[8], pm[8];  // 4 P[k], 4 P[256 - k]
    {
      float2 zpart[8];
      read8_b64_rev128(tile + (16 - l), zpart);  // zpart[k1] = Z[256 - l - 16 k1]
      // (W512^(l + 16 k1) as (cos, tan) pairs: 16 registers read once before the loop, see below)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("" : "+v"(w512q[i].x), "+v"(w512q[i].y), "+v"(w512q[i].z), "+v"(w512q[i].w));
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) {
        const float2 zk = z[k1];
        const float2 zp = zpart[k1];
        const float2 w = (k1 & 1) ? make_float2(w512q[k1 >> 1].z, w512q[k1 >> 1].w)
                                  : make_float2(w512q[k1 >> 1].x, w512q[k1 >> 1].y);
        const float c_re = zk.x + zp.x, c_im = zk.y - zp.y;
        const float d_re = zk.y + zp.y, d_im = zp.x - zk.x;
        // w = (cos, tan): d w = cos * u, u = d (1 + i tan); the scale rides on the butterfly
        const float u_re = __builtin_fmaf(-w.y, d_im, d_re), u_im = __builtin_fmaf(w.y, d_re, d_im);
        const float a_re = __builtin_fmaf(w.x, u_re, c_re), a_im = __builtin_fmaf(w.x, u_im, c_im);
        const float b_re = __builtin_fmaf(-w.x, u_re, c_re), b_im = __builtin_fmaf(w.x, u_im, -c_im);
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
    wave_lds_order();
    // ---- E: power tile ------------------------------------------------------------------------------
    {
      // two rows per instruction (ds_write2_b32: 3 LDS-path clocks per dword against 4 for ds_write_b32)
      const unsigned pk_addr = lds_addr(ptile + l), pm_addr = lds_addr(ptile + (144 - l));
#define SNF_W2(ADDR_, A_, B_, O0_, O1_) \
  asm volatile("ds_write2_b32 %0, %1, %2 offset0:" #O0_ " offset1:" #O1_ : : "v"(ADDR_), "v"(A_), "v"(B_) : "memory")
      SNF_W2(pk_addr, pk[0], pk[1], 0, 16);
      SNF_W2(pk_addr, pk[2], pk[3], 32, 48);
      SNF_W2(pk_addr, pk[4], pk[5], 64, 80);
      SNF_W2(pk_addr, pk[6], pk[7], 96, 112);
      SNF_W2(pm_addr, pm[7], pm[6], 0, 16);  // index 256 - l - 16 k1 = (144 - l) + 16 (7 - k1)
      SNF_W2(pm_addr, pm[5], pm[4], 32, 48);
      SNF_W2(pm_addr, pm[3], pm[2], 64, 80);
      SNF_W2(pm_addr, pm[1], pm[0], 96, 112);
#undef SNF_W2
      if (l == 0) ptile[128] = p128;
    }
    wave_lds_order();
    __builtin_amdgcn_sched_barrier(0);

    // ---- log-energy column ---------------------------------------------------------------------------
    float log_energy = 0.0f;
    if (ENERGY != 0) {
      if (KIND == SNF_KIND_PLP) {
        if (valid && l == 0) energy_out[g] = static_cast<double>(e_lin);  // (plp_tail_kernel takes the double log)
      } else {
        log_energy = fast_log(floor_eps(e_lin));
        if (p.has_floor && log_energy < p.log_energy_floor) log_energy = p.log_energy_floor;
      }
    }
    float* __restrict__ row = out + g * static_cast<int64_t>(p.out_cols);  // (stores of the non-BST forms)
    // the rows of this set as a buffer: rows past the last frame fall outside it (BST)
    const int64_t rows_left = b.total_frames - set * 4;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
        out + set * 4 * static_cast<int64_t>(p.out_cols), 0,
        (rows_left < 4 ? static_cast<int>(rows_left) : 4) * p.out_cols * 4, 0x00020000);
    {
      // ---- F: mel filterbank on the matrix pipe (see fbank512_kernel) -------------------------------
      const bool mvalid = set * 4 + mj <= last_frame;
      const int mm_start = reinterpret_cast<const int*>(mm_lane)[0];
      const float4* __restrict__ bsrc = reinterpret_cast<const float4*>(mtile + mm_start);
      const float4* __restrict__ asrc = t_mm_a + lane;
      f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#define SNF_QUAD(A_, X_)                                                   \
  do {                                                                     \
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A_.x, X_.x, acc0, 0, 0, 0);   \
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A_.y, X_.y, acc1, 0, 0, 0);   \
    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(A_.z, X_.z, acc0, 0, 0, 0);   \
    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(A_.w, X_.w, acc1, 0, 0, 0);   \
  } while (0)
      const int n_quads = p.mm_quads;  // wave-uniform
      if (n_quads == 7 || n_quads == 6) {
        // the common shapes (40 / 23 bins at 16 kHz): every operand read is issued before the first
        // instruction of the chain, which then runs at the pace of the matrix pipe
        float4 a[7], x[7];
#pragma unroll
        for (int t = 0; t < 7; ++t) {
          a[t] = asrc[t * 64];  // (row 6 of a six-quad table: zeros, read and not issued)
          x[t] = bsrc[t];
        }
#pragma unroll
        for (int t = 0; t < 7; ++t)
          if (t < 6 || n_quads == 7) SNF_QUAD(a[t], x[t]);
      } else {
        // any other chain length: two operand sets in flight; the weight table ends with two rows of
        // zeros, so the look-ahead reads need no test (the B side reads finite tile contents)
        float4 a0 = asrc[0], x0 = bsrc[0];
        int t = 0;
        for (; t + 1 < n_quads; t += 2) {
          const float4 a1 = asrc[(t + 1) * 64], x1 = bsrc[t + 1];
          SNF_QUAD(a0, x0);
          a0 = asrc[(t + 2) * 64];
          x0 = bsrc[t + 2];
          SNF_QUAD(a1, x1);
        }
        if (t < n_quads) SNF_QUAD(a0, x0);  // an odd quad on its own
      }
#undef SNF_QUAD
      float mel[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) mel[i] = acc0[i] + acc1[i];
      const int mm_out = reinterpret_cast<const int*>(mm_lane)[64];
      // a wide group is split over up to 4 neighbouring blocks of one 16-lane row: the first block adds
      // the sums of the others (0 / 1 factors per lane; one v_fmac_f32 with a DPP row shift each)
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
          // lane 4 b + j: 4 consecutive bins of frame j; a lane without a group stores outside the buffer
          __builtin_amdgcn_raw_buffer_store_b128(
              __builtin_bit_cast(u32x4, f32x4{mel[0], mel[1], mel[2], mel[3]}), orsrc,
              mm_out >= 0 ? (mj * p.out_cols + mm_out) * 4 : -1, 0, 2);
        } else {
          if (mvalid && mm_out >= 0) {
            float* __restrict__ dst = out + (set * 4 + mj) * static_cast<int64_t>(p.out_cols) + mel_col + mm_out;
            if (mm_out + 4 <= p.num_bins) {
              __builtin_nontemporal_store(f32x4_a4{mel[0], mel[1], mel[2], mel[3]}, reinterpret_cast<f32x4_a4*>(dst));
            } else {
#pragma unroll
              for (int i = 0; i < 3; ++i)
                if (mm_out + i < p.num_bins) dst[i] = mel[i];
            }
          }
          if (KIND == SNF_KIND_FBANK && p.use_energy && valid && l == 0)
            row[p.htk_compat ? p.num_bins : 0] = log_energy;
        }
      }
      if (KIND == SNF_KIND_MFCC) {
        // log-mel of frame j back to its (now idle) power tile; DCT-II + lifter on the vector pipe: lane l
        // of a frame's row owns cepstrum l and walks the log-mel 4 bins at a time
        wave_lds_order();
        if (mm_out >= 0)
          *reinterpret_cast<float4*>(const_cast<float*>(mtile) + mm_out) =
              make_float4(fast_log(floor_eps(mel[0])), fast_log(floor_eps(mel[1])),
                          fast_log(floor_eps(mel[2])), fast_log(floor_eps(mel[3])));
        wave_lds_order();
        const float4* __restrict__ dw = t_dd_v + l;
        const float4* __restrict__ dx = reinterpret_cast<const float4*>(ptile);
        float v = 0.0f;
        if (p.dd_groups == 6) {
          // 21-24 mel bins (the reference's default 23): straight-line, the twelve operand reads issued
          // before the first multiply-add (same products in the same order as the loop below)
          float4 w[6], x[6];
#pragma unroll
          for (int g4 = 0; g4 < 6; ++g4) {
            w[g4] = dw[g4 * 16];
            x[g4] = dx[g4];
          }
#pragma unroll
          for (int g4 = 0; g4 < 6; ++g4) {
            v += w[g4].x * x[g4].x;
            v += w[g4].y * x[g4].y;
            v += w[g4].z * x[g4].z;
            v += w[g4].w * x[g4].w;
          }
        } else {
#pragma unroll 2
          for (int g4 = 0; g4 < p.dd_groups; ++g4) {
            const float4 w = dw[g4 * 16], x = dx[g4];
            v += w.x * x.x;
            v += w.y * x.y;
            v += w.z * x.z;
            v += w.w * x.w;
          }
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
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orsrc,
                                                l < p.num_ceps ? (q * p.out_cols + oc) * 4 : -1, 0, 0);
        } else if (valid && l < p.num_ceps) {
          row[oc] = v;
        }
      }
    }
    wave_lds_order();  // the tile is reused by the next frame set
  }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
// Flat batches (no per-utterance warps, no fused deltas), snip_edges, vector-pipe DCT; dither when the caller
// made the frames' key table (capi.hip).
// SNF_FBANK512_OLD=1 keeps every batch on fbank512_kernel (A/B runs: tools/ab_fbank512.cpp, bench.py).
bool fbank512b_eligible(const Fast512Params& p, const BatchArgs& b) {
  if (const char* knob = getenv("SNF_FBANK512_OLD"))
    if (knob[0] == '1') return false;
  if (p.dual || p.fused_delta || b.blk_utt != nullptr) return false;
  if (!p.snip_edges || p.dct_mfma) return false;
  if (p.dither != 0.0f && b.frame_noise == nullptr) return false;  // (no key table: fbank512_kernel draws its own)
  // (the spectrogram's four 16-byte row stores per lane measured 7 % slower here than on fbank512_kernel)
  if (p.kind != SNF_KIND_FBANK && p.kind != SNF_KIND_MFCC && p.kind != SNF_KIND_PLP) return false;
  return static_cast<size_t>((p.table_floats * 4 + 255) & ~255) + kWaves * 4 * kTileBytes <= 160 * 1024;
}

namespace {

template <int NJ, int KIND, int ENERGY, bool BST, bool DITHER>
int launch_dither(const Fast512Params& q, const BatchArgs& b, float* out, double* energy_out, hipStream_t stream) {
  const size_t lds = static_cast<size_t>((q.table_floats * 4 + 255) & ~255) + kWaves * 4 * kTileBytes;
  const int64_t n_sets = (b.total_frames + 3) / 4;
  int64_t blocks = (n_sets + kWaves - 1) / kWaves;
  if (blocks > 256) blocks = 256;  // one resident workgroup per CU, grid-stride over the frame sets
  auto kern = fbank512b_kernel<NJ, KIND, ENERGY, BST, DITHER>;
  if (lds > 64 * 1024)
    SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kWaves * 64), lds, stream, q, b, out,
                     energy_out);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

template <int NJ, int KIND, int ENERGY, bool BST>
int launch_one(const Fast512Params& q, const BatchArgs& b, float* out, double* energy_out, hipStream_t stream) {
  if (q.dither != 0.0f) return launch_dither<NJ, KIND, ENERGY, BST, true>(q, b, out, energy_out, stream);
  return launch_dither<NJ, KIND, ENERGY, BST, false>(q, b, out, energy_out, stream);
}

template <int NJ, int KIND>
int launch_energy(const Fast512Params& q, const BatchArgs& b, float* out, double* energy_out, hipStream_t stream) {
  const int energy = q.need_raw ? 1 : (q.need_post ? 2 : 0);
  // one buffer store per set: whole groups of 4 bins and no energy column beside them / one cepstrum per lane
  constexpr bool kCanBst = KIND == SNF_KIND_FBANK || KIND == SNF_KIND_MFCC;
  const bool bst = KIND == SNF_KIND_MFCC || (KIND == SNF_KIND_FBANK && q.num_bins % 4 == 0 && !q.use_energy);
  if (energy == 0) {
    if (kCanBst && bst) return launch_one<NJ, KIND, 0, kCanBst>(q, b, out, energy_out, stream);
    return launch_one<NJ, KIND, 0, false>(q, b, out, energy_out, stream);
  }
  if (energy == 1) {
    if (kCanBst && bst) return launch_one<NJ, KIND, 1, kCanBst>(q, b, out, energy_out, stream);
    return launch_one<NJ, KIND, 1, false>(q, b, out, energy_out, stream);
  }
  if (kCanBst && bst) return launch_one<NJ, KIND, 2, kCanBst>(q, b, out, energy_out, stream);
  return launch_one<NJ, KIND, 2, false>(q, b, out, energy_out, stream);
}

template <int NJ>
int launch_kind(const Fast512Params& q, const BatchArgs& b, float* out, double* energy_out, hipStream_t stream) {
  if (q.kind == SNF_KIND_FBANK) return launch_energy<NJ, SNF_KIND_FBANK>(q, b, out, energy_out, stream);
  if (q.kind == SNF_KIND_MFCC) return launch_energy<NJ, SNF_KIND_MFCC>(q, b, out, energy_out, stream);
  return launch_energy<NJ, SNF_KIND_PLP>(q, b, out, energy_out, stream);
}

}  // namespace

namespace {
__global__ void build_frame_noise_kernel(const BatchArgs b, uint64_t* __restrict__ keys) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= b.total_frames) return;
  const int64_t u = b.frame_utt[g];
  keys[g] = wave_noise_id(b, u, g - b.frame_offsets[u]);
}
}  // namespace

namespace {
__global__ void build_utt_noise_kernel(const BatchArgs b, uint32_t* __restrict__ words) {
  const int64_t u = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (u >= b.n_utts) return;
  const int64_t s0 = b.sample_offsets[u], n = b.sample_offsets[u + 1] - s0;
  uint32_t h = 0x811C9DC5u ^ static_cast<uint32_t>(n);
  const int taps = n < 64 ? static_cast<int>(n) : 64;
  for (int k = 0; k < taps; ++k) {
    // sample k of `taps`, the first and the last included; murmur-style mixing of one sample per step
    const int64_t i = taps > 1 ? (n - 1) * k / (taps - 1) : 0;
    uint32_t v = static_cast<uint16_t>(b.wave[s0 + i]);
    v *= 0xCC9E2D51u;
    v = (v << 15) | (v >> 17);
    v *= 0x1B873593u;
    h ^= v;
    h = ((h << 13) | (h >> 19)) * 5u + 0xE6546B64u;
  }
  words[u] = fmix32(h);
}
}  // namespace

int launch_build_utt_noise(const BatchArgs& b, uint32_t* d_words, hipStream_t stream) {
  if (b.n_utts <= 0) return SNF_OK;
  hipLaunchKernelGGL(build_utt_noise_kernel, dim3(static_cast<unsigned>((b.n_utts + 255) / 256)), dim3(256), 0,
                     stream, b, d_words);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

int launch_build_frame_noise(const BatchArgs& b, uint64_t* d_keys, hipStream_t stream) {
  if (b.total_frames <= 0) return SNF_OK;
  hipLaunchKernelGGL(build_frame_noise_kernel, dim3(static_cast<unsigned>((b.total_frames + 255) / 256)), dim3(256),
                     0, stream, b, d_keys);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

int launch_fbank512b(const Fast512Params& p, const BatchArgs& b, float* out, int out_cols, double* energy_out,
                     hipStream_t stream) {
  if (b.total_frames <= 0) return SNF_OK;
  Fast512Params q = p;
  q.out_cols = out_cols;
  if ((p.win_len + 31) / 32 == 13) return launch_kind<13>(q, b, out, energy_out, stream);
  return launch_kind<16>(q, b, out, energy_out, stream);
}

}  // namespace snf
