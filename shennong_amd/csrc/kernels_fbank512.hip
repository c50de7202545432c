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
//      (half a tile), twiddles W512^k from an LDS table.
//      Scaled by 4 (the 1/2 factors of the unpack are folded into the mel weights as an exact power
//      of two).
//   E  power spectrum to a wave-private LDS tile [frame][bin].
//   F  mel filterbank on the MATRIX pipe: v_mfma_f32_4x4x1_16b_f32 computes 16 independent blocks of
//      (4 mel bins) x (4 frames) x (1 FFT bin) per instruction - exactly the shape of a wave's frame
//      set.  Block b owns the group of 4 neighbouring mel bins 4 g .. 4 g + 3 (or a run of the FFT
//      bins of a wide group: partial sums of up to 4 blocks are added through DPP); per instruction t
//      lane 4 b + i supplies the weight of bin 4 g + i at FFT bin start_b + t (A operand, LDS table)
//      and lane 4 b + j the power of frame j at that bin (B operand, read as ds_read_b128 runs from
//      the power tile).  The band structure of the mel matrix is kept (~28 instructions for 40 bins
//      instead of 256 for the dense product), the arithmetic is exact f32 (a k-ordered fmaf chain),
//      and the VALU and most of the LDS traffic of the former sparse VALU form are gone: lane 4 b + j
//      ends up with 4 consecutive mel bins of frame j -> log -> ONE 16-byte store per lane.
//      MFCC adds the 13 x 23 DCT-II as a second chain on the same instruction (blocks = 4 cepstral
//      groups x 4 partitions of the mel bins) and the lifter.
// Nothing but the int16 samples and the float32 features ever touches HBM; no workgroup barrier.
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
#include "device_fft.h"

namespace snf {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kMaxWaves = 16;             // wavefronts per workgroup: 8 when two workgroups fit the LDS
                                          // of a CU (measured 5 % faster), else one of 16
constexpr int kTileRow = 17;               // complex per transposed row (16 + 1 pad: conflict-free)
constexpr int kFrameTileBytes = 16 * kTileRow * 8;  // wave-private LDS per frame (2176 B)
constexpr int kFastHeaderFloats = 16;          // table header: the mel layout of this warp factor
constexpr int kSetsPerBlock = 64;              // PERUTT: frame sets (of 4 frames) per workgroup
constexpr int kFusedWaves = 14;                // fused deltas: 14 waves x 6 sets = 82 sets + one halo set on
constexpr int kFusedSets = kFast512FusedSets;                 // each side (the +-4 frames the delta-delta reaches); a 3 s
                                               // utterance (75 sets) is one workgroup without any halo
constexpr int kFusedCols = 16;                 // cepstra per row of the LDS buffer (num_ceps <= 16)

}  // namespace

// ENERGY: 0 = no log-energy column, 1 = raw (before pre-emphasis/window), 2 = after the window
template <int NJ, int KIND, int ENERGY, bool DITHER, bool SNIP, int MODE>
__global__ __launch_bounds__(kMaxWaves * 64, 4) void fbank512_kernel(const Fast512Params p,
                                                                   const BatchArgs b,
                                                                   float* __restrict__ out,
                                                                   double* __restrict__ energy_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem);
  // MODE 0: flat (sets of 4 consecutive global frames, grid-stride).  MODE 1 / 2: PERUTT scheduling.
  // MODE 2 (MFCC only) additionally keeps the cepstra of the workgroup's frames in LDS and writes
  // [cepstra | delta | delta-delta] rows (BASELINE config 3: no [T, 13] round trip through HBM).
  constexpr bool PERUTT = MODE != 0, FUSED = MODE == 2;
  const int sets_per_block = FUSED ? kFusedSets : kSetsPerBlock;
  // PERUTT (utterances with VTLN warps): a workgroup works on a run of frame sets of ONE utterance
  // and stages the tables of that utterance's warp factor; its mel layout comes from the table header
  // instead of the kernel arguments.
  int64_t pu_u = 0, pu_f0 = 0, pu_T = 0, pu_s0 = 0, pu_n = 0;
  int pu_set0 = 0;
  const float* __restrict__ gtab = p.tables;
  int h_mm_quads = p.mm_quads, h_mm_levels = p.mm_levels, h_dd_quads = p.dd_quads,
      h_off_mm_a = p.off_mm_a, h_off_mm_lane = p.off_mm_lane, h_off_dd_a = p.off_dd_a,
      h_off_lifter = p.off_lifter, h_table_floats = p.table_floats, h_off_dd_v = p.off_dd_v,
      h_dd_groups = p.dd_groups;
  if (PERUTT) {
    pu_u = b.blk_utt[blockIdx.x];
    pu_set0 = b.blk_set0[blockIdx.x];
    pu_f0 = b.frame_offsets[pu_u];
    pu_T = b.frame_offsets[pu_u + 1] - pu_f0;
    pu_s0 = b.sample_offsets[pu_u];
    pu_n = b.sample_offsets[pu_u + 1] - pu_s0;
    gtab = p.tables + static_cast<int64_t>(b.utt_warp ? b.utt_warp[pu_u] : 0) * p.table_stride;
    const int* __restrict__ hdr = reinterpret_cast<const int*>(gtab);
    h_mm_quads = hdr[0];
    h_mm_levels = hdr[1];
    h_dd_quads = hdr[2];
    h_off_mm_a = hdr[3];
    h_off_mm_lane = hdr[4];
    h_off_dd_a = hdr[5];
    h_off_lifter = hdr[6];
    h_table_floats = hdr[7];
    h_off_dd_v = hdr[8];
    h_dd_groups = hdr[9];
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
  const float4* __restrict__ t_mm_a = reinterpret_cast<const float4*>(tab + h_off_mm_a);
  const float4* __restrict__ t_dd_a = reinterpret_cast<const float4*>(tab + h_off_dd_a);
  const float* __restrict__ t_lifter = tab + h_off_lifter;
  const float4* __restrict__ t_dd_v = reinterpret_cast<const float4*>(tab + h_off_dd_v);

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
  // MFMA view of the wave: lane = 4 b + j, block b (a group of 4 mel bins, or a run of the FFT bins of
  // one), frame j of the set.  Lane-constant for the whole kernel: where the lane's B operands start
  // in the power tile of frame j, the first of the 4 output bins it stores (-1: none), and the 0/1
  // factors with which the partial sums of the next 1..3 blocks are added to its own.
  const int mj = lane & 3;
  const float* __restrict__ mm_lane = tab + h_off_mm_lane;
  const int mm_start = reinterpret_cast<const int*>(mm_lane)[lane];
  const int mm_out = reinterpret_cast<const int*>(mm_lane)[64 + lane];
  const float mm_f1 = mm_lane[128 + lane], mm_f2 = mm_lane[192 + lane], mm_f3 = mm_lane[256 + lane];
  const float* __restrict__ mtile =
      reinterpret_cast<const float*>(smem + tab_bytes + (wid * 4 + mj) * kFrameTileBytes) + mj * 16;

  const float win_len_f = static_cast<float>(p.win_len), inv_win_len = 1.0f / win_len_f;
  const int n_waves = blockDim.x >> 6;
  // flat mode: sets of 4 consecutive global frames, grid-stride.  PERUTT: sets of 4 consecutive frames
  // of the workgroup's utterance, kSetsPerBlock of them per workgroup.
  int64_t n_sets = (b.total_frames + 3) >> 2;
  int64_t set_stride = static_cast<int64_t>(gridDim.x) * n_waves;
  if (PERUTT) {
    const int64_t utt_sets = (pu_T + 3) >> 2;
    const int64_t end = pu_set0 + sets_per_block + (FUSED ? 1 : 0);  // (+ the halo set behind)
    n_sets = end < utt_sets ? end : utt_sets;
    set_stride = n_waves;
  }
  // fused deltas: cepstra of the local frames [4 pu_set0 - 4, 4 pu_set0 + 4 kFusedSets + 4)
  float* cepbuf = reinterpret_cast<float*>(smem + tab_bytes + n_waves * 4 * kFrameTileBytes);
  const int64_t cep_frame0 = static_cast<int64_t>(pu_set0) * 4 - 4;
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
  int64_t set = PERUTT ? pu_set0 - ((FUSED && pu_set0 > 0) ? 1 : 0) + wid
                       : static_cast<int64_t>(blockIdx.x) * n_waves + wid;
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
    if (KIND == SNF_KIND_SPECTROGRAM) {
      // five dropped stores behind the first request: the loop is entered with the same sequence of vector-
      // memory operations in flight as its back edge carries (loads, then the five stores of a set), so the
      // waits at the top of the body are counted instead of draining every store (see the stores below)
      const __amdgpu_buffer_rsrc_t none = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000);
#pragma unroll
      for (int i = 0; i < 5; ++i)   // (distinct offsets: identical stores would be merged into one)
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, none, -1 - 16 * i, 0, 2);
    }
  }
  // output row of the MFMA view (lane 4 b + j -> frame j of the set), advanced by a constant per set
  float* __restrict__ mrow =
      out + ((PERUTT ? pu_f0 : 0) + set * 4 + (lane & 3)) * static_cast<int64_t>(p.out_cols);
  const int64_t mrow_step = set_stride * 4 * static_cast<int64_t>(p.out_cols);
  for (; set < n_sets; set += set_stride, mrow += mrow_step) {
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
      // (keyed by the frame's place in its utterance, not in the batch: snf_internal.h)
      const int64_t du = PERUTT ? pu_u : static_cast<int64_t>(b.frame_utt[g < b.total_frames ? g : last_frame]);
      const unsigned long long k =
          wave_noise_id(b, du, PERUTT ? gl : g - b.frame_offsets[du]) ^ p.seed;
      dkey_lo = fmix32(static_cast<unsigned>(k));
      dkey_hi = fmix32(static_cast<unsigned>(k >> 32) ^ dkey_lo);
    }
    float xe[NJ], xo[NJ];
    float part = 0.0f;
    // Without dither (and away from reflected edges) the samples are integers whose sums stay below
    // 2^24: the float32 sum Kaldi forms is exact in any order, so v_dot2c_i32_i16 adds both halves of a
    // dword in one instruction and the result is bit-identical
    constexpr bool kIntSum = !DITHER && SNIP;
    int part_i = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      xe[j] = static_cast<float>(static_cast<short>(raw[j] & 0xffff));
      xo[j] = static_cast<float>(raw[j] >> 16);
      if (kIntSum) {
        const int both = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, raw[j]), short2v{1, 1}, part_i, false);
        part_i = in_window(j) ? both : part_i;
      }
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
        add_dither_pair(dkey_lo, dkey_hi, static_cast<unsigned>(l + 16 * j), dither_scale(p.dither), xe[j], xo[j]);
      }
      if (!kIntSum) {
        const float s2 = xe[j] + xo[j];
        part += in_window(j) ? s2 : 0.0f;
      }
    }
    if (kIntSum) part = static_cast<float>(part_i);
    // the conversions above are the last readers of `raw`: pin them (and the memory order) here so that
    // the loads of the next set below reuse the same registers instead of being hoisted into fresh ones
    // that have to be copied at the loop edge
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(xe[j]), "+v"(xo[j]) : : "memory");
    asm volatile("" : "+v"(part) : : "memory");
    // prefetch: samples of the next set (its start offset arrived during the previous iteration),
    // start offset of the set after it
    {  // (unconditional: start_of clamps to the last frame, so a wave's final prefetch re-reads it)
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
    if (p.remove_dc) {
      const float sum = row_sum16(part);
      if (kIntSum) {
        // sum / N, correctly rounded, without the 12-instruction IEEE division: q = sum * RN(1 / N)
        // plus one exact-residual correction equals RN(sum / N) for every integer |sum| < 2^24
        // (checked exhaustively for the window lengths in use; see tests/test_host_api.py)
        const float qv = sum * inv_win_len;
        neg_mean = -__builtin_fmaf(__builtin_fmaf(-qv, win_len_f, sum), inv_win_len, qv);
      } else {
        neg_mean = -sum / win_len_f;
      }
    }
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
    // (round 4: the butterflies carry their twiddles, device_fft.h; the inter-pass twiddle W256^(n l) sits
    // behind the transpose, in the first butterflies of pass 2 - the table is symmetric in n and l)
    fft16_lf(z);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) tile[k2 * kTileRow + l] = z[k2];
    wave_lds_sync();
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
    wave_lds_sync();

    // ---- D: real-FFT unpack + power (x4) -----------------------------------------------------------
    // partner of k = l + 16 k1 (k1 < 8) is 256 - k = (16 - l) + 16 (15 - k1): the upper half of the
    // spectrum goes through the tile, xbuf[r][c] = Z[c + 16 (r + 8)], rows 0..7 (+ row 8 scratch for
    // lane 0).  LDS instructions are free at the margin here (the kernel is bound by the vector pipe),
    // and two DPP moves per dword - which run at half the rate of plain VALU instructions - are not:
    // the exchange through DPP (row_mirror + row_shr:1) was measured and costs 4 % more.
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 8; ++r) tile[r * 16 + l] = z[r + 8];
    wave_lds_sync();
    const float2* __restrict__ partner = tile + (16 - l);  // Z[256 - k]: row 7 - k1, column 16 - l
    float2 zpart[8];
    read8_b64_rev128(partner, zpart);           // zpart[k1] = Z[256 - l - 16 k1]
    float4 w512q[4];
    read_quads<4>(t_tw512 + l * 10, w512q);     // W512^(l + 16 k1)
    float pk[8], pm[8];  // 4 P[k], 4 P[256-k] for k = l + 16 k1
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
    wave_lds_sync();  // (the transposed reads of the tile are complete; keeps the compiler in order)
    // ---- E: power tile ------------------------------------------------------------------------------
    if (KIND != SNF_KIND_SPECTROGRAM) {
      float* __restrict__ pmirror = ptile + (144 - l);
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) {
        ptile[l + 16 * k1] = pk[k1];
        pmirror[16 * (7 - k1)] = pm[k1];  // index 256 - l - 16 k1
      }
      if (l == 0) ptile[128] = p128;
      wave_lds_sync();
    }

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

    float* __restrict__ row = out + g * static_cast<int64_t>(p.out_cols);
    if (KIND == SNF_KIND_SPECTROGRAM) {
      // The rows are dense (257 floats: the kernel takes spectrograms of 512-point frames only), so the four
      // rows of a set are 4 112 contiguous bytes - a row alone is 4-byte aligned (1 028 B) and a 16-byte store
      // to it splits.  The log power spectrum goes from the registers to the wave's LDS in that flat order
      // (the frame tiles are dead: 4 x 257 floats over the first two) and leaves as 257 quads, 1 KB per wave
      // instruction.  Five UNCONDITIONAL buffer stores per set: what lies behind the last row of the batch is
      // dropped by the range check (per dword), lanes 1-63 of the fifth point outside the buffer.  A store
      // under a branch costs more than the branch: loads and stores share one in-order counter, the compiler
      // cannot count stores it may not issue, and the wait for the next set's samples at the top of the loop
      // becomes a wait for every store of this set (vmcnt(0)): 1.22 ms row by row under `if (valid)`, 1.19 ms
      // with aligned quads under a bounds test, see profiles/NOTEBOOK.md 4.1 for this form.
      float* __restrict__ stage = reinterpret_cast<float*>(smem + tab_bytes + wid * 4 * kFrameTileBytes);
      float* __restrict__ mine = stage + q * 257;
#pragma unroll
      for (int k1 = 0; k1 < 8; ++k1) {
        float lo_bin = fast_log(fmaxf(0.25f * pk[k1], FLT_EPSILON));
        if (k1 == 0 && l == 0) lo_bin = log_energy;   // bin 0 = energy
        mine[l + 16 * k1] = lo_bin;
        mine[256 - l - 16 * k1] = fast_log(fmaxf(0.25f * pm[k1], FLT_EPSILON));
      }
      if (l == 0) mine[128] = fast_log(fmaxf(0.25f * p128, FLT_EPSILON));
      wave_lds_sync();
      // (the set index is the same in every lane: through v_readfirstlane, or the descriptor lives in vector
      // registers and every store becomes a loop over its distinct values)
      const int64_t set_u = (static_cast<int64_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(set >> 32))) << 32) |
                            static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(set)));
      const int64_t rows_left = b.total_frames - set_u * 4;
      const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(
          out + set_u * 1028, 0, (rows_left < 4 ? static_cast<int>(rows_left) : 4) * 1028, 0x00020000);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = *reinterpret_cast<const float4*>(stage + 4 * (lane + 64 * i));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v.x, v.y, v.z, v.w}), srsrc,
                                               16 * (lane + 64 * i), 0, 2);
      }
      {
        const float4 v = *reinterpret_cast<const float4*>(stage + 1024);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v.x, v.y, v.z, v.w}), srsrc,
                                               lane == 0 ? 4096 : 0x7ffffff0, 0, 2);
      }
    } else {
      // ---- F: mel filterbank on the matrix pipe -----------------------------------------------------
      // lane 4 b + j: frame j of the set (MFMA view), its row in the output
      const int64_t mgl = set * 4 + mj;
      const bool mvalid = mgl <= last_frame;
      const float4* __restrict__ bsrc = reinterpret_cast<const float4*>(mtile + mm_start);
      f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
      const float4* __restrict__ asrc = t_mm_a + lane;
      if (h_mm_quads == 8) {
        // the common shapes (40 or 23 bins at 16 kHz: 32 taps per block): every operand read is issued
        // before the first instruction of the chain, which then runs at the pace of the matrix pipe
        float4 a[8], x[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          a[t] = asrc[t * 64];
          x[t] = bsrc[t];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t].x, x[t].x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t].y, x[t].y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t].z, x[t].z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t].w, x[t].w, acc1, 0, 0, 0);
        }
      } else {
        // any other chain length: two operand sets in flight.  mm_quads is even and the weight table
        // ends with a row of zeros, so the last look-ahead read needs no test (the B side reads finite
        // tile contents).
        float4 a0 = asrc[0], x0 = bsrc[0];
        for (int t = 0; t < h_mm_quads; t += 2) {  // wave-uniform trip count
          const float4 a1 = asrc[(t + 1) * 64], x1 = bsrc[t + 1];
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, x0.x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, x0.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, x0.z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, x0.w, acc1, 0, 0, 0);
          a0 = asrc[(t + 2) * 64];
          x0 = bsrc[t + 2];
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, x1.x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, x1.y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.z, x1.z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.w, x1.w, acc1, 0, 0, 0);
        }
      }
      float mel[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) mel[i] = acc0[i] + acc1[i];
      // a wide group is split over up to 4 neighbouring blocks of one 16-lane row: the first block adds
      // the sums of the others (0 / 1 factors per lane; one v_fmac_f32 with a DPP row shift each)
      if (h_mm_levels > 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float own = mel[i];
          fmac_row_shl<4>(mel[i], own, mm_f1);
          fmac_row_shl<8>(mel[i], own, mm_f2);
          if (h_mm_levels > 3) fmac_row_shl<12>(mel[i], own, mm_f3);
        }
      }
      const int mel_col = (KIND == SNF_KIND_FBANK && p.use_energy && !p.htk_compat) ? 1 : 0;
      if (KIND == SNF_KIND_FBANK || KIND == SNF_KIND_PLP) {
        if (KIND == SNF_KIND_FBANK && p.use_log) {
#pragma unroll
          for (int i = 0; i < 4; ++i) mel[i] = fast_log(floor_eps(mel[i]));
        }
        if (mvalid && mm_out >= 0) {
          float* __restrict__ dst = mrow + mel_col + mm_out;
          if (mm_out + 4 <= p.num_bins) {
            // (nontemporal: the rows are written once, 16 bytes per lane, whole 64-byte runs per quad)
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
      if (KIND == SNF_KIND_MFCC) {
        // log-mel of frame j back to its (now idle) power tile, then the DCT-II as a second chain:
        // block b = 4 cg + kp owns cepstra 4 cg .. 4 cg + 3 and the mel bins of partition kp
        wave_lds_sync();
        if (mm_out >= 0)
          *reinterpret_cast<float4*>(const_cast<float*>(mtile) + mm_out) =
              make_float4(fast_log(floor_eps(mel[0])), fast_log(floor_eps(mel[1])),
                          fast_log(floor_eps(mel[2])), fast_log(floor_eps(mel[3])));
        wave_lds_sync();
        if (!p.dct_mfma) {
          // DCT-II + lifter on the vector pipe (the shipped form): lane l of a frame's row owns cepstrum l
          // (num_ceps <= 16) and walks the log-mel of its frame 4 bins at a time (one broadcast 16-byte
          // read of the tile + one of the lane-major DCT table per group).  ~40 instructions and no
          // dependent matrix chain at the end of the set.  Measured a tie with the MFMA chain below on this
          // kernel (1.17-1.21 against 1.19-1.22 ms), a clear win on fbank256x2_kernel; SNF_DCT_MFMA=1
          // selects the chain.
          const float4* __restrict__ dw = t_dd_v + l;
          const float4* __restrict__ dx = reinterpret_cast<const float4*>(ptile);
          float v = 0.0f;
#pragma unroll 2
          for (int g4 = 0; g4 < h_dd_groups; ++g4) {
            const float4 w = dw[g4 * 16], x = dx[g4];
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
          if (valid && l < p.num_ceps) {
            if (FUSED) cepbuf[(gl - cep_frame0) * kFusedCols + oc] = v;
            else row[oc] = v;
          }
        } else {
          const int kp = (lane >> 2) & 3;
          const float4* __restrict__ dsrc = reinterpret_cast<const float4*>(mtile + kp * 4 * h_dd_quads);
          f32x4 c0 = {0.0f, 0.0f, 0.0f, 0.0f}, c1 = {0.0f, 0.0f, 0.0f, 0.0f};
          for (int t = 0; t < h_dd_quads; ++t) {
            const float4 a = t_dd_a[t * 64 + lane], x = dsrc[t];
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, x.x, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, x.y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, x.z, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, x.w, c1, 0, 0, 0);
          }
          const float4 lift = *reinterpret_cast<const float4*>(t_lifter + 4 * q);
          const float lf[4] = {lift.x, lift.y, lift.z, lift.w};
          const int cbase = 4 * q;  // first cepstrum of this lane's group (kp = 0 lanes store)
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            float v = c0[i] + c1[i];
            v += dpp_mov<0x104, true>(0.0f, v);  // kp 0 + 1, 1 + 2, 2 + 3, 3
            v += dpp_mov<0x108, true>(0.0f, v);  // kp 0: (0 + 1) + (2 + 3)
            v *= lf[i];
            const int c = cbase + i;
            int oc = c;
            if (p.htk_compat) {
              oc = c == 0 ? p.num_ceps - 1 : c - 1;
              if (c == 0 && !p.use_energy)
                v = static_cast<float>(static_cast<double>(v) * 1.4142135623730950488016887);
            }
            if (mvalid && kp == 0 && c < p.num_ceps && !(c == 0 && p.use_energy)) {
              if (FUSED) cepbuf[(mgl - cep_frame0) * kFusedCols + oc] = v;
              else mrow[oc] = v;
            }
          }
          if (p.use_energy && valid && l == 0) {
            if (FUSED) cepbuf[(gl - cep_frame0) * kFusedCols + (p.htk_compat ? p.num_ceps - 1 : 0)] = log_energy;
            else row[p.htk_compat ? p.num_ceps - 1 : 0] = log_energy;
          }
        }
      }
    }
    wave_lds_sync();  // the tile is reused by the next frame set
  }
  if (FUSED) {
    // ---- [cepstra | delta | delta-delta] of the workgroup's frames (DeltaPostProcessor order 2, window
    // 2; [KALDI-UPSTREAM] ComputeDeltas): frame indices clamp at the ends of the utterance, the same
    // products in the same order as delta_flat_o2w2_kernel (kernels_post.hip) ------------------------
    __syncthreads();
    const int D = p.num_ceps, OD = 3 * D;
    const int64_t f_first = static_cast<int64_t>(pu_set0) * 4;
    const int64_t rows64 = pu_T - f_first < 4 * kFusedSets ? pu_T - f_first : 4 * kFusedSets;
    float sc[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) sc[i] = p.delta_scales[i];
    float* __restrict__ obase = out + (pu_f0 + f_first) * static_cast<int64_t>(p.out_cols);
    const int last = static_cast<int>(pu_T - 1 - cep_frame0);   // buffer row of the utterance's last frame
    const int first = static_cast<int>(-cep_frame0);            // ... and of its first frame
    // element (row r, column c): its 9 clamped neighbours are read once for the three orders
    for (int idx = threadIdx.x; idx < static_cast<int>(rows64) * D; idx += blockDim.x) {
      const int r = idx / D, c = idx - r * D;
      const int centre = r + 4;
      float x[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        int t = centre + j - 4;
        t = t < first ? first : (t > last ? last : t);
        x[j] = cepbuf[t * kFusedCols + c];
      }
      float* __restrict__ orow = obase + r * OD + c;
      int soff = 0;
#pragma unroll
      for (int i = 0; i <= 2; ++i) {
        const int max_off = 2 * i;
        float acc = 0.0f;
#pragma unroll
        for (int j = -max_off; j <= max_off; ++j) {
          const float w = sc[soff + j + max_off];
          if (w != 0.0f) acc += w * x[j + 4];
        }
        orow[i * D] = acc;
        soff += 2 * max_off + 1;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Frames that pad to 256 samples (8 kHz audio, 10-16 ms windows at 16 kHz): TWO real frames per 16-lane
// row, packed as one complex signal z[n] = x_a[n] + i x_b[n], n < 256.  The same two register passes as
// above transform it, and the spectra separate without a twiddle: X_a[k] = (Z[k] + conj Z[256 - k]) / 2,
// X_b[k] = (Z[k] - conj Z[256 - k]) / 2i.  A wave64 therefore works on 8 frames per iteration instead of
// the 4 zero-extended ones of the 512-point form (half the transform arithmetic per frame); the mel
// filterbank (129 bins, the same MFMA block tables, built without the zero-tap spreading) and the DCT
// run once per sub-frame.  Flat scheduling only: VTLN batches and the spectrogram keep the 512-point form.
// The two real transforms share their roundings (the last bits of X_a depend on x_b), so the pairing is
// a property of the utterance, not of the batch: frames 2 m and 2 m + 1 of one utterance (PairRec table,
// built once per offsets table by build_pair_table_kernel); an odd last frame is transformed with itself.
// ---------------------------------------------------------------------------------------------------
constexpr int kDualSub = 136;  // float offset of sub-frame b inside a row's power tile (129 bins + pad)

template <int NJ, int KIND, int ENERGY, bool DITHER, bool SNIP>
__global__ __launch_bounds__(kMaxWaves * 64, 4) void fbank256x2_kernel(const Fast512Params p,
                                                                     const BatchArgs b,
                                                                     float* __restrict__ out,
                                                                     double* __restrict__ energy_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tab = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < p.table_floats; i += blockDim.x) tab[i] = p.tables[i];
  __syncthreads();
  const float2* __restrict__ t_win = reinterpret_cast<const float2*>(tab + kFastHeaderFloats);
  const float2* __restrict__ t_tw16 = t_win + 16 * 18;
  const float4* __restrict__ t_mm_a = reinterpret_cast<const float4*>(tab + p.off_mm_a);
  const float* __restrict__ t_lifter = tab + p.off_lifter;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l = lane & 15, q = lane >> 4;
  const int tab_bytes = (p.table_floats * 4 + 255) & ~255;
  char* wave_base = smem + tab_bytes + (wid * 4 + q) * kFrameTileBytes;
  float2* tile = reinterpret_cast<float2*>(wave_base);
  float* ptile = reinterpret_cast<float*>(wave_base) + q * 16;   // sub-frame a at 0, b at kDualSub
  tile[l * kTileRow + 16] = make_float2(0.0f, 0.0f);  // (padding column: see fbank512_kernel)
  const int mj = lane & 3;
  const float* __restrict__ mm_lane = tab + p.off_mm_lane;
  const int mm_start = reinterpret_cast<const int*>(mm_lane)[lane];
  const int mm_out = reinterpret_cast<const int*>(mm_lane)[64 + lane];
  const float mm_f1 = mm_lane[128 + lane], mm_f2 = mm_lane[192 + lane], mm_f3 = mm_lane[256 + lane];
  float* __restrict__ mtile =
      reinterpret_cast<float*>(smem + tab_bytes + (wid * 4 + mj) * kFrameTileBytes) + mj * 16;

  const float win_len_f = static_cast<float>(p.win_len), inv_win_len = 1.0f / win_len_f;
  const int n_waves = blockDim.x >> 6;
  // (the fix-up launch walks a list whose length only the device knows: BatchArgs::n_pairs_dev)
  const int64_t n_pairs = b.n_pairs_dev ? static_cast<int64_t>(*b.n_pairs_dev) : b.n_pairs;
  const int64_t n_sets = (n_pairs + 3) >> 2;
  const int64_t set_stride = static_cast<int64_t>(gridDim.x) * n_waves;
  const int64_t last_pair = n_pairs - 1;
  // a pair record as two 16-byte halves: {start_a, start_b} and {frame_a, utt1, flags}
  auto starts_of = [&](int64_t pi) -> longlong2 {
    return reinterpret_cast<const longlong2*>(b.pair_tab + (pi < last_pair ? pi : last_pair))[0];
  };
  auto meta_of = [&](int64_t pi) -> int4 {
    return reinterpret_cast<const int4*>(b.pair_tab + (pi < last_pair ? pi : last_pair))[1];
  };
  // NJ = 13: the 25 ms / 8 kHz window (200 samples: only element j = 12 can fall outside the window)
  const bool in_last = l + 16 * (NJ - 1) < p.win_len;
  auto in_window = [&](int j) -> bool {
    if (NJ == 13) return j < NJ - 1 || in_last;
    return l + 16 * j < p.win_len;
  };
  // Sample ingest (round 6; first built for kernels_fbank1024x2.hip): the two frames of a pair lie in ONE span of
  // the utterance - frame b starts `off_b` = start_b - start_a samples (0 ... the frame shift) behind frame a -
  // which comes in as 16-byte pieces, lane l of the row fetching pieces l, l + 16, ... (3 wave instructions for a
  // 25 ms / 10 ms pair at 8 kHz, where a 16-bit load per element and frame took 26: the texture addresser takes a
  // wave instruction per 64 addresses whatever their width), is laid down in the row's LDS tile at the top of
  // the next iteration and read back element by element.  The last bytes of a span that do not fill a piece come
  // through one 16-bit load of the first lanes: no load reaches beyond the span, i.e. beyond the utterance.
  typedef int int4_a2 __attribute__((ext_vector_type(4), aligned(2)));
  constexpr int NP = NJ == 13 ? 3 : 4;   // pieces per lane: spans of up to 384 / 512 samples (launch_fbank512)
  int4_a2 qv[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) qv[i] = int4_a2{0, 0, 0, 0};
  int tailv = 0, offb_next = 0;
  auto load_span = [&](const longlong2 st) __attribute__((always_inline)) {
    const char* __restrict__ pa = reinterpret_cast<const char*>(b.wave + st.x);
    const int ob = static_cast<int>(st.y - st.x);
    const int span_bytes = 2 * (ob + p.win_len);
    const int nf = span_bytes >> 4, nt = (span_bytes & 15) >> 1;
#pragma unroll
    for (int i = 0; i < NP; ++i)
      qv[i] = *reinterpret_cast<const int4_a2*>(pa + (l + 16 * i < nf ? 16 * (l + 16 * i) : 0));
    tailv = *reinterpret_cast<const short*>(pa + (l < nt ? 16 * nf + 2 * l : 0));
    offb_next = ob;
  };
  int64_t set = static_cast<int64_t>(blockIdx.x) * n_waves + wid;
  longlong2 next_starts = {0, 0};
  int4 meta_next = {0, 0, 0, 0};
  if (set < n_sets) {
    load_span(starts_of(set * 4 + q));
    next_starts = starts_of((set + set_stride) * 4 + q);
    meta_next = meta_of(set * 4 + q);
  }
  // Loads and stores share one in-order counter (vmcnt): the wait for the prefetched samples at the top of
  // the loop also covers the output stores issued after them.  Here the prefetch is settled by hand in
  // front of the iteration's first store (and once before the loop, so that no path into the loop head
  // carries a pending load): 0.32 against 0.34 ms per 1.19 M frames.  (The same change on fbank512_kernel
  // measured 5-10 % SLOWER, wherever the settle point was put, and was dropped there.)
  auto settle_prefetch = [&]() {
#pragma unroll
    for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(qv[i]));
    asm volatile("" : "+v"(tailv), "+v"(offb_next));
    asm volatile("" : "+v"(next_starts.x), "+v"(next_starts.y));
    asm volatile("" : "+v"(meta_next.x), "+v"(meta_next.y), "+v"(meta_next.z), "+v"(meta_next.w));
  };
  settle_prefetch();
  for (; set < n_sets; set += set_stride) {
    const int4 meta = meta_next;
    int4 mmeta = meta_of(set * 4 + mj);  // (the MFMA view's pair, below)
    const int64_t ga = static_cast<int64_t>(static_cast<unsigned>(meta.x)) | (static_cast<int64_t>(meta.y) << 32);
    const bool valid_a = set * 4 + q <= last_pair, valid_b = valid_a && (meta.w & 4) != 0;
    const int edge_a = (meta.w & 1) ? meta.z : 0, edge_b = (meta.w & 2) ? meta.z : 0;

    // ---- A: DC removal, pre-emphasis, window (per sub-frame: xe = frame a, xo = frame b) ---------------
    float4 win4[(NJ + 1) / 2];
    read_quads<(NJ + 1) / 2>(t_win + l * 18, win4);
    unsigned dk[4] = {0, 0, 0, 0};
    if (DITHER) {
      // (keyed by the frames' places in their utterance - PairRec.utt1 - not in the batch: snf_internal.h)
      const int64_t du = meta.z > 0 ? meta.z - 1 : 0;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const unsigned long long k =
            wave_noise_id(b, du, ga + s2 - b.frame_offsets[du]) ^ p.seed;
        dk[2 * s2] = fmix32(static_cast<unsigned>(k));
        dk[2 * s2 + 1] = fmix32(static_cast<unsigned>(k >> 32) ^ dk[2 * s2]);
      }
    }
    float xe[NJ], xo[NJ];
    float part_a = 0.0f, part_b = 0.0f;
    constexpr bool kIntSum = !DITHER && SNIP;  // (integer sums are exact: see fbank512_kernel)
    int sum_a = 0, sum_b = 0;
    {
      // the span of this iteration's pair: registers -> the row's tile (whole pieces, then the tail), elements back
      const int ob = offb_next;
      const int span_bytes = 2 * (ob + p.win_len);
      const int nf = span_bytes >> 4, nt = (span_bytes & 15) >> 1;
      char* __restrict__ sb = reinterpret_cast<char*>(tile);
#pragma unroll
      for (int i = 0; i < NP; ++i)
        if (l + 16 * i < nf)
          *reinterpret_cast<int4*>(sb + 16 * (l + 16 * i)) = make_int4(qv[i].x, qv[i].y, qv[i].z, qv[i].w);
      if (l < nt) *reinterpret_cast<short*>(sb + 16 * nf + 2 * l) = static_cast<short>(tailv);
      wave_lds_sync();
      const short* __restrict__ sa = reinterpret_cast<const short*>(tile) + l;
      const short* __restrict__ sbb = sa + ob;
      int va[NJ], vb[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        va[j] = sa[16 * j];
        vb[j] = sbb[16 * j];
      }
      lds_wait();
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        xe[j] = static_cast<float>(va[j]);
        xo[j] = static_cast<float>(vb[j]);
        if (kIntSum) {
          sum_a += in_window(j) ? va[j] : 0;
          sum_b += in_window(j) ? vb[j] : 0;
        }
      }
      wave_lds_sync();
    }
    if (!SNIP && (edge_a | edge_b) != 0) {
      // [KALDI-UPSTREAM] ExtractWindow, snip_edges = false: reflected samples for the frames that reach
      // outside their utterance (their prefetched samples came from a clamped window)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int edge = s2 ? edge_b : edge_a;
        if (edge != 0 && (s2 ? valid_b : valid_a)) {
          const int64_t u = edge - 1;
          const int64_t s0 = b.sample_offsets[u], n = b.sample_offsets[u + 1] - s0;
          const int64_t rel =
              (ga + s2 - b.frame_offsets[u]) * p.win_shift + p.win_shift / 2 - p.win_len / 2;
          const int16_t* __restrict__ w0 = b.wave + s0;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (in_window(j)) {
              int64_t k = rel + l + 16 * j;
              while (k < 0 || k >= n) k = k < 0 ? -k - 1 : 2 * n - 1 - k;
              const float v = static_cast<float>(w0[k]);
              if (s2) xo[j] = v; else xe[j] = v;
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (DITHER) {  // Kaldi dithers before the DC removal
        xe[j] += p.dither * gauss_pair(dk[0], dk[1], static_cast<unsigned>(l + 16 * j)).x;
        xo[j] += p.dither * gauss_pair(dk[2], dk[3], static_cast<unsigned>(l + 16 * j)).x;
      }
      if (!kIntSum) {
        part_a += in_window(j) ? xe[j] : 0.0f;
        part_b += in_window(j) ? xo[j] : 0.0f;
      }
    }
    if (kIntSum) {
      part_a = static_cast<float>(sum_a);
      part_b = static_cast<float>(sum_b);
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) asm volatile("" : "+v"(xe[j]), "+v"(xo[j]) : : "memory");
    asm volatile("" : "+v"(part_a), "+v"(part_b) : : "memory");
    {  // prefetch: samples of the next set, start offsets of the set after it
      load_span(next_starts);
      next_starts = starts_of((set + 2 * set_stride) * 4 + q);
      meta_next = meta_of((set + set_stride) * 4 + q);
    }
    float nm_a = 0.0f, nm_b = 0.0f;
    if (p.remove_dc) {
      const float sa = row_sum16(part_a), sb = row_sum16(part_b);
      if (kIntSum) {
        const float qa = sa * inv_win_len, qb = sb * inv_win_len;
        nm_a = -__builtin_fmaf(__builtin_fmaf(-qa, win_len_f, sa), inv_win_len, qa);
        nm_b = -__builtin_fmaf(__builtin_fmaf(-qb, win_len_f, sb), inv_win_len, qb);
      } else {
        nm_a = -sa / win_len_f;
        nm_b = -sb / win_len_f;
      }
    }
    lds_wait();
    // the left neighbour x[n-1] is element n-1 = lane l-1 (same j), lane 15 of j-1 for lane 0
    float2 z[16];
    float er_a = 0.0f, er_b = 0.0f, ep_a = 0.0f, ep_b = 0.0f;
    float prev_a = xe[0] + nm_a, prev_b = xo[0] + nm_b;  // lane 0, j = 0: x[-1] := x[0]
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < NJ) {
        const bool in = in_window(j);
        const float ae = xe[j] + nm_a, ao = xo[j] + nm_b;
        const float rot_a = dpp_row_ror<0x121>(ae), rot_b = dpp_row_ror<0x121>(ao);
        const float pa = l == 0 ? prev_a : rot_a, pb = l == 0 ? prev_b : rot_b;
        prev_a = rot_a;
        prev_b = rot_b;
        const float2 w = (j & 1) ? make_float2(win4[j >> 1].z, win4[j >> 1].w)
                                 : make_float2(win4[j >> 1].x, win4[j >> 1].y);  // (w[n], w[n]); 0 outside
        if (ENERGY == 1 && in) {
          er_a += ae * ae;
          er_b += ao * ao;
        }
        const float ye = (ae - p.preemph * pa) * w.x;
        const float yo = (ao - p.preemph * pb) * w.y;
        z[j] = make_float2(ye, yo);
        ep_a += ye * ye;   // (the energies of the windowed frames: the pairing decision below, ENERGY == 2)
        ep_b += yo * yo;
      } else {
        z[j] = make_float2(0.0f, 0.0f);
      }
    }
    float e_lin_a = 0.0f, e_lin_b = 0.0f;
    {
      const float ew_a = row_sum16(ep_a), ew_b = row_sum16(ep_b);
      if (ENERGY == 2) {
        e_lin_a = ew_a;
        e_lin_b = ew_b;
      } else if (ENERGY == 1) {
        e_lin_a = row_sum16(er_a);
        e_lin_b = row_sum16(er_b);
      }
      // The two frames share the roundings of one transform.  A pair of very different energies (a quiet frame
      // beside an onset, digital silence beside anything) is put down for the fix-up launch, which transforms
      // each of its frames alone and rewrites the two rows (BatchArgs::fix_tab).
      if (b.fix_tab != nullptr && valid_b && l == 0 &&
          fmaxf(ew_a, ew_b) > b.split_ratio * fminf(ew_a, ew_b)) {
        const longlong2 st = starts_of(set * 4 + q);
        const unsigned at = atomicAdd(b.fix_count, 2u);
        longlong2* rec = reinterpret_cast<longlong2*>(b.fix_tab + at);
        int4* rec_meta = reinterpret_cast<int4*>(b.fix_tab + at);
        rec[0] = make_longlong2(st.x, st.x);
        rec_meta[1] = make_int4(meta.x, meta.y, meta.z, meta.w & 1);
        const int64_t gb = ga + 1;
        rec[2] = make_longlong2(st.y, st.y);
        rec_meta[3] = make_int4(static_cast<int>(gb), static_cast<int>(gb >> 32), meta.z, (meta.w >> 1) & 1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- B / C: the 256-point complex transform (identical to fbank512_kernel) ------------------------
    fft16_lf(z);
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) tile[k2 * kTileRow + l] = z[k2];
    wave_lds_sync();
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
    fft16_twin(z, ct);
    __builtin_amdgcn_sched_barrier(0);
    wave_lds_sync();

    // ---- D: separate the two spectra, power (x4 like fbank512_kernel): bins k = l + 16 k1 <= 128 --------
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 8; ++r) tile[r * 16 + l] = z[r + 8];
    wave_lds_sync();
    float2 zpart[8];
    read8_b64_rev128(tile + (16 - l), zpart);   // zpart[k1] = Z[256 - l - 16 k1]
    float pa[8], pb[8];
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
      const float2 zk = z[k1], zp = zpart[k1];
      const float c_re = zk.x + zp.x, c_im = zk.y - zp.y;   // 2 X_a[k]
      const float d_re = zk.y + zp.y, d_im = zp.x - zk.x;   // 2 X_b[k]
      pa[k1] = c_re * c_re + c_im * c_im;
      pb[k1] = d_re * d_re + d_im * d_im;
    }
    if (l == 0) {  // k = 0 pairs with itself: X_a[0] = Re Z[0], X_b[0] = Im Z[0]
      pa[0] = 4.0f * z[0].x * z[0].x;
      pb[0] = 4.0f * z[0].y * z[0].y;
    }
    const float p128_a = 4.0f * z[8].x * z[8].x, p128_b = 4.0f * z[8].y * z[8].y;  // lane 0: Z[128]
    wave_lds_sync();
    settle_prefetch();
    asm volatile("" : "+v"(mmeta.x), "+v"(mmeta.y), "+v"(mmeta.w));
    // ---- E: power tiles ----------------------------------------------------------------------------------
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) {
      ptile[l + 16 * k1] = pa[k1];
      ptile[kDualSub + l + 16 * k1] = pb[k1];
    }
    if (l == 0) {
      ptile[128] = p128_a;
      ptile[kDualSub + 128] = p128_b;
    }
    wave_lds_sync();

    // ---- log-energy column ---------------------------------------------------------------------------------
    float log_e[2] = {0.0f, 0.0f};
    if (ENERGY != 0) {
      if (KIND == SNF_KIND_PLP) {
        if (l == 0) {
          if (valid_a) energy_out[ga] = static_cast<double>(e_lin_a);  // (plp_tail_kernel takes the double log)
          if (valid_b) energy_out[ga + 1] = static_cast<double>(e_lin_b);  // (plp_tail_kernel takes the double log)
        }
      } else {
        log_e[0] = fast_log(floor_eps(e_lin_a));
        log_e[1] = fast_log(floor_eps(e_lin_b));
        if (p.has_floor) {
          if (log_e[0] < p.log_energy_floor) log_e[0] = p.log_energy_floor;
          if (log_e[1] < p.log_energy_floor) log_e[1] = p.log_energy_floor;
        }
      }
    }
    float* __restrict__ row = out + ga * static_cast<int64_t>(p.out_cols);  // frame a; frame b: + out_cols

    if (KIND == SNF_KIND_SPECTROGRAM) {
      // the 129 log powers of each frame straight from the registers: lane l of the row holds bins l + 16 k1
      // (16 lanes x 4 bytes per store), lane 0 bin 128 as well; column 0 is the log energy (Kaldi's
      // SpectrogramComputer).  The powers above carry a factor 4 (see fbank512_kernel).
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (s2 ? valid_b : valid_a) {
          float* __restrict__ r = row + s2 * p.out_cols;
#pragma unroll
          for (int k1 = 0; k1 < 8; ++k1) {
            float v = fast_log(fmaxf(0.25f * (s2 ? pb[k1] : pa[k1]), FLT_EPSILON));
            if (k1 == 0 && l == 0) v = log_e[s2];
            r[l + 16 * k1] = v;
          }
          if (l == 0) r[128] = fast_log(fmaxf(0.25f * (s2 ? p128_b : p128_a), FLT_EPSILON));
        }
      }
      wave_lds_sync();  // the tile is reused by the next frame set
      continue;
    }
    // ---- F: mel filterbank on the matrix pipe, one chain per sub-frame ------------------------------------
    float mel[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float4* __restrict__ bsrc = reinterpret_cast<const float4*>(mtile + kDualSub * s2 + mm_start);
      const float4* __restrict__ asrc = t_mm_a + lane;
      f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
      float4 a0 = asrc[0], x0 = bsrc[0];
      for (int t = 0; t < p.mm_quads; t += 2) {  // (mm_quads is even; the table ends with a row of zeros)
        const float4 a1 = asrc[(t + 1) * 64], x1 = bsrc[t + 1];
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, x0.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, x0.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, x0.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, x0.w, acc1, 0, 0, 0);
        a0 = asrc[(t + 2) * 64];
        x0 = bsrc[t + 2];
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, x1.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, x1.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.z, x1.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.w, x1.w, acc1, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) mel[s2][i] = acc0[i] + acc1[i];
      if (p.mm_levels > 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float own = mel[s2][i];
          fmac_row_shl<4>(mel[s2][i], own, mm_f1);
          fmac_row_shl<8>(mel[s2][i], own, mm_f2);
          if (p.mm_levels > 3) fmac_row_shl<12>(mel[s2][i], own, mm_f3);
        }
      }
    }
    // MFMA view: lane 4 b + j -> pair j of the set
    const int64_t mga = static_cast<int64_t>(static_cast<unsigned>(mmeta.x)) | (static_cast<int64_t>(mmeta.y) << 32);
    const bool mvalid[2] = {set * 4 + mj <= last_pair, set * 4 + mj <= last_pair && (mmeta.w & 4) != 0};
    float* __restrict__ mrow = out + mga * static_cast<int64_t>(p.out_cols);
    const int mel_col = (KIND == SNF_KIND_FBANK && p.use_energy && !p.htk_compat) ? 1 : 0;
    if (KIND == SNF_KIND_FBANK || KIND == SNF_KIND_PLP) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if (KIND == SNF_KIND_FBANK && p.use_log) {
#pragma unroll
          for (int i = 0; i < 4; ++i) mel[s2][i] = fast_log(floor_eps(mel[s2][i]));
        }
        if (mvalid[s2] && mm_out >= 0) {
          float* __restrict__ dst = mrow + s2 * p.out_cols + mel_col + mm_out;
          if (mm_out + 4 <= p.num_bins) {
            __builtin_nontemporal_store(f32x4_a4{mel[s2][0], mel[s2][1], mel[s2][2], mel[s2][3]},
                                        reinterpret_cast<f32x4_a4*>(dst));
          } else {
#pragma unroll
            for (int i = 0; i < 3; ++i)
              if (mm_out + i < p.num_bins) dst[i] = mel[s2][i];
          }
        }
        if (KIND == SNF_KIND_FBANK && p.use_energy && l == 0 && (s2 ? valid_b : valid_a))
          row[s2 * p.out_cols + (p.htk_compat ? p.num_bins : 0)] = log_e[s2];
      }
    }
    if (KIND == SNF_KIND_MFCC) {
      // log-mel of both sub-frames back to their (now idle) power tiles, then the DCT-II chains
      wave_lds_sync();
      if (mm_out >= 0) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          *reinterpret_cast<float4*>(mtile + kDualSub * s2 + mm_out) =
              make_float4(fast_log(floor_eps(mel[s2][0])), fast_log(floor_eps(mel[s2][1])),
                          fast_log(floor_eps(mel[s2][2])), fast_log(floor_eps(mel[s2][3])));
      }
      wave_lds_sync();
      // DCT-II + lifter on the vector pipe: lane l of a row owns cepstrum l of the row's two sub-frames
      const float4* __restrict__ t_dd_v = reinterpret_cast<const float4*>(tab + p.off_dd_v);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const float4* __restrict__ dw = t_dd_v + l;
        const float4* __restrict__ dx = reinterpret_cast<const float4*>(ptile + kDualSub * s2);
        float v = 0.0f;
#pragma unroll 2
        for (int g4 = 0; g4 < p.dd_groups; ++g4) {
          const float4 w = dw[g4 * 16], x = dx[g4];
          v += w.x * x.x;
          v += w.y * x.y;
          v += w.z * x.z;
          v += w.w * x.w;
        }
        v *= t_lifter[l];
        if (l == 0 && p.use_energy) v = log_e[s2];
        int oc = l;
        if (p.htk_compat) {
          oc = l == 0 ? p.num_ceps - 1 : l - 1;
          if (l == 0 && !p.use_energy)
            v = static_cast<float>(static_cast<double>(v) * 1.4142135623730950488016887);
        }
        if ((s2 ? valid_b : valid_a) && l < p.num_ceps) row[s2 * p.out_cols + oc] = v;
      }
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
                                         int64_t total_frames, int64_t total_samples, int win_shift,
                                         int win_len, int snip_edges, int64_t* __restrict__ frame_start,
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
  if (safe + win_len > n) safe = n - win_len;
  // an utterance shorter than a window (its frames are recomputed by the generic kernel afterwards, the
  // host sees to that): any window inside the batch will do
  int64_t abs_start = s0 + safe;
  if (abs_start + win_len > total_samples) abs_start = total_samples - win_len;
  frame_start[g] = abs_start < 0 ? 0 : abs_start;
  frame_edge[g] = edge ? static_cast<int32_t>(u + 1) : 0;
}

// one thread per frame pair of fbank256x2_kernel (see PairRec): pair_offsets[u] = pairs of the utterances
// before u, an utterance of T frames holding (T + 1) / 2 of them.
__global__ void build_pair_table_kernel(const int64_t* __restrict__ frame_offsets,
                                        const int64_t* __restrict__ sample_offsets,
                                        const int64_t* __restrict__ pair_offsets, int64_t n_utts,
                                        int64_t n_pairs, int win_shift, int win_len, int snip_edges,
                                        PairRec* __restrict__ pairs) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= n_pairs) return;
  const int64_t u = find_utt(pair_offsets, n_utts, g);
  const int64_t s0 = sample_offsets[u], n = sample_offsets[u + 1] - s0;
  const int64_t f0 = frame_offsets[u], n_frames = frame_offsets[u + 1] - f0;
  const int64_t f = 2 * (g - pair_offsets[u]);
  const bool has_b = f + 1 < n_frames;
  PairRec r;
  r.frame_a = f0 + f;
  r.utt1 = static_cast<int32_t>(u + 1);  // (always: the dither stream is keyed per utterance; `flags` marks edges)
  r.flags = has_b ? 4 : 0;
  int64_t start[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    const int64_t fr = f + (has_b ? s2 : 0);
    if (snip_edges) {
      start[s2] = s0 + fr * win_shift;
    } else {  // (as build_frame_start_kernel)
      const int64_t rel = fr * win_shift + win_shift / 2 - win_len / 2;
      int64_t safe = rel < 0 ? 0 : rel;
      if (safe + win_len > n) safe = n - win_len;
      start[s2] = s0 + safe;
      if (rel < 0 || rel + win_len > n) r.flags |= 1 << s2;
    }
  }
  r.start_a = start[0];
  r.start_b = start[1];
  pairs[g] = r;
}

int launch_build_pair_table(const int64_t* d_frame_offsets, const int64_t* d_sample_offsets,
                            const int64_t* d_pair_offsets, int64_t n_utts, int64_t n_pairs, int win_shift,
                            int win_len, int snip_edges, PairRec* d_pairs, hipStream_t stream) {
  if (n_pairs <= 0) return SNF_OK;
  hipLaunchKernelGGL(build_pair_table_kernel, dim3(static_cast<unsigned>((n_pairs + 255) / 256)),
                     dim3(256), 0, stream, d_frame_offsets, d_sample_offsets, d_pair_offsets, n_utts,
                     n_pairs, win_shift, win_len, snip_edges, d_pairs);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

int launch_build_frame_start(const int64_t* d_frame_offsets, const int64_t* d_sample_offsets,
                             int64_t n_utts, int64_t total_frames, int64_t total_samples, int win_shift,
                             int win_len, int snip_edges, int64_t* d_frame_start, int32_t* d_frame_edge,
                             int32_t* d_frame_utt, hipStream_t stream) {
  if (total_frames <= 0) return SNF_OK;
  hipLaunchKernelGGL(build_frame_start_kernel,
                     dim3(static_cast<unsigned>((total_frames + 255) / 256)), dim3(256), 0, stream,
                     d_frame_offsets, d_sample_offsets, n_utts, total_frames, total_samples, win_shift,
                     win_len, snip_edges, d_frame_start, d_frame_edge, d_frame_utt);
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
  // needs, still several times faster than the LDS radix-2 kernel.
  if ((mp.padded != 512 && mp.padded != 256 && mp.padded != 128) || !mp.pow2) return false;
  if (mp.win_len & 1) return false;
  if (mp.kind != SNF_KIND_FBANK && mp.kind != SNF_KIND_MFCC && mp.kind != SNF_KIND_PLP &&
      mp.kind != SNF_KIND_SPECTROGRAM && mp.kind != SNF_KIND_ENERGY)
    return false;
  // (the spectrogram needs the N / 2 + 1 bins themselves: 512-sample frames, or 256-sample frames on the
  // two-frames-per-transform kernel, whose conditions are those of fast512_dual_eligible)
  if (mp.kind == SNF_KIND_SPECTROGRAM && mp.padded != 512 &&
      !(mp.padded == 256 && mp.win_len + mp.win_shift <= 512 && !getenv("SNF_DISABLE_DUAL256")))
    return false;
  if (mp.kind == SNF_KIND_FBANK && !mp.use_power) return false;
  if (mp.num_bins > kFast512MaxBins) return false;
  if (mp.kind == SNF_KIND_MFCC && mp.num_ceps > 16) return false;
  return true;
}

bool fast512_dual_eligible(const MelParams& mp) {
  if (getenv("SNF_DISABLE_DUAL256")) return false;
  // (a pair is fetched as one span of the utterance: at most 512 samples, kernels_fbank512.hip load_span)
  return fast512_eligible(mp, false) && mp.padded == 256 && mp.win_len + mp.win_shift <= 512 &&
         (mp.kind == SNF_KIND_FBANK || mp.kind == SNF_KIND_MFCC || mp.kind == SNF_KIND_PLP ||
          mp.kind == SNF_KIND_SPECTROGRAM);
}

// Builds the packed LDS table blob from the plan's host tables (warp 1.0 mel banks).  `dual`: tables of
// fbank256x2_kernel (window value per sample instead of per pair, mel taps on the 129 bins themselves).
int fast512_build(const MelParams& mp, const std::vector<float>& window, const MelBanksHost& mb_in,
                  const std::vector<float>& dct, const std::vector<float>& lifter, bool dual,
                  std::vector<float>* blob, Fast512Params* out) {
  constexpr double kTwoPi = 6.283185307179586476925286766559005;
  // frames shorter than 512 samples: spread the taps of every bin over the bins of the 512-point
  // spectrum of the zero-extended frame (see fast512_eligible)
  MelBanksHost spread;
  const int bin_stride = dual ? 1 : 512 / mp.padded;
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
  p.dual = dual ? 1 : 0;
  blob->clear();
  blob->resize(kFastHeaderFloats, 0.0f);  // header, filled in at the end
  // window pairs, lane-major: row l = elements l + 16 j (j < 16), 2 complex of padding
  for (int l = 0; l < 16; ++l)
    for (int j = 0; j < 18; ++j) {
      const int n = l + 16 * j;
      if (dual) {  // element n of the row = sample n of BOTH sub-frames
        const float w = j < 16 && n < mp.win_len ? window[n] : 0.0f;
        blob->push_back(w);
        blob->push_back(w);
        continue;
      }
      blob->push_back(j < 16 && 2 * n < mp.win_len ? window[2 * n] : 0.0f);
      blob->push_back(j < 16 && 2 * n + 1 < mp.win_len ? window[2 * n + 1] : 0.0f);
    }
  // Twiddles as (cos, tan) pairs: W = cos (1 + i tan) (device_fft.h, Linzer-Feig butterflies).  An exact
  // -i (zero cosine) is (2^-40, -2^40): every product with it is an exact power-of-two scaling.
  auto push_cos_tan = [&](int num, int den) {
    const int r = ((num % den) + den) % den;  // exp(-2 pi i r / den)
    if (4 * r == den) {                        // -i
      blob->push_back(9.094947017729282e-13f);
      blob->push_back(-1099511627776.0f);
      return;
    }
    if (4 * r == 3 * den) {                    // +i
      blob->push_back(9.094947017729282e-13f);
      blob->push_back(1099511627776.0f);
      return;
    }
    const double a = -kTwoPi * r / den;
    blob->push_back(static_cast<float>(std::cos(a)));
    blob->push_back(static_cast<float>(std::tan(a)));
  };
  // inter-pass twiddles, lane-major: row n1 = exp(-2 pi i n1 k2 / 256), k2 < 16
  for (int n1 = 0; n1 < 16; ++n1)
    for (int k2 = 0; k2 < 18; ++k2) push_cos_tan(n1 * (k2 < 16 ? k2 : 0), 256);
  // unpack twiddles, lane-major: row l = exp(-2 pi i (l + 16 k1) / 512), k1 < 8 (angles in (-pi/2, 0])
  for (int l = 0; l < 16; ++l)
    for (int k1 = 0; k1 < 10; ++k1) push_cos_tan(l + 16 * (k1 < 8 ? k1 : 0), 512);
  // ---- mel filterbank as MFMA blocks --------------------------------------------------------------
  // Group g = mel bins 4 g .. 4 g + 3 spans the FFT bins [glo, ghi); it is cut into parts[g] runs of at
  // most 4 mm_quads bins, one MFMA block each.  More parts for the widest groups shorten the chain;
  // the parts of a group must sit in ONE row of 4 blocks (their sums are added through row DPP).
  const int n_groups = (mb.num_bins + 3) / 4;
  std::vector<int> glo(n_groups, 0), ghi(n_groups, 0), parts(n_groups, 1);
  for (int g = 0; g < n_groups; ++g) {
    int lo = 1 << 30, hi = 0;
    for (int m = 4 * g; m < 4 * g + 4 && m < mb.num_bins; ++m) {
      if (mb.size[m] <= 0) continue;
      lo = std::min(lo, mb.first[m]);
      hi = std::max(hi, mb.first[m] + mb.size[m]);
    }
    if (hi == 0) lo = 0;
    glo[g] = lo & ~3;
    ghi[g] = std::max(hi, glo[g]);
  }
  auto per_part = [&](int g, int np) { return ((ghi[g] - glo[g] + np - 1) / np + 3) & ~3; };
  // first-fit-decreasing of the part counts into 4 rows of 4 blocks; returns false when they do not fit
  auto pack = [&](const std::vector<int>& np, std::vector<int>* first_block) {
    std::vector<int> order(np.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int>(i);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return np[x] > np[y]; });
    int used[4] = {0, 0, 0, 0};
    if (first_block) first_block->assign(np.size(), -1);
    for (int g : order) {
      int r = 0;
      while (r < 4 && used[r] + np[g] > 4) ++r;
      if (r == 4) return false;
      if (first_block) (*first_block)[g] = 4 * r + used[r];
      used[r] += np[g];
    }
    return true;
  };
  if (n_groups > 0 && !pack(parts, nullptr)) return 1;  // (more than 16 groups: not eligible)
  for (;;) {
    int worst = -1, worst_len = 0;
    for (int g = 0; g < n_groups; ++g)
      if (per_part(g, parts[g]) > worst_len) {
        worst_len = per_part(g, parts[g]);
        worst = g;
      }
    if (worst < 0 || parts[worst] >= 4 || per_part(worst, parts[worst] + 1) >= worst_len) break;
    std::vector<int> trial = parts;
    ++trial[worst];
    if (!pack(trial, nullptr)) break;
    parts = trial;
  }
  // The chain is as long as the longest part (a multiple of 4 bins = whole quads): 28 for 40 bins and 24
  // for 23 bins at 16 kHz (round 3; it used to be padded to 32).  fbank512b_kernel issues an odd quad on its
  // own; fbank512_kernel issues quads in pairs and runs the row of zeros behind the table as the eighth
  // (same sums, four idle instructions on the per-utterance / dithered paths); the two-frames-per-row
  // kernel keeps an even count.
  int chain = 16;
  for (int g = 0; g < n_groups; ++g) chain = std::max(chain, per_part(g, parts[g]));
  chain = dual ? ((chain + 7) & ~7) : ((chain + 3) & ~3);
  p.mm_quads = chain / 4;
  p.mm_levels = 1;
  for (int g = 0; g < n_groups; ++g) p.mm_levels = std::max(p.mm_levels, parts[g]);
  std::vector<int> first_block;
  pack(parts, &first_block);
  struct Block { int group, part, lo, hi, start; };  // FFT bins [lo, hi) of `group`; operands from `start`
  std::vector<Block> blocks(16, Block{-1, 0, 0, 0, 0});
  const int max_start = (260 - chain) & ~3;  // the B operands stay inside the written part of the tile
  for (int g = 0; g < n_groups; ++g) {
    const int per = per_part(g, parts[g]);
    for (int q = 0; q < parts[g]; ++q) {
      Block& bk = blocks[first_block[g] + q];
      bk.group = g;
      bk.part = q;
      bk.lo = std::min(glo[g] + q * per, ghi[g]);
      bk.hi = std::min(glo[g] + (q + 1) * per, ghi[g]);
      bk.start = std::max(0, std::min(bk.lo, max_start)) & ~3;
    }
  }
  // LDS banks of the B reads: a ds_read_b128 is served in groups of 16 lanes - the blocks
  // {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15} - and the four frames of a block already sit
  // on the four 16-byte slots s, s+4, s+8, s+12 (mod 16) of its start: a group is conflict-free when
  // the starts / 4 of its 4 blocks differ modulo 4.  A block shorter than the chain may start up to its
  // slack earlier (leading taps carry zero weights): choose the shifts greedily, least slack first.
  {
    static const int kGroups[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};
    for (const auto& grp : kGroups) {
      bool taken[4] = {false, false, false, false};
      int order[4] = {grp[0], grp[1], grp[2], grp[3]};
      auto slack = [&](int bi) {
        const Block& bk = blocks[bi];
        if (bk.group < 0) return 1 << 20;
        return std::min(bk.start / 4, (bk.start + chain - std::max(bk.hi, bk.start)) / 4);
      };
      std::stable_sort(order, order + 4, [&](int x, int y) { return slack(x) < slack(y); });
      for (int bi : order) {
        Block& bk = blocks[bi];
        if (bk.group < 0) {  // idle block: any free residue
          int r = 0;
          while (r < 3 && taken[r]) ++r;
          bk.start = 4 * r;
          taken[r] = true;
          continue;
        }
        const int sl = slack(bi);
        int best = 0;
        for (int k = 0; k <= sl && k < 4; ++k)
          if (!taken[((bk.start / 4) - k) & 3]) {
            best = k;
            break;
          }
        bk.start -= 4 * best;
        taken[(bk.start / 4) & 3] = true;
      }
    }
  }
  while (blob->size() % 4) blob->push_back(0.0f);  // 16-byte alignment of the float4 tables
  p.off_mm_a = static_cast<int>(blob->size());
  for (int t = 0; t < p.mm_quads + 2; ++t)  // (+ two rows of zeros: the look-ahead reads of the pair loops)
    for (int lane = 0; lane < 64; ++lane)
      for (int c = 0; c < 4; ++c) {
        const Block& bk = blocks[lane >> 2];
        const int m = bk.group < 0 ? -1 : 4 * bk.group + (lane & 3);
        const int k = bk.start + 4 * t + c;  // FFT bin of this tap
        float w = 0.0f;
        if (t < p.mm_quads && m >= 0 && m < mb.num_bins && k >= bk.lo && k < bk.hi && k >= mb.first[m] &&
            k < mb.first[m] + mb.size[m])
          w = 0.25f * mb.w[mb.offset[m] + k - mb.first[m]];  // exact power-of-two scaling
        blob->push_back(w);
      }
  auto push_int = [&](int v) {
    float as_float;
    std::memcpy(&as_float, &v, 4);
    blob->push_back(as_float);
  };
  p.off_mm_lane = static_cast<int>(blob->size());
  for (int lane = 0; lane < 64; ++lane) push_int(blocks[lane >> 2].start);
  for (int lane = 0; lane < 64; ++lane) {
    const Block& bk = blocks[lane >> 2];
    push_int(bk.group >= 0 && bk.part == 0 ? 4 * bk.group : -1);
  }
  for (int level = 1; level <= 3; ++level)
    for (int lane = 0; lane < 64; ++lane) {
      const Block& bk = blocks[lane >> 2];
      blob->push_back(bk.group >= 0 && bk.part == 0 && parts[bk.group] > level ? 1.0f : 0.0f);
    }
  // ---- DCT-II as MFMA blocks: block 4 cg + kp = cepstra 4 cg .. 4 cg + 3 x mel bins of partition kp --
  // The DCT-II of MFCC plans ships on the vector pipe (measured: a tie with the MFMA chain on the
  // 512-point kernel - 1.17-1.21 against 1.19-1.22 ms -, 19 % faster on the two-frames-per-row kernel,
  // which only has that form); SNF_DCT_MFMA=1 selects the chain.  Only the table of the selected form
  // goes into the blob (both would not fit in LDS beside the fused-delta buffer).
  {
    const char* knob = getenv("SNF_DCT_MFMA");
    p.dct_mfma = (!dual && knob && knob[0] == '1') ? 1 : 0;
  }
  p.dd_quads = 0;
  p.off_dd_a = static_cast<int>(blob->size());
  if (mp.kind == SNF_KIND_MFCC && p.dct_mfma) {
    p.dd_quads = ((mp.num_bins + 3) / 4 + 3) / 4;
    const int per = 4 * p.dd_quads;
    for (int t = 0; t < p.dd_quads; ++t)
      for (int lane = 0; lane < 64; ++lane)
        for (int c = 0; c < 4; ++c) {
          const int blk = lane >> 2, cep = 4 * (blk >> 2) + (lane & 3), m = (blk & 3) * per + 4 * t + c;
          blob->push_back(cep < mp.num_ceps && m < mp.num_bins
                              ? dct[static_cast<size_t>(cep) * mp.num_bins + m] : 0.0f);
        }
  }
  p.off_lifter = static_cast<int>(blob->size());
  for (int c = 0; c < 16; ++c)
    blob->push_back(c < static_cast<int>(lifter.size()) ? lifter[c] : 1.0f);
  // DCT-II for the vector pipe, lane-major: group g4 (mel bins 4 g4 .. 4 g4 + 3) x cepstrum l < 16
  while (blob->size() % 4) blob->push_back(0.0f);
  p.off_dd_v = static_cast<int>(blob->size());
  p.dd_groups = 0;
  if (mp.kind == SNF_KIND_MFCC && !p.dct_mfma) {
    p.dd_groups = (mp.num_bins + 3) / 4;
    for (int g4 = 0; g4 < p.dd_groups; ++g4)
      for (int l = 0; l < 16; ++l)
        for (int c = 0; c < 4; ++c) {
          const int m = 4 * g4 + c;
          blob->push_back(l < mp.num_ceps && m < mp.num_bins ? dct[static_cast<size_t>(l) * mp.num_bins + m]
                                                             : 0.0f);
        }
  }
  p.table_floats = static_cast<int>(blob->size());
  {  // header: what the PERUTT kernel needs to know about THIS warp factor's tables
    int hdr[kFastHeaderFloats] = {};
    hdr[0] = p.mm_quads;
    hdr[1] = p.mm_levels;
    hdr[2] = p.dd_quads;
    hdr[3] = p.off_mm_a;
    hdr[4] = p.off_mm_lane;
    hdr[5] = p.off_dd_a;
    hdr[6] = p.off_lifter;
    hdr[7] = p.table_floats;
    hdr[8] = p.off_dd_v;
    hdr[9] = p.dd_groups;
    std::memcpy(blob->data(), hdr, sizeof(hdr));
  }
  // One 16-wave workgroup needs 16 x 4 frame tiles (139 264 bytes) beside the tables in the 160 KB of a
  // CU: wide banks (many bins on zero-extended frames, long DCT tables) that leave no room run on the
  // generic kernel instead of failing at launch (found by tests/tools/fuzz_parity.py).
  if (((static_cast<size_t>(p.table_floats) * 4 + 255) & ~static_cast<size_t>(255)) +
          static_cast<size_t>(kMaxWaves) * 4 * kFrameTileBytes > 160 * 1024)
    return 1;
  *out = p;
  return SNF_OK;
}

int launch_fbank512(const Fast512Params& p, const BatchArgs& b, float* out, int out_cols,
                    double* energy_out, hipStream_t stream) {
  if (b.total_frames <= 0) return SNF_OK;
  if (fbank512b_eligible(p, b)) return launch_fbank512b(p, b, out, out_cols, energy_out, stream);
  Fast512Params q = p;
  q.out_cols = out_cols;
  const bool per_utt = b.blk_utt != nullptr;
  const bool fused = p.fused_delta != 0;  // (MFCC + deltas: always scheduled per utterance)
  if (per_utt && q.table_stride == 0) q.table_stride = (p.table_floats + 3) & ~3;  // (one blob, warp 1.0)
  const int tab_bytes = ((per_utt ? q.table_stride : p.table_floats) * 4 + 255) & ~255;
  int n_waves = 8;
  size_t lds = static_cast<size_t>(tab_bytes) + n_waves * 4 * kFrameTileBytes;
  if (2 * (lds + 512) > 160 * 1024) {  // two 8-wave workgroups do not fit: one of 16 waves
    n_waves = kMaxWaves;
    lds = static_cast<size_t>(tab_bytes) + n_waves * 4 * kFrameTileBytes;
  }
  if (fused) {
    if (!per_utt || p.kind != SNF_KIND_MFCC || p.num_ceps > kFusedCols)
      return set_error(SNF_E_RUNTIME, "fast512: fused deltas need the per-utterance MFCC schedule");
    n_waves = kFusedWaves;
    lds = static_cast<size_t>(tab_bytes) + n_waves * 4 * kFrameTileBytes +
          sizeof(float) * 4 * (kFusedSets + 2) * kFusedCols;
  }
  if (const char* forced = getenv("SNF_FAST512_WAVES")) {  // developer knob: occupancy experiments
    n_waves = atoi(forced);
    lds = static_cast<size_t>(tab_bytes) + n_waves * 4 * kFrameTileBytes;
  }
  if (lds > 160 * 1024) return set_error(SNF_E_RUNTIME, "fast512: tables do not fit in LDS");
  const int energy = p.need_raw ? 1 : (p.need_post ? 2 : 0);
  if (p.dual) {
    // two frames per 16-lane row (fbank256x2_kernel): flat batches of FBANK / MFCC / PLP plans
    if (per_utt || fused) return set_error(SNF_E_RUNTIME, "fast512: the dual tables serve flat batches only");
    if (!b.pair_tab || b.n_pairs <= 0) return set_error(SNF_E_RUNTIME, "fast512: no frame pair table");
    const int64_t n_sets8 = (b.n_pairs + 3) / 4;
    int64_t blocks8 = (n_sets8 + n_waves - 1) / n_waves;
    const int64_t max_blocks8 = 256 * 4 * (kMaxWaves / n_waves);
    if (blocks8 > max_blocks8) blocks8 = max_blocks8;
    const dim3 block8(n_waves * 64);
    auto run_dual = [&](const BatchArgs& b, const dim3 grid8) -> int {
#define SNF_DUAL5(NJ_, KIND_, EN_, DI_, SN_)                                                         \
  do {                                                                                              \
    if (lds > 64 * 1024)                                                                            \
      SNF_HIP_CHECK(hipFuncSetAttribute(                                                            \
          reinterpret_cast<const void*>(fbank256x2_kernel<NJ_, KIND_, EN_, DI_, SN_>),              \
          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                      \
    hipLaunchKernelGGL((fbank256x2_kernel<NJ_, KIND_, EN_, DI_, SN_>), grid8, block8, lds, stream,  \
                       q, b, out, energy_out);                                                      \
  } while (0)
#define SNF_DUAL3(NJ_, KIND_, EN_)                                                                   \
  do {                                                                                              \
    if (p.dither != 0.0f) {                                                                         \
      if (p.snip_edges) SNF_DUAL5(NJ_, KIND_, EN_, true, true);                                     \
      else SNF_DUAL5(NJ_, KIND_, EN_, true, false);                                                 \
    } else {                                                                                        \
      if (p.snip_edges) SNF_DUAL5(NJ_, KIND_, EN_, false, true);                                    \
      else SNF_DUAL5(NJ_, KIND_, EN_, false, false);                                                \
    }                                                                                               \
  } while (0)
#define SNF_DUAL(NJ_, KIND_)                                                                         \
  do {                                                                                              \
    if (energy == 0) SNF_DUAL3(NJ_, KIND_, 0);                                                      \
    else if (energy == 1) SNF_DUAL3(NJ_, KIND_, 1);                                                 \
    else SNF_DUAL3(NJ_, KIND_, 2);                                                                  \
  } while (0)
    // (the 13-row form fetches spans of up to 384 samples, the 16-row form of up to 512: fast512_dual_eligible)
    if ((p.win_len + 15) / 16 == 13 && p.win_len + p.win_shift <= 384) {
      if (p.kind == SNF_KIND_FBANK) SNF_DUAL(13, SNF_KIND_FBANK);
      else if (p.kind == SNF_KIND_MFCC) SNF_DUAL(13, SNF_KIND_MFCC);
      else if (p.kind == SNF_KIND_SPECTROGRAM) SNF_DUAL(13, SNF_KIND_SPECTROGRAM);
      else SNF_DUAL(13, SNF_KIND_PLP);
    } else {
      if (p.kind == SNF_KIND_FBANK) SNF_DUAL(16, SNF_KIND_FBANK);
      else if (p.kind == SNF_KIND_MFCC) SNF_DUAL(16, SNF_KIND_MFCC);
      else if (p.kind == SNF_KIND_SPECTROGRAM) SNF_DUAL(16, SNF_KIND_SPECTROGRAM);
      else SNF_DUAL(16, SNF_KIND_PLP);
    }
#undef SNF_DUAL
#undef SNF_DUAL3
#undef SNF_DUAL5
      SNF_HIP_CHECK(hipGetLastError());
      return SNF_OK;
    };
    int rc = run_dual(b, dim3(static_cast<unsigned>(blocks8)));
    if (rc != SNF_OK || b.fix_tab == nullptr) return rc;
    // the fix-up launch: the frames of the pairs the first launch put down (two records of one frame each), as
    // many as `fix_count` says when the first launch is done - a grid sized for a twentieth of the pairs walks
    // whatever the list holds
    BatchArgs bf = b;
    bf.pair_tab = b.fix_tab;
    bf.n_pairs_dev = b.fix_count;
    bf.fix_tab = nullptr;
    bf.fix_count = nullptr;
    int64_t blocks_fix = (blocks8 + 9) / 10;
    if (blocks_fix < 16) blocks_fix = blocks8 < 16 ? blocks8 : 16;
    return run_dual(bf, dim3(static_cast<unsigned>(blocks_fix)));
  }
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
    if (fused && KIND_ == SNF_KIND_MFCC)                                                            \
      SNF_LAUNCH6(NJ_, KIND_, EN_, DI_, SN_, (KIND_ == SNF_KIND_MFCC ? 2 : 0));                     \
    else if (per_utt && KIND_ != SNF_KIND_SPECTROGRAM && KIND_ != SNF_KIND_ENERGY)                  \
      SNF_LAUNCH6(NJ_, KIND_, EN_, DI_, SN_,                                                        \
                  ((KIND_ != SNF_KIND_SPECTROGRAM && KIND_ != SNF_KIND_ENERGY) ? 1 : 0));           \
    else SNF_LAUNCH6(NJ_, KIND_, EN_, DI_, SN_, 0);                                                 \
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
