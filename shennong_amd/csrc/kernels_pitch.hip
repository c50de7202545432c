// Kaldi pitch tracker on gfx950: the reference reaches it through
// kaldi.feat.pitch.compute_kaldi_pitch (shennong/processor/pitch_kaldi.py:296-299).
//
// Restates [KALDI-UPSTREAM] pitch-functions.cc (OnlinePitchFeatureImpl::AcceptWaveform /
// InputFinished / RecomputeBacktraces, ComputeCorrelation, ComputeNccf, ComputeLocalCost,
// PitchFrameInfo::ComputeBacktraces) and resample.cc (LinearResample, ArbitraryResample) for the
// offline single-chunk call the reference makes (frames_per_chunk = 0).
//
// Four launches per batch:
//   1. pitch_resample_kernel  one thread per downsampled sample: 16 kHz -> 4 kHz windowed-sinc FIR (a
//                             workgroup filters 4 chunks of 256 samples, the next chunk's input in flight)
//   2. pitch_stats_kernel     one workgroup per utterance: signal sum / sum of squares, from which
//                             one thread derives the NCCF ballasts of the utterance
//   3. pitch_nccf_kernel      FRAME-parallel (nothing in it depends on the previous frame): wave64 =
//                             4 frames x 16 lanes; lane l correlates the lags 5 l .. 5 l + 4 against
//                             the frame's window in LDS (5 x 5 register blocks), NCCF with and
//                             without ballast, then the NCCF is resampled to the log-spaced lags of the
//                             Viterbi states on the matrix pipe (v_mfma_f32_4x4x1: 4 states x 4 frames
//                             per block, one exact fused multiply-add per tap) -> [frames, states] in HBM
//   4. pitch_viterbi_kernel   the only sequential part: one wavefront per utterance walks the frames,
//                             local cost from the row of 3, exact argmin of the transition cost with
//                             the monotone divide-and-conquer search, backpointers to HBM, traceback.
//
// Arithmetic contract.  Kaldi hands the sums of this algorithm to BLAS, whose summation order depends
// on the library build; the CPU oracle (oracle/kaldi_oracle.c, chain_dot / tree16) fixes ONE order and
// these kernels implement exactly that order, so the tracker agrees with the oracle bit for bit and the
// Viterbi paths are identical (a one-ulp difference in a cost flips near-ties in unvoiced regions):
//   - FIR taps, lag correlations, sinc resampling: s = fmaf(a[i], b[i], s), i ascending, from 0 (the
//     matrix-pipe form of the sinc resampling is the same chain: K = 1 per instruction, lags ascending,
//     zero weights outside a state's taps);
//   - frame mean / frame energy / norm average: lane l of 16 sums its elements l, l + 16, ...
//     ascending, the 16 partial sums are added as a balanced tree of neighbours (DPP quad_perm xor 1,
//     xor 2, row_half_mirror, row_mirror: every lane ends with the same bits);
//   - divisions and square roots are IEEE correctly rounded; nothing else is fused: this file is
//     compiled with -ffp-contract=off and every fused multiply-add is written as one.
#include <float.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "snf_internal.h"
#include "device_fft.h"

namespace snf {

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace

// ---- 1. LinearResample ---------------------------------------------------------------------------
// A workgroup computes 256 consecutive output samples of one utterance: the input span they touch
// (256 x in_unit / out_unit samples + one filter length) is staged into LDS as floats with coalesced
// loads - zeros outside the utterance: fmaf(w, 0, s) = s exactly, the accumulator never being -0 - and
// every thread then runs Kaldi's tap loop (one sequential fmaf chain) from LDS.  The per-output gather
// of 2-byte samples from global memory that this replaces took 0.68 ms per 192 M input samples.
constexpr int kRsChunks = 4;   // 256-output chunks per workgroup of the resampler
constexpr int kRsMaxPer = 6;   // staged samples per thread and chunk that travel through registers

__global__ __launch_bounds__(256) void pitch_resample_kernel(const PitchDevTables t, const PitchBatch b,
                                                             float* __restrict__ down) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);
  // blockIdx.y = utterance (no per-thread search), blockIdx.x = kRsChunks consecutive 256-sample chunks of
  // its output.  A workgroup that handles one chunk is a chain of four dependent memory round trips
  // (offsets, samples, taps, store: 4 us for 250 instructions per wave, 0.375 ms per 192 M samples);
  // here the offsets are read once and the samples of chunk c + 1 are in flight, in registers, while
  // chunk c is filtered.
  const int64_t u = blockIdx.y;
  const int64_t d0 = b.down_offsets[u], nd = b.down_offsets[u + 1] - d0;
  const int64_t k_begin = static_cast<int64_t>(blockIdx.x) * (256 * kRsChunks);
  if (k_begin >= nd) return;
  const int64_t s0 = b.sample_offsets[u], n = b.sample_offsets[u + 1] - s0;
  const int16_t* __restrict__ w = b.wave + s0;
  // (32-bit index arithmetic: a 64-bit division costs ~100 instructions, three of them per output was
  // more than the filter itself; an utterance has fewer than 2^31 samples)
  const int out_unit = t.rs_out_unit, in_unit = t.rs_in_unit;
  const int tid = static_cast<int>(threadIdx.x);
  auto first_of = [&](int k) -> int64_t {
    const int unit = k / out_unit;
    return t.rs_first[k - unit * out_unit] + static_cast<int64_t>(unit) * in_unit;
  };
  // input span of the chunk that starts at output k0: [base, base + span)
  auto geometry = [&](int64_t k0, int64_t* base, int* span) {
    const int k_first = static_cast<int>(k0);
    const int k_last = k0 + 255 < nd - 1 ? k_first + 255 : static_cast<int>(nd) - 1;
    *base = first_of(k_first);                               // (first inputs are non-decreasing in k)
    *span = static_cast<int>(first_of(k_last) + t.rs_max_taps - *base);
  };
  auto sample = [&](int64_t j) -> float { return (j >= 0 && j < n) ? static_cast<float>(w[j]) : 0.0f; };
  // samples in flight: raw 16-bit values from clamped (always valid) addresses, no branch and no
  // conversion until they are staged - a conversion here would wait for every load on the spot
  auto fetch = [&](int64_t j) -> int {
    const int64_t jc = j < 0 ? 0 : (j < n ? j : n - 1);
    return w[jc];
  };
  int pre[kRsMaxPer];
  int64_t base;
  int span;
  geometry(k_begin, &base, &span);
#pragma unroll
  for (int q = 0; q < kRsMaxPer; ++q) pre[q] = fetch(base + tid + 256 * q);
  for (int c = 0; c < kRsChunks; ++c) {
    const int64_t k0 = k_begin + 256 * c;
    if (k0 >= nd) break;
    // stage the chunk: zeros outside the utterance (fmaf(w, 0, s) = s exactly, s never being -0)
#pragma unroll
    for (int q = 0; q < kRsMaxPer; ++q) {
      const int64_t j = base + tid + 256 * q;
      if (tid + 256 * q < span) xs[tid + 256 * q] = (j >= 0 && j < n) ? static_cast<float>(pre[q]) : 0.0f;
    }
    for (int i = tid + 256 * kRsMaxPer; i < span; i += 256) xs[i] = sample(base + i);  // (rate ratios above 5)
    __syncthreads();
    const int64_t base_cur = base;
    if (c + 1 < kRsChunks && k0 + 256 < nd) {
      geometry(k0 + 256, &base, &span);
#pragma unroll
      for (int q = 0; q < kRsMaxPer; ++q) pre[q] = fetch(base + tid + 256 * q);
    }
    const int k = static_cast<int>(k0) + tid;
    if (k < nd) {
      float s = 0.0f;
      if (out_unit == 1) {
        // one filter for every output (integer rate ratios, e.g. 16 kHz -> 4 kHz): uniform weights.  (A
        // de-interleaved LDS layout that makes the tap reads conflict-free was measured: slower, 0.63 ms
        // against 0.375 - the strided staging costs more than the conflicts)
        const float* __restrict__ x = xs + (t.rs_first[0] + static_cast<int64_t>(k) * in_unit - base_cur);
        const int ntaps = t.rs_ntaps[0];
        for (int i = 0; i < ntaps; ++i) s = __builtin_fmaf(t.rs_w[i], x[i], s);
      } else {
        const int unit = k / out_unit, wrapped = k - unit * out_unit;
        const float* __restrict__ x =
            xs + (t.rs_first[wrapped] + static_cast<int64_t>(unit) * in_unit - base_cur);
        const float* __restrict__ wt = t.rs_w + wrapped * t.rs_max_taps;
        const int ntaps = t.rs_ntaps[wrapped];
        for (int i = 0; i < ntaps; ++i) s = __builtin_fmaf(wt[i], x[i], s);
      }
      down[d0 + k] = s;
    }
    __syncthreads();  // (the next chunk overwrites the staged span)
  }
}

// ---- 2. signal statistics for the NCCF ballast -----------------------------------------------------
// Kaldi accumulates the float BLAS dot / sum of each chunk into doubles; phase 1 = what the resampler
// emitted before the flush.  ub[u] = {ballast of the frames of phase 1, of phase 2, the two "old"
// ballasts RecomputeBacktraces derives from the float mean squares, the new ballast, 1 if
// RecomputeBacktraces has to run (utterance shorter than recompute_frame whose phase-1 mean square is
// more than 1 % away from the final one)}
__global__ void pitch_stats_kernel(const PitchDevTables t, const PitchBatch b,
                                   const float* __restrict__ down, float* __restrict__ ub) {
  const int64_t u = blockIdx.x;
  const int64_t d0 = b.down_offsets[u], nd = b.down_offsets[u + 1] - d0, nd1 = b.down_phase1[u];
  const float* __restrict__ x = down + d0;
  double sq1 = 0, s1 = 0, sq2 = 0, s2 = 0;
  for (int64_t i = threadIdx.x; i < nd; i += blockDim.x) {
    const double v = x[i];
    if (i < nd1) { sq1 += v * v; s1 += v; } else { sq2 += v * v; s2 += v; }
  }
  __shared__ double red[4][16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  sq1 = wave_sum_d(sq1); s1 = wave_sum_d(s1); sq2 = wave_sum_d(sq2); s2 = wave_sum_d(s2);
  if (lane == 0) { red[0][wid] = sq1; red[1][wid] = s1; red[2][wid] = sq2; red[3][wid] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0, c = 0, d = 0, e = 0;
    for (int i = 0; i < nw; ++i) { a += red[0][i]; c += red[1][i]; d += red[2][i]; e += red[3][i]; }
    // each chunk's BLAS result is a float that is then added to a double accumulator
    const double fsq1 = static_cast<float>(a), fs1 = static_cast<float>(c);
    const double sumsq2 = fsq1 + static_cast<double>(static_cast<float>(d));
    const double sum2 = fs1 + static_cast<double>(static_cast<float>(e));
    const double n1 = static_cast<double>(nd1), n2 = static_cast<double>(nd);
    const double m1 = nd1 > 0 ? fs1 / n1 : 0.0, m2 = nd > 0 ? sum2 / n2 : 0.0;
    const double ms1 = nd1 > 0 ? fsq1 / n1 - m1 * m1 : 0.0;
    const double ms2 = nd > 0 ? sumsq2 / n2 - m2 * m2 : 0.0;
    const double W = t.win_size, bal = t.nccf_ballast;
    float* __restrict__ o = ub + u * 6;
    o[0] = static_cast<float>((ms1 * W) * (ms1 * W) * bal);
    o[1] = static_cast<float>((ms2 * W) * (ms2 * W) * bal);
    const float f1 = static_cast<float>(ms1), f2 = static_cast<float>(ms2);
    o[2] = static_cast<float>((static_cast<double>(f1) * W) * (static_cast<double>(f1) * W) * bal);
    o[3] = static_cast<float>((static_cast<double>(f2) * W) * (static_cast<double>(f2) * W) * bal);
    o[4] = o[3];
    // ApproxEqual(a, b, 0.01): |a - b| <= 0.01 (|a| + |b|); frames of phase 2 compare equal
    const bool differ = b.frames_phase1[u] > 0 && !(fabsf(f1 - f2) <= 0.01f * (fabsf(f1) + fabsf(f2)));
    o[5] = differ ? 1.0f : 0.0f;
  }
}

namespace {

// Ordering point between the phases of ONE wave over LDS that only this wave touches (every wave of the
// Viterbi / NCCF kernels owns its slice; the one shared table is written before the __syncthreads of the
// prologue): the LDS executes a wave's instructions in issue order, so only the compiler must keep the
// accesses in program order - wavefront-scope fences emit no instruction, where workgroup scope costs an
// s_waitcnt lgkmcnt(0) (the wave sits until every store is acknowledged) at each of the ~12 points per frame
// (round 4; SNF_PITCH_WG_FENCE restores the old form at build time for A/B runs)
__device__ __forceinline__ void wave_sync() {
#ifdef SNF_PITCH_WG_FENCE
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// Cross-lane steps on the VALU (DPP + v_readlane); __shfl_xor lowers to ds_bpermute_b32, an LDS round
// trip per step.  quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E; row_ror:4 = 0x124, row_ror:8 = 0x128
// (rotations reach the other three quads of a 16-lane row).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf,
                                                               0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
// a value every lane of the wave holds, moved to scalar registers
__device__ __forceinline__ int64_t uniform_i64(int64_t v) {
  const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(v))));
  const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint64_t>(v) >> 32)));
  return static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo);
}
__device__ __forceinline__ float lane_f(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// sum over the wave: DPP all-reduce inside the 16-lane rows, then the four row values (read with
// v_readlane) are combined uniformly, so every lane gets the same bits
__device__ __forceinline__ float wave_sum_v(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x124>(v);
  v += dpp_f<0x128>(v);
  return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float wave_min_f(float v) {
  v = fminf(v, dpp_f<0xB1>(v));
  v = fminf(v, dpp_f<0x4E>(v));
  v = fminf(v, dpp_f<0x124>(v));
  v = fminf(v, dpp_f<0x128>(v));
  return fminf(fminf(lane_f(v, 0), lane_f(v, 16)), fminf(lane_f(v, 32), lane_f(v, 48)));
}
// (cost, index) argmin step: lower cost wins, lower index on ties.  Costs are non-negative floats and indices
// non-negative ints, so that is the unsigned order of (cost bits << 32 | index): one 64-bit compare and two selects
// (the float / int form compiled to four compares and two exec-masked blocks per step)
__device__ __forceinline__ void argmin_take(float& c, int& j, float oc, int oj) {
  const unsigned long long mine = (static_cast<unsigned long long>(__builtin_bit_cast(unsigned, c)) << 32) | static_cast<unsigned>(j);
  const unsigned long long other = (static_cast<unsigned long long>(__builtin_bit_cast(unsigned, oc)) << 32) | static_cast<unsigned>(oj);
  const bool take = other < mine;
  c = take ? oc : c;
  j = take ? oj : j;
}
__device__ __forceinline__ void quad_argmin(float& c, int& j) {
  argmin_take(c, j, dpp_f<0xB1>(c), dpp_i<0xB1>(j));
  argmin_take(c, j, dpp_f<0x4E>(c), dpp_i<0x4E>(j));
}
__device__ __forceinline__ void wave_argmin(float& c, int& j) {
  quad_argmin(c, j);
  argmin_take(c, j, dpp_f<0x124>(c), dpp_i<0x124>(j));
  argmin_take(c, j, dpp_f<0x128>(c), dpp_i<0x128>(j));
  float bc = lane_f(c, 0);
  int bj = __builtin_amdgcn_readlane(j, 0);
  argmin_take(bc, bj, lane_f(c, 16), __builtin_amdgcn_readlane(j, 16));
  argmin_take(bc, bj, lane_f(c, 32), __builtin_amdgcn_readlane(j, 32));
  argmin_take(bc, bj, lane_f(c, 48), __builtin_amdgcn_readlane(j, 48));
  c = bc;
  j = bj;
}

// cost of reaching state i from state j: must round exactly like Kaldi's
// (j - i)^2 * inter_frame_factor + prev_forward_cost[j] (no FMA contraction)
__device__ __forceinline__ float trans_cost(int j, float fi, float factor, float fwd_j) {
  const float d = static_cast<float>(j) - fi;
  return __fadd_rn(__fmul_rn(d * d, factor), fwd_j);
}

// exact argmin over j in [lo, hi] (lowest index wins ties), 4 candidates in flight per step.  The
// last step may look at up to 3 states beyond `hi`: the argmin is monotone in i, so none of them can
// beat the optimum inside the range (an equal cost loses to the lower index), and the forward costs
// are padded with FLT_MAX beyond the last state.
__device__ __forceinline__ void scan_range(const float* __restrict__ fwd, int lo, int hi, float fi,
                                           float factor, float& best, int& best_j) {
  // The running value is d = j - i (small integers: exact in float, so d equals Kaldi's j - i whichever way
  // it is reached); per candidate: one add, two multiplies, one add, one compare, one select (the best d)
  // and one minimum (costs are >= 0 and never NaN: the minimum IS the select of the cost).
  float d = static_cast<float>(lo) - fi;
  float b = __fadd_rn(__fmul_rn(d * d, factor), fwd[lo]), bd = d;
  // the four forward costs of a step are read one step ahead (behind the window: the FLT_MAX padding): the
  // LDS round trip of step n + 1 runs under the 28 vector instructions of step n instead of in front of them
  float n0 = fwd[lo + 1], n1 = fwd[lo + 2], n2 = fwd[lo + 3], n3 = fwd[lo + 4];
  for (int j = lo + 1; j <= hi; j += 4) {
    const float f0 = n0, f1 = n1, f2 = n2, f3 = n3;
    n0 = fwd[j + 4];
    n1 = fwd[j + 5];
    n2 = fwd[j + 6];
    n3 = fwd[j + 7];
    const float d0 = d + 1.0f, d1 = d + 2.0f, d2 = d + 3.0f, d3 = d + 4.0f;
    const float c0 = __fadd_rn(__fmul_rn(d0 * d0, factor), f0);
    const float c1 = __fadd_rn(__fmul_rn(d1 * d1, factor), f1);
    const float c2 = __fadd_rn(__fmul_rn(d2 * d2, factor), f2);
    const float c3 = __fadd_rn(__fmul_rn(d3 * d3, factor), f3);
    bd = c0 < b ? d0 : bd;
    b = fminf(b, c0);
    bd = c1 < b ? d1 : bd;
    b = fminf(b, c1);
    bd = c2 < b ? d2 : bd;
    b = fminf(b, c2);
    bd = c3 < b ? d3 : bd;
    b = fminf(b, c3);
    d = d3;
  }
  best = b;
  best_j = static_cast<int>(fi + bd);
}


// sum of the 16 values of a DPP row as a balanced tree of neighbours; every lane gets the same bits
__device__ __forceinline__ float tree16(float v) {
  v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_f<0x141>(v);  // row_half_mirror: quad 0 <-> 1, 2 <-> 3
  v += dpp_f<0x140>(v);  // row_mirror: half 0 <-> 1
  return v;
}

constexpr int kLagGroup = 5;   // lags per lane and pass of the correlation
constexpr int kNccfWaves = 12;  // 48 frames per workgroup: two workgroups per CU (LDS) = 6 waves per SIMD
constexpr int kFwdPad = 36;    // FLT_MAX entries behind the forward costs (unclamped scan steps)
constexpr int kLongRange3 = 32;  // level 3 / levels 4-5: windows of at least this many candidates go to the
constexpr int kLongRange4 = 12;  // 8-lane teams instead of one lane (flat between 8 and 32: measured)
constexpr bool kSplitLevel5 = false;  // refine 4 -> 2 -> 1 instead of 4 -> 1: 27 % fewer candidates, one more level; measured slower (6.41 against 6.22 ms)
constexpr bool kTracePrefetch = true; // traceback: touch the backpointer rows of a 64-frame chunk first
constexpr int kQueueEntries = 128;                // long-window queue of a wave: int4 (state, lo, hi, -)
constexpr int kQueueFloats = kQueueEntries * 4;
constexpr int kTeam4Utts = 768;    // batches up to this many utterances: four waves per utterance,
constexpr int kTeam2Utts = 1280;   // up to this many: two (tools/time_pitch.py: 250 / 500 / 1 000 / 1 500
                                   // utterances take 3.20 / 3.29 / 3.46 / 3.61 ms with one wave, 2.40 / 2.63 /
                                   // 3.06 / 3.69 with two, 2.23 / 2.43 / 3.20 / 4.14 with four)
constexpr int kVitWaves = 8;   // utterances (= wavefronts) per workgroup of the Viterbi kernel

// LDS layout of one frame of an NCCF wave: the window (wl floats), then the NCCF at the integer lags.
// When one pass of the correlation covers every lag (num_lags <= 80: the window is dead by the time the
// NCCF is written) the NCCF overlays the head of the window.  The frames of a wave sit 16 mod 32 floats
// apart, so that the two 16-lane rows of a ds_read_b32 lane group (banks = dword address mod 32) never
// meet in the correlation - lane l of a row reads dword base + 5 l (banks 0, 5, .., 30, 3, .., 28, 1, 6,
// 11) and that set shifted by 16 is its complement -, and frames 2 and 3 keep their NCCF 8 floats
// further in, which spreads the four frames of the matrix-pipe resampler over the banks 0, 16, 8, 24.
struct NccfLayout {
  int nccf_off;  // floats from the window to the NCCF of frames 0, 1 (frames 2, 3: + 8)
  int pitch;     // floats between frames
};
__host__ __device__ inline NccfLayout nccf_layout(int wl, int ln, int num_lags) {
  NccfLayout y;
  const bool overlay = num_lags <= kLagGroup * 16 && ln + 8 <= wl;
  y.nccf_off = overlay ? 0 : wl;
  const int n = overlay ? wl : wl + ln + 8;
  y.pitch = n + ((16 - n % 32) + 32) % 32;
  return y;
}

}  // namespace

// one thread per frame: where its window starts in the resampled batch, which of the window's samples exist, and
// the frame's NCCF ballast - 16 bytes that the NCCF kernel reads with ONE load per frame (it used to walk
// frame -> utterance -> offsets -> samples: three dependent trips to HBM at the head of every set of frames,
// half of its wave cycles, profiles/r05_pmc_pitch10k_after_summary.txt)
__global__ void pitch_frame_meta_kernel(const PitchDevTables t, const PitchBatch b, const float* __restrict__ ub,
                                        int4* __restrict__ frame_meta) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= b.total_frames) return;
  const int64_t u = find_utt(b.frame_offsets, b.n_utts, g);
  const int64_t frame = g - b.frame_offsets[u];
  const int64_t d0 = b.down_offsets[u], nd = b.down_offsets[u + 1] - d0;
  int64_t start;
  if (t.snip_edges) start = frame * t.win_shift;
  else start = static_cast<int64_t>((static_cast<double>(frame) + 0.5) * t.win_shift) - t.full_len / 2;
  // window sample i is signal sample start + i: it exists for lo <= i < hi (everything else reads as zero)
  const int64_t lo = start < 0 ? (-start < t.full_len ? -start : t.full_len) : 0;
  const int64_t room = nd - start;
  const int64_t hi = room < 0 ? 0 : (room < t.full_len ? room : t.full_len);
  const int64_t first = d0 + start;   // (may lie in front of the batch for a centred first frame: masked by lo)
  const float ballast = frame < b.frames_phase1[u] ? ub[u * 6 + 0] : ub[u * 6 + 1];
  frame_meta[g] = make_int4(static_cast<int>(static_cast<uint32_t>(first)), static_cast<int>(first >> 32),
                            static_cast<int>(lo | (hi << 16)), __builtin_bit_cast(int, ballast));
}

// ---- 3. NCCF at the integer lags, resampled to the lags of the Viterbi states ------------------------
__global__ __launch_bounds__(kNccfWaves * 64, 6) void pitch_nccf_kernel(
    const PitchDevTables t, const PitchBatch b, const float* __restrict__ down,
    const int4* __restrict__ frame_meta, float* __restrict__ nccf_res,
    float* __restrict__ pov_nccf, float* __restrict__ anp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = t.num_states, L = t.num_lags, W = t.win_size;
  const int KQ = t.ar_quad_taps >> 2, G = t.ar_groups;
  // sinc taps of every state quad in matrix-pipe order (PitchDevTables), shared by the workgroup
  float4* quad_w = reinterpret_cast<float4*>(smem);
  int* quad_base = reinterpret_cast<int*>(quad_w + static_cast<size_t>(G) * KQ * 64);
  for (int i = threadIdx.x; i < G * KQ * 64; i += blockDim.x)
    quad_w[i] = reinterpret_cast<const float4*>(t.ar_quad_w)[i];
  for (int i = threadIdx.x; i < G * 16; i += blockDim.x) quad_base[i] = t.ar_quad_base[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int l = lane & 15, q = lane >> 4;
  const int WL = (t.full_len + 16 + 3) & ~3, LN = (L + t.ar_quad_taps + 3) & ~3;
  const NccfLayout lay = nccf_layout(WL, LN, L);
  float* frames0 = reinterpret_cast<float*>(quad_base + G * 16) + wid * 4 * lay.pitch;
  float* win = frames0 + q * lay.pitch;
  float* nccf = win + lay.nccf_off + (q >> 1) * 8;
  // matrix-pipe view of the wave: lane 4 b + j -> state quad b of a group (A operand: state 4 b + j),
  // frame j of the set (B operand and the four results of the lane)
  const int mb = lane >> 2, mj = lane & 3;
  const float* __restrict__ nccf_m = frames0 + mj * lay.pitch + lay.nccf_off + (mj >> 1) * 8;
  const int64_t n_sets = (b.total_frames + 3) >> 2;
  for (int64_t set = static_cast<int64_t>(blockIdx.x) * kNccfWaves + wid; set < n_sets;
       set += static_cast<int64_t>(gridDim.x) * kNccfWaves) {
    const int64_t g = set * 4 + q;
    const bool valid = g < b.total_frames;
    const int4 meta = frame_meta[valid ? g : b.total_frames - 1];
    const float ballast = __builtin_bit_cast(float, meta.w);
    // ---- window (zero beyond the signal and in the read-ahead padding), mean removal, e1 -----------
    wave_sync();
    {
      const float* __restrict__ x = down + ((static_cast<int64_t>(meta.y) << 32) | static_cast<uint32_t>(meta.x));
      // (hi < lo never happens: both are clamped to [0, full_len]; hi == lo: no sample of the window exists)
      const unsigned lo = static_cast<unsigned>(meta.z) & 0xffffu, span = (static_cast<unsigned>(meta.z) >> 16) - lo;
      for (int i = l; i < WL; i += 16) win[i] = static_cast<unsigned>(i) - lo < span ? x[i] : 0.0f;
    }
    float part = 0.0f;
    for (int i = l; i < W; i += 16) part += win[i];
    const float neg_mean = -tree16(part) / static_cast<float>(W);
    for (int i = l; i < t.full_len; i += 16) win[i] += neg_mean;
    float pe = 0.0f;
    for (int i = l; i < W; i += 16) pe = __builtin_fmaf(win[i], win[i], pe);
    const float e1 = tree16(pe);
    wave_sync();
    // ---- lag correlation: 5 x 5 (sample, lag) blocks out of 9 window values in registers ---------
    float pnorm = 0.0f;
    for (int g0 = 0; kLagGroup * 16 * g0 < L; ++g0) {
      const int lb = kLagGroup * (l + 16 * g0);
      float e2[kLagGroup], ip[kLagGroup];
#pragma unroll
      for (int d = 0; d < kLagGroup; ++d) e2[d] = ip[d] = 0.0f;
      if (lb < L) {
        const float* __restrict__ a = win;
        const float* __restrict__ cw = win + t.first_lag + lb;
        // blocks of 5 samples x 5 lags out of 9 window values; the 10 LDS values of the next block are
        // requested before the 50 multiply-adds of the current one (two register sets, A and B; the
        // reads past the last block land in the window's zero padding and are dropped)
        float c[kLagGroup - 1];
#pragma unroll
        for (int k = 0; k < kLagGroup - 1; ++k) c[k] = cw[k];
        float av_a[kLagGroup], cn_a[kLagGroup], av_b[kLagGroup], cn_b[kLagGroup];
        auto fetch = [&](float (&av)[kLagGroup], float (&cn)[kLagGroup], int at) {
#pragma unroll
          for (int k = 0; k < kLagGroup; ++k) {
            av[k] = a[at + k];
            cn[k] = cw[at + kLagGroup - 1 + k];
          }
        };
        auto block = [&](const float (&av)[kLagGroup], const float (&cn)[kLagGroup]) {
          float cc[2 * kLagGroup - 1];
#pragma unroll
          for (int k = 0; k < kLagGroup - 1; ++k) cc[k] = c[k];
#pragma unroll
          for (int k = 0; k < kLagGroup; ++k) cc[kLagGroup - 1 + k] = cn[k];
#pragma unroll
          for (int k = 0; k < kLagGroup; ++k)
#pragma unroll
            for (int d = 0; d < kLagGroup; ++d) {
              ip[d] = __builtin_fmaf(av[k], cc[k + d], ip[d]);
              e2[d] = __builtin_fmaf(cc[k + d], cc[k + d], e2[d]);
            }
#pragma unroll
          for (int k = 0; k < kLagGroup - 1; ++k) c[k] = cc[kLagGroup + k];
        };
        fetch(av_a, cn_a, 0);
        int i = 0;
        for (; i + 2 * kLagGroup <= W; i += 2 * kLagGroup) {
          fetch(av_b, cn_b, i + kLagGroup);
          block(av_a, cn_a);
          fetch(av_a, cn_a, i + 2 * kLagGroup);
          block(av_b, cn_b);
        }
        if (i + kLagGroup <= W) {
          block(av_a, cn_a);
          i += kLagGroup;
        }
        for (; i < W; ++i) {  // windows that are not a multiple of 5 samples
          const float ai = a[i];
#pragma unroll
          for (int d = 0; d < kLagGroup; ++d) {
            const float vc = cw[i + d];
            ip[d] = __builtin_fmaf(ai, vc, ip[d]);
            e2[d] = __builtin_fmaf(vc, vc, e2[d]);
          }
        }
      }
#pragma unroll
      for (int d = 0; d < kLagGroup; ++d) {
        const int lag = lb + d;
        if (lag < L) {
          const float norm = e1 * e2[d];
          const float den = sqrtf(norm + ballast);
          nccf[lag] = den != 0.0f ? ip[d] / den : 0.0f;
          // the POV feature uses the NCCF without ballast; only the lags around the state chosen by
          // the traceback are read back
          const float den0 = sqrtf(norm);
          if (valid) pov_nccf[g * L + lag] = den0 != 0.0f ? ip[d] / den0 : 0.0f;
          pnorm += norm;
        }
      }
    }
    const float avg_norm_prod = tree16(pnorm) / static_cast<float>(L);  // (RecomputeBacktraces)
    if (valid && l == 0) anp[g] = avg_norm_prod;
    for (int i = L + l; i < LN; i += 16) nccf[i] = 0.0f;  // (taps are zero padded)
    wave_sync();
    // ---- ArbitraryResample: NCCF at the lag of every state, on the matrix pipe.  One
    // v_mfma_f32_4x4x1_16b_f32 = 16 blocks of (4 states) x (4 frames) += w[state][lag] * nccf[frame][lag]:
    // a fused multiply-add per element, lags ascending = Kaldi's tap order (zero weights in front of
    // and behind a state's taps leave its sum unchanged); two groups of 64 states run as independent
    // chains.  12 steps per group instead of 27 x 12 multiply-adds (+ as many LDS reads) per lane. ------
    {
      const int64_t gm = set * 4 + mj;
      const bool valid_m = gm < b.total_frames;
      float* __restrict__ row = nccf_res + gm * static_cast<int64_t>(S);
      // (the usual 12-lag quad window as straight-line code: a loop over a run-time step count makes the
      // register allocator rotate the two accumulators through v_accvgpr moves every iteration)
      auto groups = [&](auto kq_const) {
        constexpr int kKq = decltype(kq_const)::value;
        const int kq = kKq > 0 ? kKq : KQ;
        for (int g2 = 0; g2 < G; g2 += 2) {
          const float* __restrict__ x0 = nccf_m + quad_base[g2 * 16 + mb];
          const float* __restrict__ x1 = nccf_m + quad_base[g2 * 16 + 16 + mb];
          const float4* __restrict__ w0 = quad_w + static_cast<size_t>(g2) * kq * 64 + lane;
          const float4* __restrict__ w1 = w0 + kq * 64;
          f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
          for (int k4 = 0; k4 < kq; ++k4) {
            const float4 a0 = w0[k4 * 64], a1 = w1[k4 * 64];
            const float b00 = x0[4 * k4], b01 = x0[4 * k4 + 1], b02 = x0[4 * k4 + 2], b03 = x0[4 * k4 + 3];
            const float b10 = x1[4 * k4], b11 = x1[4 * k4 + 1], b12 = x1[4 * k4 + 2], b13 = x1[4 * k4 + 3];
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, b00, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, b10, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, b01, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, b11, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, b02, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.z, b12, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, b03, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.w, b13, acc1, 0, 0, 0);
          }
          if (valid_m) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int s0 = 64 * (g2 + h) + 4 * mb;
              const f32x4 acc = h ? acc1 : acc0;
              if (s0 + 4 <= S) {
                *reinterpret_cast<f32x4_a4*>(row + s0) = f32x4_a4{acc[0], acc[1], acc[2], acc[3]};
              } else {
#pragma unroll
                for (int i = 0; i < 3; ++i)
                  if (s0 + i < S) row[s0 + i] = acc[i];
              }
            }
          }
        }
      };
      if (KQ == 3) groups(std::integral_constant<int, 3>{});
      else groups(std::integral_constant<int, 0>{});
    }
  }
}

// ---- 4. Viterbi: one wavefront per utterance ----------------------------------------------------------
namespace {

struct VitShared {
  float* fwd;    // [num_states + kFwdPad]
  float* nxt;    // [num_states]
  int* bpw;      // [num_states]   backpointers of the current frame
  int4* queue;   // [64]           (state, lo, hi) of the long windows of a pass
};

// ordering point between two levels of the search: one wave owns the whole state space (W == 1: see
// wave_sync) or a team of W waves shares it (a workgroup barrier)
template <int W>
__device__ __forceinline__ void level_sync() {
  if (W == 1) wave_sync();
  else __syncthreads();
}

// Levels 3 - 5 of a Viterbi step (see viterbi_forward), for one wave (W == 1) or a team of W waves per
// utterance; `queue`: the calling wave's own long-window queue
template <int W>
__device__ __forceinline__ void refine_levels(const VitShared& sh, int4* __restrict__ queue, const int S,
                                              const float factor, const int lane, const int wid) {
  // Level 3: the states 8, 16, 24, 40, ... (multiples of 8 that are not multiples of 32) between the
  // backpointers of their two level-2 neighbours; level 4: every other state between the backpointers
  // of its two neighbours at the multiples of 8.  One lane per state, 64 states per pass, a serial
  // exact scan of the window (4 candidates in flight); the states whose window is long - a step of the
  // backpointer function from one attracting state to the next - go to teams of 8 lanes (below).
  // (Round 2 refined through the strides 16, 8, 4, 2, 1: fewer candidates - 2 350 against 4 000 per
  // frame - but five levels of setup; measured with the same long-window teams: 9.6 against 8.5 ms.)
  for (int level = 3; level <= (kSplitLevel5 ? 6 : 5); ++level) {
    level_sync<W>();
    // level 3: multiples of 8 that are not multiples of 32, neighbours 32 apart; level 4: the states
    // 4, 12, 20, ..., neighbours 8 apart; level 5: every other state, neighbours 4 apart (split form:
    // level 5 the states 2, 6, 10, ..., level 6 the odd states, neighbours 2 apart)
    const int gap = level == 3 ? 32 : (level == 4 ? 8 : (level == 5 ? 4 : 2));
    const int count = level == 3 ? ((S + 7) >> 3) - ((S + 31) >> 5)
                      : level == 4 ? (S + 3) >> 3
                      : !kSplitLevel5 ? S - ((S + 3) >> 2)
                      : level == 5 ? (S + 1) >> 2 : S >> 1;
    const int long_range = level == 3 ? kLongRange3 : kLongRange4;
    // a team of W waves deals the states of a level round robin: wave `wid` owns k = kk W + wid
    const int count_w = W == 1 ? count : (count - wid + W - 1) / W;
    int n_queued = 0;
    for (int k0 = 0; k0 < count_w; k0 += 64) {
      const int kk = k0 + lane < count_w ? k0 + lane : 0;
      const int k = W == 1 ? kk : kk * W + wid;
      const int i = level == 3 ? (k + k / 3 + 1) << 3
                    : level == 4 ? 4 + 8 * k
                    : !kSplitLevel5 ? k + k / 3 + 1
                    : level == 5 ? 2 + 4 * k : 1 + 2 * k;
      const bool active = k0 + lane < count_w;
      const int below = i & ~(gap - 1), above = below + gap;
      const int lo = sh.bpw[below];
      const int hi = above < S ? sh.bpw[above] : S - 1;
      const bool is_long = active && hi - lo >= long_range;
      float best = FLT_MAX;
      int best_j = lo;
      if (active && !is_long) scan_range(sh.fwd, lo, hi, static_cast<float>(i), factor, best, best_j);
      if (active && !is_long) {
        sh.bpw[i] = best_j;
        sh.nxt[i] = best + sh.nxt[i];
      }
      // Long windows (a step of the backpointer function between two attracting states: ~5 states of
      // level 3 and ~15-45 of levels 4 and 5 per frame) are queued in LDS; the queue is worked off once
      // per level (nothing in a level reads another state of the level), so that the rounds below run
      // full: teams of 8 lanes, 8 windows per round, 32 candidates per step, a 3-step DPP argmin, the
      // team's first lane stores.  (Round 2 handed them to 16-lane rows four at a time through
      // v_readlane broadcasts: ~250 instructions per round; that was most of the tracker's time.)
      const unsigned long long long_mask = __ballot(is_long);
      if (long_mask != 0) {
        if (is_long)
          queue[n_queued + __popcll(long_mask & ((1ull << lane) - 1ull))] = make_int4(i, lo, hi, 0);
        n_queued += __popcll(long_mask);
      }
      if (n_queued > kQueueEntries - 64 || (k0 + 64 >= count_w && n_queued > 0)) {
        wave_sync();
        const int team = lane >> 3, tl = lane & 7;
        for (int q0 = 0; q0 < n_queued; q0 += 8) {
          const bool on = q0 + team < n_queued;
          const int4 e = queue[on ? q0 + team : 0];
          const float fi = static_cast<float>(e.x);
          float cb = FLT_MAX, cd = 1.0e9f;   // best cost, its d = j - i (no candidate: beyond every state)
          int cj = 0x7fffffff;
          if (on) {
            // four candidates of the lane in flight per step; a step may look up to 31 states beyond the
            // window: the argmin over ALL states lies inside it (monotonicity), so the extra candidates
            // cannot win, and the forward costs are padded with FLT_MAX behind the last state
            float d = static_cast<float>(e.y + tl) - fi;
            for (int j = e.y + tl; j <= e.z; j += 32) {
              float ff[4];
#pragma unroll
              for (int w = 0; w < 4; ++w) ff[w] = sh.fwd[j + 8 * w];
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                const float dw = d + static_cast<float>(8 * w);
                const float c = __fadd_rn(__fmul_rn(dw * dw, factor), ff[w]);
                cd = c < cb ? dw : cd;
                cb = fminf(cb, c);
              }
              d += 32.0f;
            }
            cj = static_cast<int>(fi + cd);
          }
          quad_argmin(cb, cj);
          argmin_take(cb, cj, dpp_f<0x141>(cb), dpp_i<0x141>(cj));  // row_half_mirror: the other quad
          if (on && tl == 0) {
            sh.bpw[e.x] = cj;
            sh.nxt[e.x] = cb + sh.nxt[e.x];
          }
        }
        n_queued = 0;
        wave_sync();
      }
    }
  }
}

// one forward pass over all frames; returns with sh.fwd = final normalised forward cost
// NK: 64-state slices of the state space held in registers (7: up to 448 states - the default 417 -,
// 8: up to 512); 0: any size, rows read from HBM where they are used
template <int NK>
__device__ void viterbi_forward(const PitchDevTables& t, const float* __restrict__ res, const float* __restrict__ anp,
                                int64_t T, int64_t T1, bool rescale, float old_b1, float old_b2,
                                float new_ballast, int16_t* __restrict__ bp, const VitShared& sh,
                                const float* __restrict__ st_lag, const int lane) {
  const int S = t.num_states;
  for (int s = lane; s < S; s += 64) sh.fwd[s] = 0.0f;
  for (int s = S + lane; s < S + kFwdPad; s += 64) sh.fwd[s] = FLT_MAX;  // scan read-ahead padding
  const float factor = t.inter_frame_factor;
  // up to 512 states: the lane's states lane + 64 k live in registers with compile-time indices (a
  // run-time k puts the array in scratch memory, and the flat load that fetches it back waits for the
  // row prefetch too: the HBM latency the prefetch is there to hide).  Slots beyond the last state work
  // on the last state again (same inputs, same stores): the per-frame code has no branch, so the compiler
  // can count the memory operations in flight and never waits for more than it needs.
  constexpr bool in_regs = NK > 0;
  constexpr int kRowRegs = NK > 0 ? NK : 1;
  float ahead[kRowRegs], soft_lag[kRowRegs];
  int col[kRowRegs];
#pragma unroll
  for (int k = 0; k < kRowRegs; ++k) {
    col[k] = lane + 64 * k < S ? lane + 64 * k : S - 1;
    ahead[k] = (in_regs && T > 0) ? res[col[k]] : 0.0f;
    soft_lag[k] = t.soft_min_f0 * st_lag[col[k]];
  }
  // Loads and stores share one in-order counter (vmcnt), so the wait for the prefetched row would also
  // wait for every store issued after it.  The backpointers of frame t therefore stay in registers and
  // are stored at the top of frame t + 1, BEFORE the row of frame t + 2 is requested: the row loads are
  // always the youngest memory operations when their wait comes, and everything older is a frame old.
  int bpv[kRowRegs];
#pragma unroll
  for (int k = 0; k < kRowRegs; ++k) bpv[k] = 0;
  const int n_super = (S + 127) >> 7;          // states 0, 128, 256, ...
  const int n_reps = (S + 31) >> 5;            // states 0, 32, 64, ...
  for (int64_t frame = 0; frame < T; ++frame) {
    // ---- local cost of every state: 1 - nccf + soft_min_f0 * lag * nccf -> nxt ---------------------
    float scale = 1.0f;
    if (rescale) {
      const float old_ballast = frame < T1 ? old_b1 : old_b2, a = anp[frame];
      scale = sqrtf((old_ballast + a) / (new_ballast + a));
    }
    const float* __restrict__ row = res + frame * static_cast<int64_t>(S);
    wave_sync();
    // (rows are read one frame ahead: the HBM latency of frame t + 1 hides behind the search of frame t)
    if (in_regs) {
#pragma unroll
      for (int k = 0; k < kRowRegs; ++k) {
        float v = ahead[k];
        if (rescale) v *= scale;
        float local = 1.0f - v;
        local += soft_lag[k] * v;
        sh.nxt[col[k]] = local;
      }
      // the row of frame t + 1, requested now and first used a whole search later.  Unconditional loads
      // into the registers that just died (clamped column, the last frame reads its own row again): a
      // load under a branch comes back through a copy, and the copy waits for it right here
      if (frame > 0) {
        int16_t* __restrict__ bp_row = bp + (frame - 1) * S;
#pragma unroll
        for (int k = 0; k < kRowRegs; ++k) bp_row[col[k]] = static_cast<int16_t>(bpv[k]);
      }
      const float* __restrict__ nrow = frame + 1 < T ? row + S : row;
#pragma unroll
      for (int k = 0; k < kRowRegs; ++k) ahead[k] = nrow[col[k]];
    } else {
      for (int s = lane; s < S; s += 64) {
        float v = row[s];
        if (rescale) v *= scale;
        float local = 1.0f - v;
        local += t.soft_min_f0 * st_lag[s] * v;
        sh.nxt[s] = local;
      }
    }
    // ---- Viterbi step.  cost(i, j) = (j - i)^2 * factor + fwd[j]; its argmin is monotone in i
    // (Kaldi's own search relies on it).  Level 1: exact argmin of the states 0, 128, 256, ... (a
    // 16-lane row each, strided scan, lowest index wins ties).  Level 2: the states 32, 64, 96, 160, ...
    // (4 lanes each) between the backpointers of their two level-1 neighbours.  Then the strides
    // 16, 8, 4, 2, 1: every new state scans only between the backpointers of its two known
    // neighbours. -------------------------------------------------------------------------------
    wave_sync();
    for (int base = 0; base < n_super; base += 4) {
      const int r = base + (lane >> 4), sub = lane & 15;
      const int i_rep = r << 7;
      float best = FLT_MAX;
      int best_j = 0x7fffffff;
      if (r < n_super) {
        const float fi = static_cast<float>(i_rep);
        float d = static_cast<float>(sub) - fi, bd = 1.0e9f;
#pragma unroll 9
        for (int j = sub; j < S; j += 16) {
          const float c = __fadd_rn(__fmul_rn(d * d, factor), sh.fwd[j]);
          bd = c < best ? d : bd;
          best = fminf(best, c);
          d += 16.0f;
        }
        best_j = static_cast<int>(fi + bd);
      }
      quad_argmin(best, best_j);
      argmin_take(best, best_j, dpp_f<0x124>(best), dpp_i<0x124>(best_j));
      argmin_take(best, best_j, dpp_f<0x128>(best), dpp_i<0x128>(best_j));
      if (sub == 0 && r < n_super) {
        if (best_j >= S) best_j = 0;  // (only if every cost was NaN / inf: no runaway scans below)
        sh.bpw[i_rep] = best_j;
        sh.nxt[i_rep] = best + sh.nxt[i_rep];
      }
    }
    wave_sync();
    for (int base = 0; base < n_reps; base += 16) {
      // k-th state of this level -> multiple of 32 that is not a multiple of 128
      const int k = base + (lane >> 2), sub = lane & 3;
      const int m = k + k / 3 + 1;
      const int i_rep = m << 5;
      float best = FLT_MAX;
      int best_j = 0x7fffffff;
      if (i_rep < S) {
        const int below = i_rep & ~127, above = below + 128;
        const int lo = sh.bpw[below];
        const int hi = above < S ? sh.bpw[above] : S - 1;
        const float fi = static_cast<float>(i_rep);
        // (a step may look beyond `hi`: the argmin over ALL states lies inside the range, so the
        // extra candidates cannot win; the forward costs are padded with FLT_MAX behind the last state)
        float d = static_cast<float>(lo + sub) - fi, bd = d;
        for (int j = lo + sub; j <= hi; j += 32) {
          float ff[8];
#pragma unroll
          for (int w = 0; w < 8; ++w) ff[w] = sh.fwd[j + 4 * w];
#pragma unroll
          for (int w = 0; w < 8; ++w) {
            const float dw = d + static_cast<float>(4 * w);
            const float c = __fadd_rn(__fmul_rn(dw * dw, factor), ff[w]);
            bd = c < best ? dw : bd;
            best = fminf(best, c);
          }
          d += 32.0f;
        }
        best_j = static_cast<int>(fi + bd);
      }
      quad_argmin(best, best_j);
      if (sub == 0 && i_rep < S) {
        if (best_j >= S) best_j = sh.bpw[i_rep & ~127];
        sh.bpw[i_rep] = best_j;
        sh.nxt[i_rep] = best + sh.nxt[i_rep];
      }
    }
    refine_levels<1>(sh, sh.queue, S, factor, lane, 0);
    wave_sync();
    float lane_min = FLT_MAX;
    if (in_regs) {
      float nx[kRowRegs];
#pragma unroll
      for (int k = 0; k < kRowRegs; ++k) {
        nx[k] = sh.nxt[col[k]];
        bpv[k] = sh.bpw[col[k]];
        lane_min = fminf(lane_min, nx[k]);
      }
      const float mn = wave_min_f(lane_min);
#pragma unroll
      for (int k = 0; k < kRowRegs; ++k) sh.fwd[col[k]] = nx[k] + (-mn);
    } else {
      for (int s = lane; s < S; s += 64) {
        lane_min = fminf(lane_min, sh.nxt[s]);
        bp[frame * S + s] = static_cast<int16_t>(sh.bpw[s]);
      }
      const float mn = wave_min_f(lane_min);
      for (int s = lane; s < S; s += 64) sh.fwd[s] = sh.nxt[s] + (-mn);
    }
  }
  if (in_regs && T > 0) {
    int16_t* __restrict__ bp_row = bp + (T - 1) * S;
#pragma unroll
    for (int k = 0; k < kRowRegs; ++k) bp_row[col[k]] = static_cast<int16_t>(bpv[k]);
  }
  wave_sync();
}


// traceback of one utterance by ONE wave: best final state (lowest index wins ties), then the chain of
// backpointers -> states[f0 ...]; `sh.fwd` holds the final forward costs, `sh.nxt` is scratch
__device__ __forceinline__ void viterbi_traceback(const int S, const int64_t T, const int64_t f0,
                                                  const int16_t* __restrict__ bp, int32_t* __restrict__ states,
                                                  const VitShared& sh, const int lane) {
  __threadfence_block();
  {
    float bv = FLT_MAX;
    int best = 0x7fffffff;
    for (int s = lane; s < S; s += 64) {
      const float c = sh.fwd[s];
      if (c < bv) { bv = c; best = s; }
    }
    wave_argmin(bv, best);
    if (!kTracePrefetch) {
      if (lane == 0) {
        for (int64_t frame = T - 1; frame >= 0; --frame) {
          states[f0 + frame] = best;
          best = bp[frame * S + best];
        }
      }
    } else {
      // 64 frames at a time.  The chain of dependent 2-byte loads is the whole cost (a trip to HBM per
      // frame), so the wave first touches the rows of the chunk at the columns around the current state -
      // the path moves a few states per frame, the touched lines hold 64 - and lane 0 then walks through
      // lines that are already in the cache; the path goes to LDS and is written out 64 frames at once
      // (a store inside the chain would be waited for together with every load).
      int* trace = reinterpret_cast<int*>(sh.nxt);
      for (int64_t hi = T; hi > 0; hi -= 64) {
        const int n = hi < 64 ? static_cast<int>(hi) : 64;
        best = __builtin_amdgcn_readfirstlane(best);
        {
          int col_t = best + ((lane & 32) ? 24 : -24);
          col_t = col_t < 0 ? 0 : (col_t > S - 1 ? S - 1 : col_t);
          const int64_t fa = hi - 1 - (lane & 31), fb = fa - 32;
          const int16_t va = bp[(fa > 0 ? fa : 0) * S + col_t], vb = bp[(fb > 0 ? fb : 0) * S + col_t];
          asm volatile("" : : "v"(va), "v"(vb));
        }
        if (lane == 0) {
          for (int k = 0; k < n; ++k) {
            trace[k] = best;
            best = bp[(hi - 1 - k) * S + best];
          }
        }
        wave_sync();
        if (lane < n) states[f0 + hi - 1 - lane] = trace[lane];
        wave_sync();
      }
    }
  }
}

// output rows of one utterance: (POV NCCF resampled at the chosen lag, 1 / lag); frames first, first + stride, ...
__device__ __forceinline__ void pitch_output_rows(const PitchDevTables& t, const int L, const int64_t T,
                                                  const int64_t f0, const int32_t* __restrict__ states,
                                                  const float* __restrict__ pov_nccf, float* __restrict__ out,
                                                  const int first, const int stride) {
  for (int64_t frame = first; frame < T; frame += stride) {
    const int s = states[f0 + frame];
    const float* __restrict__ wt = t.ar_w + s * t.ar_max_taps;
    const float* __restrict__ src = pov_nccf + frame * L + t.ar_first[s];
    const int n = t.ar_n[s];
    float pov = 0.0f;
    for (int j = 0; j < n; ++j) pov = __builtin_fmaf(src[j], wt[j], pov);
    out[(f0 + frame) * 2 + 0] = pov;
    out[(f0 + frame) * 2 + 1] = 1.0f / t.lags[s];
  }
}

// ---- 4b. the same forward pass by a TEAM of W wavefronts per utterance (small batches) ----------------
// One wave per utterance walks its frames at ~11 us per frame whatever its neighbours do: below ~2 000
// utterances most SIMDs hold one wave and the tracker's time is that wave's latency (the LDS round trips of
// five levels in series), not the machine's throughput.  Here the W waves of a workgroup share ONE utterance:
// the 64-state slices of the row (local cost, normalisation, backpointer rows) are dealt round robin, level 1
// gives every exact scan 16 W lanes, level 2 4 W lanes per state, the states of levels 3 - 5 are dealt round
// robin (refine_levels<W>: every wave has its own long-window queue), and workgroup barriers stand where the
// single wave has its ordering points.  Same candidates, same arithmetic, same tie-breaks per state: the
// result is bit-identical to viterbi_forward (the pitch tests run both).  Up to 512 states.
template <int W>
__device__ void viterbi_forward_team(const PitchDevTables& t, const float* __restrict__ res,
                                     const float* __restrict__ anp, int64_t T, int64_t T1, bool rescale,
                                     float old_b1, float old_b2, float new_ballast, int16_t* __restrict__ bp,
                                     const VitShared& sh, int4* __restrict__ queue, float* __restrict__ red,
                                     const float* __restrict__ st_lag, const int lane, const int wid) {
  constexpr int NKW = 8 / W;                   // 64-state slices per wave
  constexpr int LPS = 16 * W;                  // lanes per exact scan of level 1
  constexpr int LP2 = 4 * W;                   // lanes per state of level 2
  constexpr int IF2 = 32 / LP2;                // its candidates in flight per lane: 32 per step and state
  const int S = t.num_states, tid = wid * 64 + lane;
  for (int s = tid; s < S; s += 64 * W) sh.fwd[s] = 0.0f;
  for (int s = S + tid; s < S + kFwdPad; s += 64 * W) sh.fwd[s] = FLT_MAX;  // scan read-ahead padding
  const float factor = t.inter_frame_factor;
  float ahead[NKW], soft_lag[NKW];
  int col[NKW], bpv[NKW];
#pragma unroll
  for (int k = 0; k < NKW; ++k) {
    const int c = lane + 64 * (k * W + wid);   // (slots beyond the last state redo the last state)
    col[k] = c < S ? c : S - 1;
    ahead[k] = T > 0 ? res[col[k]] : 0.0f;
    soft_lag[k] = t.soft_min_f0 * st_lag[col[k]];
    bpv[k] = 0;
  }
  const int n_super = (S + 127) >> 7;          // states 0, 128, 256, ... (at most 4)
  for (int64_t frame = 0; frame < T; ++frame) {
    float scale = 1.0f;
    if (rescale) {
      const float old_ballast = frame < T1 ? old_b1 : old_b2, a = anp[frame];
      scale = sqrtf((old_ballast + a) / (new_ballast + a));
    }
    const float* __restrict__ row = res + frame * static_cast<int64_t>(S);
#pragma unroll
    for (int k = 0; k < NKW; ++k) {
      float v = ahead[k];
      if (rescale) v *= scale;
      float local = 1.0f - v;
      local += soft_lag[k] * v;
      sh.nxt[col[k]] = local;
    }
    if (frame > 0) {
      int16_t* __restrict__ bp_row = bp + (frame - 1) * S;
#pragma unroll
      for (int k = 0; k < NKW; ++k) bp_row[col[k]] = static_cast<int16_t>(bpv[k]);
    }
    const float* __restrict__ nrow = frame + 1 < T ? row + S : row;
#pragma unroll
    for (int k = 0; k < NKW; ++k) ahead[k] = nrow[col[k]];
    __syncthreads();   // local costs and the forward costs of the last frame are in LDS
    // ---- level 1: exact argmin of the states 0, 128, 256, 384 - LPS lanes each
    {
      const int r = wid * (64 / LPS) + lane / LPS, sub = lane % LPS;
      const int i_rep = r << 7;
      float best = FLT_MAX;
      int best_j = 0x7fffffff;
      if (r < n_super) {
        const float fi = static_cast<float>(i_rep);
        float d = static_cast<float>(sub) - fi, bd = 1.0e9f;
#pragma unroll 4
        for (int j = sub; j < S; j += LPS) {
          const float c = __fadd_rn(__fmul_rn(d * d, factor), sh.fwd[j]);
          bd = c < best ? d : bd;
          best = fminf(best, c);
          d += static_cast<float>(LPS);
        }
        best_j = static_cast<int>(fi + bd);
      }
      quad_argmin(best, best_j);
      argmin_take(best, best_j, dpp_f<0x124>(best), dpp_i<0x124>(best_j));
      argmin_take(best, best_j, dpp_f<0x128>(best), dpp_i<0x128>(best_j));
      if (LPS > 16) {
        // the 16-lane rows of a scan: row values through v_readlane, combined in index order
        float c0 = lane_f(best, 0), c1 = lane_f(best, 16), c2 = lane_f(best, 32), c3 = lane_f(best, 48);
        int j0 = __builtin_amdgcn_readlane(best_j, 0), j1 = __builtin_amdgcn_readlane(best_j, 16);
        int j2 = __builtin_amdgcn_readlane(best_j, 32), j3 = __builtin_amdgcn_readlane(best_j, 48);
        argmin_take(c0, j0, c1, j1);
        argmin_take(c2, j2, c3, j3);
        if (LPS == 64) {
          argmin_take(c0, j0, c2, j2);
          best = c0;
          best_j = j0;
        } else {
          best = lane < 32 ? c0 : c2;
          best_j = lane < 32 ? j0 : j2;
        }
      }
      if (sub == 0 && r < n_super) {
        if (best_j >= S) best_j = 0;  // (only if every cost was NaN / inf: no runaway scans below)
        sh.bpw[i_rep] = best_j;
        sh.nxt[i_rep] = best + sh.nxt[i_rep];
      }
    }
    __syncthreads();
    // ---- level 2: the other multiples of 32 between the backpointers of their level-1 neighbours, LP2
    // lanes per state (at most 12 states: 64 / LP2 per wave), 32 candidates per step
    {
      const int k = wid * (64 / LP2) + lane / LP2, sub = lane % LP2;
      const int m = k + k / 3 + 1;
      const int i_rep = m << 5;
      float best = FLT_MAX;
      int best_j = 0x7fffffff;
      if (i_rep < S) {
        const int below = i_rep & ~127, above = below + 128;
        const int lo = sh.bpw[below];
        const int hi = above < S ? sh.bpw[above] : S - 1;
        const float fi = static_cast<float>(i_rep);
        float d = static_cast<float>(lo + sub) - fi, bd = d;
        for (int j = lo + sub; j <= hi; j += 32) {
          float ff[IF2];
#pragma unroll
          for (int w = 0; w < IF2; ++w) ff[w] = sh.fwd[j + LP2 * w];
#pragma unroll
          for (int w = 0; w < IF2; ++w) {
            const float dw = d + static_cast<float>(LP2 * w);
            const float c = __fadd_rn(__fmul_rn(dw * dw, factor), ff[w]);
            bd = c < best ? dw : bd;
            best = fminf(best, c);
          }
          d += 32.0f;
        }
        best_j = static_cast<int>(fi + bd);
      }
      quad_argmin(best, best_j);
      if (LP2 == 8) {
        argmin_take(best, best_j, dpp_f<0x141>(best), dpp_i<0x141>(best_j));  // row_half_mirror
      } else {
        argmin_take(best, best_j, dpp_f<0x124>(best), dpp_i<0x124>(best_j));
        argmin_take(best, best_j, dpp_f<0x128>(best), dpp_i<0x128>(best_j));
      }
      if (sub == 0 && i_rep < S) {
        if (best_j >= S) best_j = sh.bpw[i_rep & ~127];
        sh.bpw[i_rep] = best_j;
        sh.nxt[i_rep] = best + sh.nxt[i_rep];
      }
    }
    refine_levels<W>(sh, queue, S, factor, lane, wid);   // (starts with the barrier behind level 2)
    __syncthreads();
    // ---- normalisation: the minimum over all states, through one LDS slot per wave
    float nx[NKW], lane_min = FLT_MAX;
#pragma unroll
    for (int k = 0; k < NKW; ++k) {
      nx[k] = sh.nxt[col[k]];
      bpv[k] = sh.bpw[col[k]];
      lane_min = fminf(lane_min, nx[k]);
    }
    const float wave_mn = wave_min_f(lane_min);
    if (lane == 0) red[wid] = wave_mn;
    __syncthreads();
    float mn = red[0];
#pragma unroll
    for (int w = 1; w < W; ++w) mn = fminf(mn, red[w]);
#pragma unroll
    for (int k = 0; k < NKW; ++k) sh.fwd[col[k]] = nx[k] + (-mn);
  }
  if (T > 0) {
    int16_t* __restrict__ bp_row = bp + (T - 1) * S;
#pragma unroll
    for (int k = 0; k < NKW; ++k) bp_row[col[k]] = static_cast<int16_t>(bpv[k]);
  }
  __syncthreads();
}


// ---- 4c. the same forward pass with a lane per CANDIDATE (round 5) ------------------------------------------
// The search above gives a lane to every STATE and lets it scan its window, which is as long as the backpointer
// function is steep there: windows of 1 to 100+ candidates side by side in one wave, hence the scan loops, the
// queue of long windows, the 8-lane teams and five levels of set-up - 3 070 instructions per frame and wave
// of which the candidate arithmetic is a sixth (profiles/r05_pmc_pitch10k_before_summary.txt,
// tools/pitch_search_replay.py).  Here every lane owns seven consecutive CANDIDATES j = 7 lane + k.  Per level
// (the multiples of 128 exactly, then of 32, 8, 4, then every state - the same five levels):
//   * the backpointers of the known states are counted into marks[j]; a prefix sum over j (six adds per lane, one
//     scan over the lanes) tells every candidate between which two known states' backpointers it lies: its gap;
//   * a candidate strictly inside a gap offers its cost to the (one or three) new states of that gap with ONE LDS
//     instruction each: an atomic minimum on the 64-bit key (cost bits << 32 | j) - costs are non-negative
//     floats, so the unsigned order of the keys is "lower cost first, lower index on ties", Kaldi's rule;
//   * the two ends of a window (candidates that ARE a known backpointer belong to two windows) are offered by the
//     new state itself, which thereby also initialises its key.
// No loop depends on the data, there is no long-window case, and a jump of the backpointer function costs what
// its candidates cost, once per level.  Same candidates per state, same arithmetic (d = j - i exact in float,
// fl(fl(d d) f) + fwd[j] without contraction), same tie-break: bit-identical to viterbi_forward.
// MEASURED (10 000 x 3 s, same box, whole pitch call; lane-per-state kernel 14.1-14.3 ms, 3 070 instructions per
// frame and wave).  An LDS atomic costs ~0.6 clocks per active LANE: one per candidate: 35 ms; one per run of a
// lane's candidates that share a gap: 16.0; level 1 through DPP instead of four 64-lane atomics on one key: 15.2;
// the two coarse levels (13 states, windows hundreds of candidates wide, where every lane's last run hits one of
// 3 keys) in the lane-per-state form of viterbi_forward: 13.8; keys in four planes by state mod 4 (bank conflicts
// 490 -> 320 clocks per frame): 13.6; window ends by a lane per gap instead of per new state: 13.1; marks cumulative
// over the levels of a frame: 13.0; a run's restart as arithmetic: 12.6; (cost, index) keys reduced over the lanes with
// one 64-bit compare per step, level 1 as a scalar loop, level 2 on 60 lanes: **11.8-11.9 ms**, 2 050 instructions per
// frame and wave (tools/experiments/README.md has the table).  The default since then (SNF_PITCH_FLAT=0: the
// lane-per-state kernel).
constexpr int kFlatCand = 7;               // candidates per lane: 7 x 64 = 448 states at most
constexpr int kFlatSlots = 448 + 32;       // keys: a last partial gap of the first candidate level names states up to S + 15
constexpr int kFlatWaves = 8;              // utterances (= wavefronts) per workgroup: two workgroups per CU = 4 waves per SIMD
// (10 waves per workgroup and 96 registers = 5 waves per SIMD: 36 spilled registers, 14 scratch accesses per frame,
// 15.9 against 12.6 ms per 10 000 utterances)
constexpr int kFlatWaveBytes = kFlatSlots * 8 + (448 + kFwdPad) * 4 + 448 * 4;   // keys, forward costs, marks of a wave

struct FlatShared {
  float* fwd;                   // [448 + kFwdPad]  normalised forward costs of the previous frame (FLT_MAX behind S)
  unsigned long long* slots;    // [kFlatSlots]     key of every state: cost bits << 32 | backpointer
  unsigned* marks;              // [448]            number of known states whose backpointer is j
};

// exclusive prefix sum over the 64 lanes (DPP inside the 16-lane rows, the three row totals through v_readlane)
__device__ __forceinline__ int wave_exclusive_sum(int v, int lane) {
  int t = v;
  t += __builtin_amdgcn_update_dpp(0, t, 0x111, 0xf, 0xf, true);   // row_shr:1
  t += __builtin_amdgcn_update_dpp(0, t, 0x112, 0xf, 0xf, true);   // row_shr:2
  t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xf, true);   // row_shr:4
  t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xf, true);   // row_shr:8
  const int r0 = __builtin_amdgcn_readlane(t, 15), r1 = __builtin_amdgcn_readlane(t, 31),
            r2 = __builtin_amdgcn_readlane(t, 47);
  int base = lane >= 16 ? r0 : 0;
  base += lane >= 32 ? r1 : 0;
  base += lane >= 48 ? r2 : 0;
  return t + base - v;
}

// Where the key of state u lives: four planes by u mod 4, each indexed by u / 4.  At the last level the known
// states (multiples of 4) and each of the three kinds of new states are then contiguous, at the level before the
// known states are two entries apart: with the keys in state order every access of a level was 32 / 64 / 256
// bytes apart - 4, 2 or 1 banks for the whole wave (SQ_LDS_BANK_CONFLICT 490 clocks per frame and wave).
constexpr int kFlatPlane = kFlatSlots / 4;
__device__ __forceinline__ int flat_slot(int u) { return (u & 3) * kFlatPlane + (u >> 2); }

// minimum of two floats that are never NaN: fminf() quiets its operands first (a v_max_f32 x, x per call)
__device__ __forceinline__ float min_nonan(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ unsigned long long flat_key(float cost, int j) {
  return (static_cast<unsigned long long>(__builtin_bit_cast(unsigned, cost)) << 32) | static_cast<unsigned>(j);
}

// one step of an argmin over lanes ON THE KEYS: the unsigned order of (cost bits, index) is "lower cost, then lower
// index" for non-negative costs - two DPP moves, one 64-bit compare and two selects, where the same step on a
// (float, int) pair compiles to four compares and two exec-masked blocks
template <int CTRL>
__device__ __forceinline__ unsigned long long key_min_step(unsigned long long key) {
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<unsigned>(key)), CTRL, 0xf, 0xf, true));
  const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(static_cast<unsigned>(key >> 32)), CTRL, 0xf, 0xf, true));
  const unsigned long long other = (static_cast<unsigned long long>(hi) << 32) | lo;
  return other < key ? other : key;
}
// the same with the key of the lane `CTRL` names taken from `from` (row_shl:n: lane + n of the row; a lane whose
// source lies outside its row keeps its own halves of `from`)
template <int CTRL>
__device__ __forceinline__ unsigned long long key_min_shl(unsigned long long key, unsigned long long from) {
  const int flo = static_cast<int>(static_cast<unsigned>(from)), fhi = static_cast<int>(static_cast<unsigned>(from >> 32));
  const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_update_dpp(flo, flo, CTRL, 0xf, 0xf, false));
  const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_update_dpp(fhi, fhi, CTRL, 0xf, 0xf, false));
  const unsigned long long other = (static_cast<unsigned long long>(hi) << 32) | lo;
  return other < key ? other : key;
}

// one level: the states u = g KNOWN + (s + 1) NEW (s < KNOWN / NEW - 1) of every gap g between the known states
// g KNOWN and (g + 1) KNOWN
// FRESH: the stride of the states that became known since the marks were last brought up to date (the marks are
// cumulative over the levels of a frame: only the backpointers of those states are added)
template <int KNOWN, int NEW, int FRESH_FROM>
__device__ __forceinline__ void flat_level(const FlatShared& sh, const int S, const float factor, const int lane,
                                           const float (&fj)[kFlatCand], const float jf0) {
  constexpr int M = KNOWN / NEW - 1;
  static_assert(M == 1 || M == 3, "one or three new states per gap");
  const int n_gaps = S > NEW ? (S - NEW - 1) / KNOWN + 1 : 0;   // gaps that hold a new state below S
  const unsigned* __restrict__ bps = reinterpret_cast<const unsigned*>(sh.slots);   // plane 0, low words: backpointers
  unsigned* __restrict__ mk_lane = sh.marks + kFlatCand * lane;
  // The phases of a level are LDS round trips of one wave; what does not depend on each other is issued together:
  // (1) the backpointers the marks still lack and the two ends of every window, (2) the marks' atomic adds and the
  // forward costs at the window ends, (3) the lane's seven marks and the keys of the window ends.
  // ---- marks: the states t FRESH (FRESH_FROM == 0: all known states; else those that are not multiples of
  // FRESH_FROM) became known at the previous level
  constexpr int FRESH = KNOWN;
  const int n_fresh = (S + FRESH - 1) / FRESH;
  for (int t0 = 0; t0 < n_fresh; t0 += 64) {
    const int t = t0 + lane;
    const bool fresh = t < n_fresh && (FRESH_FROM == 0 || (t * FRESH) % FRESH_FROM != 0);
    if (fresh) atomicAdd(&sh.marks[bps[2 * (t * (FRESH / 4))]], 1u);
  }
  // ---- the ends of every window: a lane per GAP offers lo and hi to each of the gap's new states (the three
  // states of a gap share the window: one pair of reads, keys written one entry or one plane apart) -----------
  for (int g0 = 0; g0 < n_gaps; g0 += 64) {
    const int g = g0 + lane;
    if (g < n_gaps) {
      const int above = (g + 1) * KNOWN;
      const int lo = static_cast<int>(bps[2 * (g * (KNOWN / 4))]);
      const int hi = above < S ? static_cast<int>(bps[2 * (above / 4)]) : S - 1;
      const float f_lo = sh.fwd[lo], f_hi = sh.fwd[hi];
      const float e_lo = static_cast<float>(lo - g * KNOWN), e_hi = static_cast<float>(hi - g * KNOWN);
      unsigned long long* __restrict__ first = sh.slots + g * (KNOWN / 4);
#pragma unroll
      for (int sidx = 0; sidx < M; ++sidx) {
        // (a new state behind the last state of a partial gap gets a key nobody reads: the array has the room)
        const float d_lo = e_lo - static_cast<float>((sidx + 1) * NEW), d_hi = e_hi - static_cast<float>((sidx + 1) * NEW);
        const float c_lo = __fadd_rn(__fmul_rn(d_lo * d_lo, factor), f_lo);
        const float c_hi = __fadd_rn(__fmul_rn(d_hi * d_hi, factor), f_hi);
        first[NEW >= 4 ? (sidx + 1) * (NEW / 4) : (sidx + 1) * kFlatPlane] =
            c_hi < c_lo ? flat_key(c_hi, hi) : flat_key(c_lo, lo);   // (lo <= hi: the lower index on ties)
      }
    }
  }
  wave_sync();
  int mk[kFlatCand], gap[kFlatCand];
  int total = 0;
#pragma unroll
  for (int k = 0; k < kFlatCand; ++k) {
    mk[k] = static_cast<int>(mk_lane[k]);
    total += mk[k];
  }
  int count = wave_exclusive_sum(total, lane);   // known states with a backpointer below this lane's candidates
#pragma unroll
  for (int k = 0; k < kFlatCand; ++k) {
    count += mk[k];
    gap[k] = count - 1;                           // candidate j lies right of (or on) the backpointer of state gap
  }
  // ---- the candidates strictly inside a window ---------------------------------------------------------
  // An LDS atomic costs ~0.6 clocks per LANE whatever the addresses (measured: one atomic per candidate and
  // state, 4 400 lane operations per frame, was 2 500 LDS clocks per frame and wave - three times the whole
  // lane-per-state search).  The seven candidates of a lane are consecutive, so they fall into runs that share a
  // gap: the lane keeps the best of the run in registers (ascending j, strict <: the lower index on ties) and
  // offers it once, when the gap changes or its candidates end - 1 300 lane operations per frame.
  // A candidate that carries a mark (it IS a known backpointer) opens a new gap and is itself outside every
  // window: the lane's running best restarts there (and at its first candidate); it is offered at the last
  // candidate before the next mark (or at the lane's last one) if the run saw any candidate.
  float bc[M];
  int bj[M];
#pragma unroll
  for (int sidx = 0; sidx < M; ++sidx) {
    bc[sidx] = FLT_MAX;
    bj[sidx] = 0;
  }
#pragma unroll
  for (int k = 0; k < kFlatCand; ++k) {
    const int j = kFlatCand * lane + k;
    const bool restart = k == 0 || mk[k] != 0;
    // (one unsigned comparison for 0 <= gap < n_gaps; candidates behind the last state need no test: their
    // forward cost is the FLT_MAX padding)
    const bool inside = mk[k] == 0 && static_cast<unsigned>(gap[k]) < static_cast<unsigned>(n_gaps);
    // (a candidate that is not inside a window offers FLT_MAX: x + FLT_MAX = FLT_MAX for these x, never < best)
    const float fjm = inside ? fj[k] : FLT_MAX;
    const int ub = gap[k] * KNOWN;
    const float e = (jf0 + static_cast<float>(k)) - static_cast<float>(ub);   // j - g KNOWN: exact
    // (a restart as arithmetic: best + inf = inf, which every cost is below - an add in place of two selects on
    // a scalar mask per state; best + 0 = best exactly, the costs being >= +0)
    const float reset = restart ? __builtin_inff() : 0.0f;
#pragma unroll
    for (int sidx = 0; sidx < M; ++sidx) {
      const float d = e - static_cast<float>((sidx + 1) * NEW);
      const float c = __fadd_rn(__fmul_rn(d * d, factor), fjm);
      const float kept = __fadd_rn(bc[sidx], reset);
      bj[sidx] = c < kept ? j : bj[sidx];
      bc[sidx] = min_nonan(kept, c);
    }
    const bool last_of_run = k == kFlatCand - 1 || mk[k + (k < kFlatCand - 1 ? 1 : 0)] != 0;
    if (last_of_run && bc[0] < FLT_MAX) {   // (the states of a gap see the same candidates: one test for all)
      // (ub is a multiple of 4: plane 0 at ub / 4; the new states are whole entries or whole planes further)
      unsigned long long* __restrict__ first = sh.slots + (ub >> 2);
#pragma unroll
      for (int sidx = 0; sidx < M; ++sidx)
        atomicMin(first + (NEW >= 4 ? (sidx + 1) * (NEW / 4) : (sidx + 1) * kFlatPlane), flat_key(bc[sidx], bj[sidx]));
    }
  }
  wave_sync();
}

// one forward pass over all frames (S <= 448); returns with sh.fwd = final normalised forward cost
__device__ void viterbi_forward_flat(const PitchDevTables& t, const float* __restrict__ res,
                                     const float* __restrict__ anp, int64_t T, int64_t T1, bool rescale,
                                     float old_b1, float old_b2, float new_ballast, int16_t* __restrict__ bp,
                                     const FlatShared& sh, const float* __restrict__ st_lag, const int lane) {
  constexpr int NK = 7;
  const int S = t.num_states;
  for (int s = lane; s < S; s += 64) sh.fwd[s] = 0.0f;
  for (int s = S + lane; s < 448 + kFwdPad; s += 64) sh.fwd[s] = FLT_MAX;
  const float factor = t.inter_frame_factor;
  float ahead[NK], soft_lag[NK];
  int col[NK], bpv[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    col[k] = lane + 64 * k < S ? lane + 64 * k : S - 1;
    ahead[k] = T > 0 ? res[col[k]] : 0.0f;
    soft_lag[k] = t.soft_min_f0 * st_lag[col[k]];
    bpv[k] = 0;
  }
  const float jf0 = static_cast<float>(kFlatCand * lane);
  const float* __restrict__ fwd_lane = sh.fwd + kFlatCand * lane;
  for (int64_t frame = 0; frame < T; ++frame) {
    float scale = 1.0f;
    if (rescale) {
      const float old_ballast = frame < T1 ? old_b1 : old_b2, a = anp[frame];
      scale = sqrtf((old_ballast + a) / (new_ballast + a));
    }
    const float* __restrict__ row = res + frame * static_cast<int64_t>(S);
    // local costs (kept in registers until the end of the frame), the backpointers of the previous frame, the
    // row of the next one: as in viterbi_forward (see there for the order of the memory operations)
    float local[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      float v = ahead[k];
      if (rescale) v *= scale;
      local[k] = 1.0f - v;
      local[k] += soft_lag[k] * v;
    }
    if (frame > 0) {
      int16_t* __restrict__ bp_row = bp + (frame - 1) * S;
#pragma unroll
      for (int k = 0; k < NK; ++k) bp_row[col[k]] = static_cast<int16_t>(bpv[k]);
    }
    const float* __restrict__ nrow = frame + 1 < T ? row + S : row;
#pragma unroll
    for (int k = 0; k < NK; ++k) ahead[k] = nrow[col[k]];
    // ---- the lane's seven candidates ----------------------------------------------------------------------
    wave_sync();
    float fj[kFlatCand];
#pragma unroll
    for (int k = 0; k < kFlatCand; ++k) fj[k] = fwd_lane[k];
    {
      // (the marks of a frame are cumulative over its levels: zeroed once, here)
      unsigned* __restrict__ mk_lane = sh.marks + kFlatCand * lane;
#pragma unroll
      for (int k = 0; k < kFlatCand; ++k) mk_lane[k] = 0u;
    }
    // ---- levels 1 and 2 as in viterbi_forward (a 16-lane row per state of level 1 over all candidates, 4 lanes per
    // state of level 2 over its window: 13 states whose windows are hundreds of candidates wide - the lane-per-
    // candidate form pays for them with one DPP reduction per state or with 64 lanes on 3 keys), keys to `slots` --
    {
      const int n_super = (S + 127) >> 7, n_reps = (S + 31) >> 5;
      for (int base = 0; base < n_super; base += 4) {
        const int r = base + (lane >> 4), sub = lane & 15;
        const int i_rep = r << 7;
        float best = FLT_MAX;
        int best_j = 0x7fffffff;
        if (r < n_super) {
          const float fi = static_cast<float>(i_rep);
          float d = static_cast<float>(sub) - fi, bd = 1.0e9f;
          // (the same number of steps in every lane - a scalar loop, no exec mask: the reads behind S - 1 land in
          // the FLT_MAX padding, whose cost FLT_MAX is below nothing)
          // (three steps per trip, their reads first: at most 47 entries behind S - 1, inside the 448 + kFwdPad
          // entries that hold FLT_MAX from S on for every S <= 448)
          const int n_steps = (S + 15) >> 4;
          const float* __restrict__ fw = sh.fwd + sub;
          float f3[3], g3[3];
#pragma unroll
          for (int w = 0; w < 3; ++w) f3[w] = fw[16 * w];
          for (int t = 0; t < n_steps; t += 3) {
            if (t + 3 < n_steps) {   // (scalar branch: the next trip's reads run under this trip's arithmetic)
#pragma unroll
              for (int w = 0; w < 3; ++w) g3[w] = fw[16 * (t + 3 + w)];
            }
#pragma unroll
            for (int w = 0; w < 3; ++w) {
              const float c = __fadd_rn(__fmul_rn(d * d, factor), f3[w]);
              const bool better = c < best;   // (one compare, two selects on vcc: no minimum, no canonicalisation)
              bd = better ? d : bd;
              best = better ? c : best;
              d += 16.0f;
            }
#pragma unroll
            for (int w = 0; w < 3; ++w) f3[w] = g3[w];
          }
          best_j = static_cast<int>(fi + bd);
        }
        unsigned long long key = flat_key(best, best_j);
        key = key_min_step<0xB1>(key);    // quad_perm [1,0,3,2]
        key = key_min_step<0x4E>(key);    // quad_perm [2,3,0,1]
        key = key_min_step<0x124>(key);   // row_ror:4
        key = key_min_step<0x128>(key);   // row_ror:8
        if (sub == 0 && r < n_super) {
          if (static_cast<unsigned>(key) >= static_cast<unsigned>(S)) key &= 0xffffffff00000000ull;   // (index 0)
          sh.slots[flat_slot(i_rep)] = key;
        }
      }
      wave_sync();
      // (five lanes per state, three states per 16-lane row: up to twelve states per trip - the ten multiples
      // of 32 below 448 that are not multiples of 128 - on 60 lanes; four lanes per state used 40 and took a
      // third more trips of the candidate loop)
      for (int base = 0; base < n_reps; base += 12) {
        const int in_row = lane & 15, group = (in_row * 13) >> 6, sub = in_row - 5 * group;   // in_row / 5, % 5
        const int k = base + 3 * (lane >> 4) + group;
        const int m = k + k / 3 + 1;
        const int i_rep = group < 3 ? m << 5 : S;   // (lane 15 of a row: idle)
        float best = FLT_MAX;
        int best_j = 0x7fffffff, lo = 0;
        if (i_rep < S) {
          const int below = i_rep & ~127, above = below + 128;
          lo = static_cast<int>(static_cast<unsigned>(sh.slots[flat_slot(below)]));
          const int hi = above < S ? static_cast<int>(static_cast<unsigned>(sh.slots[flat_slot(above)])) : S - 1;
          const float fi = static_cast<float>(i_rep);
          float d = static_cast<float>(lo + sub) - fi, bd = d;
          // (reads up to 35 entries behind hi <= S - 1: inside the kFwdPad entries of FLT_MAX)
          for (int j = lo + sub; j <= hi; j += 40) {
            float ff[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) ff[w] = sh.fwd[j + 5 * w];
#pragma unroll
            for (int w = 0; w < 8; ++w) {
              const float dw = d + static_cast<float>(5 * w);
              const float c = __fadd_rn(__fmul_rn(dw * dw, factor), ff[w]);
              bd = c < best ? dw : bd;
              best = fminf(best, c);
            }
            d += 40.0f;
          }
          best_j = static_cast<int>(fi + bd);
        }
        // the five keys of a state -> its lane sub == 0: neighbours (row_shl:1), pairs (row_shl:2), the fifth
        unsigned long long key = flat_key(best, best_j);
        const unsigned long long own = key;
        key = key_min_shl<0x101>(key, key);
        key = key_min_shl<0x102>(key, key);
        key = key_min_shl<0x104>(key, own);
        if (sub == 0 && i_rep < S) {
          // (only if every cost was NaN / inf: the lower end of the window, which lane sub == 0 still holds)
          if (static_cast<unsigned>(key) >= static_cast<unsigned>(S)) key = (key & 0xffffffff00000000ull) | static_cast<unsigned>(lo);
          sh.slots[flat_slot(i_rep)] = key;
        }
      }
      wave_sync();
    }
    flat_level<32, 8, 0>(sh, S, factor, lane, fj, jf0);
    flat_level<8, 4, 32>(sh, S, factor, lane, fj, jf0);
    flat_level<4, 1, 8>(sh, S, factor, lane, fj, jf0);
    // ---- new forward costs: best + local cost, minus their minimum ----------------------------------------
    float nx[NK], lane_min = FLT_MAX;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const unsigned long long key = sh.slots[flat_slot(col[k])];
      bpv[k] = static_cast<int>(static_cast<unsigned>(key));
      nx[k] = __builtin_bit_cast(float, static_cast<unsigned>(key >> 32)) + local[k];
      lane_min = fminf(lane_min, nx[k]);
    }
    const float mn = wave_min_f(lane_min);
    wave_sync();
#pragma unroll
    for (int k = 0; k < NK; ++k) sh.fwd[col[k]] = nx[k] + (-mn);
  }
  if (T > 0) {
    int16_t* __restrict__ bp_row = bp + (T - 1) * S;
#pragma unroll
    for (int k = 0; k < NK; ++k) bp_row[col[k]] = static_cast<int16_t>(bpv[k]);
  }
  wave_sync();
}

}  // namespace

__global__ __launch_bounds__(kVitWaves * 64, 4) void pitch_viterbi_kernel(
    const PitchDevTables t, const PitchBatch b, const float* __restrict__ nccf_res,
    const float* __restrict__ anp, const float* __restrict__ ub, int16_t* __restrict__ backptr,
    int32_t* __restrict__ states, const float* __restrict__ pov_all, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = t.num_states, L = t.num_lags, S4 = (S + 3) & ~3;
  float* st_lag = reinterpret_cast<float*>(smem);
  for (int s = threadIdx.x; s < S; s += blockDim.x) st_lag[s] = t.lags[s];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t slot = static_cast<int64_t>(blockIdx.x) * kVitWaves + wid;
  if (slot >= b.n_utts) return;
  const int64_t u = b.order ? b.order[slot] : slot;
  const int64_t f0 = b.frame_offsets[u], T = b.frame_offsets[u + 1] - f0;
  if (T <= 0) return;
  const int64_t T1 = b.frames_phase1[u];
  const int per_wave = 3 * S4 + kFwdPad + kQueueFloats;  // (S4 and kFwdPad are multiples of 4: 16-byte queue)
  VitShared sh;
  sh.fwd = st_lag + S4 + wid * per_wave;
  sh.nxt = sh.fwd + S4 + kFwdPad;
  sh.bpw = reinterpret_cast<int*>(sh.nxt + S4);
  sh.queue = reinterpret_cast<int4*>(sh.bpw + S4);
  int16_t* __restrict__ bp = backptr + f0 * S;
  const float* __restrict__ res = nccf_res + f0 * static_cast<int64_t>(S);
  const float* __restrict__ pov_nccf = pov_all + f0 * L;
  const float* __restrict__ o = ub + u * 6;

  // InputFinished(): Kaldi runs RecomputeBacktraces when the utterance is shorter than recompute_frame and
  // some frame saw a mean-square energy more than 1 % away from the final one (pitch_stats_kernel).  That
  // pass starts from zero forward costs and overwrites every backpointer: nothing of the first (online)
  // pass survives it, so an offline batch runs ONE forward pass - the recomputing one when Kaldi would
  // recompute, the plain one otherwise (round 2 ran both: half of the Viterbi kernel's time).
  // (... or when frame recompute_frame - 1 falls into the frames the flush adds: T1 < recompute_frame <= T,
  // utterances of 500 - 502 frames; frames >= T1 rescale by exactly 1, so it is the same pass)
  const bool recompute = (T < t.recompute_frame || T1 < t.recompute_frame) && o[5] != 0.0f;
  const float ob1 = recompute ? o[2] : 0.0f, ob2 = recompute ? o[3] : 0.0f, nb = recompute ? o[4] : 0.0f;
  if (S <= 448) viterbi_forward<7>(t, res, anp + f0, T, T1, recompute, ob1, ob2, nb, bp, sh, st_lag, lane);
  else if (S <= 512) viterbi_forward<8>(t, res, anp + f0, T, T1, recompute, ob1, ob2, nb, bp, sh, st_lag, lane);
  else viterbi_forward<0>(t, res, anp + f0, T, T1, recompute, ob1, ob2, nb, bp, sh, st_lag, lane);

  viterbi_traceback(S, T, f0, bp, states, sh, lane);
  wave_sync();
  __threadfence_block();
  pitch_output_rows(t, L, T, f0, states, pov_nccf, out, lane, 64);
}

// one wavefront per utterance with the lane-per-candidate search (viterbi_forward_flat): 129 .. 448 states
__global__ __launch_bounds__(kFlatWaves * 64, 4) void pitch_viterbi_flat_kernel(
    const PitchDevTables t, const PitchBatch b, const float* __restrict__ nccf_res,
    const float* __restrict__ anp, const float* __restrict__ ub, int16_t* __restrict__ backptr,
    int32_t* __restrict__ states, const float* __restrict__ pov_all, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = t.num_states, L = t.num_lags, S4 = (S + 3) & ~3;
  float* st_lag = reinterpret_cast<float*>(smem);
  for (int s = threadIdx.x; s < S; s += blockDim.x) st_lag[s] = t.lags[s];
  __syncthreads();
  // (the wave's index as a SCALAR: the utterance, its offsets and the row pointers derived from them then live in
  // scalar registers - 128 -> 124 vector registers and no spill left in the kernel)
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int64_t slot = static_cast<int64_t>(blockIdx.x) * kFlatWaves + wid;
  if (slot >= b.n_utts) return;
  // (values loaded at a uniform address arrive in vector registers: back to scalars, see wid)
  const int64_t u = uniform_i64(b.order ? b.order[slot] : slot);
  const int64_t f0 = uniform_i64(b.frame_offsets[u]), T = uniform_i64(b.frame_offsets[u + 1]) - f0;
  if (T <= 0) return;
  const int64_t T1 = uniform_i64(b.frames_phase1[u]);
  // per wave: keys (8-byte aligned: first), forward costs, marks
  char* mine = smem + ((S4 * 4 + 15) & ~15) + wid * kFlatWaveBytes;
  FlatShared fs;
  fs.slots = reinterpret_cast<unsigned long long*>(mine);
  fs.fwd = reinterpret_cast<float*>(mine + kFlatSlots * 8);
  fs.marks = reinterpret_cast<unsigned*>(mine + kFlatSlots * 8 + (448 + kFwdPad) * 4);
  int16_t* __restrict__ bp = backptr + f0 * S;
  const float* __restrict__ res = nccf_res + f0 * static_cast<int64_t>(S);
  const float* __restrict__ pov_nccf = pov_all + f0 * L;
  const float* __restrict__ o = ub + u * 6;
  const bool recompute = (T < t.recompute_frame || T1 < t.recompute_frame) && o[5] != 0.0f;   // (see pitch_viterbi_kernel)
  const float ob1 = recompute ? o[2] : 0.0f, ob2 = recompute ? o[3] : 0.0f, nb = recompute ? o[4] : 0.0f;
  viterbi_forward_flat(t, res, anp + f0, T, T1, recompute, ob1, ob2, nb, bp, fs, st_lag, lane);
  // traceback: its scratch (64 ints) goes where the marks were
  VitShared sh;
  sh.fwd = fs.fwd;
  sh.nxt = reinterpret_cast<float*>(fs.marks);
  sh.bpw = nullptr;
  sh.queue = nullptr;
  viterbi_traceback(S, T, f0, bp, states, sh, lane);
  wave_sync();
  __threadfence_block();
  pitch_output_rows(t, L, T, f0, states, pov_nccf, out, lane, 64);
}

// W waves per utterance, one utterance per workgroup (viterbi_forward_team); up to 512 states
template <int W>
__global__ __launch_bounds__(W * 64, 4) void pitch_viterbi_team_kernel(
    const PitchDevTables t, const PitchBatch b, const float* __restrict__ nccf_res,
    const float* __restrict__ anp, const float* __restrict__ ub, int16_t* __restrict__ backptr,
    int32_t* __restrict__ states, const float* __restrict__ pov_all, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = t.num_states, L = t.num_lags, S4 = (S + 3) & ~3;
  float* st_lag = reinterpret_cast<float*>(smem);
  for (int s = threadIdx.x; s < S; s += blockDim.x) st_lag[s] = t.lags[s];
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t slot = blockIdx.x;
  const int64_t u = b.order ? b.order[slot] : slot;
  const int64_t f0 = b.frame_offsets[u], T = b.frame_offsets[u + 1] - f0;
  if (T <= 0) return;   // (the whole workgroup: one utterance)
  const int64_t T1 = b.frames_phase1[u];
  VitShared sh;
  sh.fwd = st_lag + S4;
  sh.nxt = sh.fwd + S4 + kFwdPad;
  sh.bpw = reinterpret_cast<int*>(sh.nxt + S4);
  sh.queue = reinterpret_cast<int4*>(sh.bpw + S4);          // W queues, then W floats
  float* red = reinterpret_cast<float*>(sh.queue + W * kQueueEntries);
  int16_t* __restrict__ bp = backptr + f0 * S;
  const float* __restrict__ res = nccf_res + f0 * static_cast<int64_t>(S);
  const float* __restrict__ o = ub + u * 6;
  // (which pass runs: see pitch_viterbi_kernel)
  const bool recompute = (T < t.recompute_frame || T1 < t.recompute_frame) && o[5] != 0.0f;
  const float ob1 = recompute ? o[2] : 0.0f, ob2 = recompute ? o[3] : 0.0f, nb = recompute ? o[4] : 0.0f;
  viterbi_forward_team<W>(t, res, anp + f0, T, T1, recompute, ob1, ob2, nb, bp, sh, sh.queue + wid * kQueueEntries,
                          red, st_lag, lane, wid);
  __threadfence_block();
  if (wid == 0) viterbi_traceback(S, T, f0, bp, states, sh, lane);
  __threadfence_block();
  __syncthreads();
  pitch_output_rows(t, L, T, f0, states, pov_all + f0 * L, out, static_cast<int>(threadIdx.x), 64 * W);
}

int launch_pitch(const PitchDevTables& t, const PitchBatch& b, const PitchScratch& w, float* out,
                 hipStream_t stream) {
  if (b.total_frames <= 0) return SNF_OK;
  if (t.num_states > 32767) return set_error(SNF_E_RUNTIME, "too many pitch states (delta_pitch too small)");
  if (b.max_down > 0x7fffff00)
    return set_error(SNF_E_RUNTIME, "pitch: an utterance of more than 2^31 resampled samples");
  if (b.total_down > 0) {
    const int threads = 256;
    // grid.y is limited to 65535: utterances are launched in slices
    for (int64_t u0 = 0; u0 < b.n_utts; u0 += 65535) {
      PitchBatch bs = b;
      bs.sample_offsets = b.sample_offsets + u0;
      bs.down_offsets = b.down_offsets + u0;
      const int64_t nu = b.n_utts - u0 < 65535 ? b.n_utts - u0 : 65535;
      // LDS: the input span of 256 outputs (ceil(256 in_unit / out_unit) + a unit of slack + one filter)
      const size_t span_lds = sizeof(float) * (static_cast<size_t>(threads + t.rs_out_unit) * t.rs_in_unit /
                                                   t.rs_out_unit + t.rs_in_unit + t.rs_max_taps + 8);
      if (span_lds > 64 * 1024)
        return set_error(SNF_E_RUNTIME, "pitch resampler: the input span of a workgroup does not fit in LDS");
      hipLaunchKernelGGL(pitch_resample_kernel,
                         dim3(static_cast<unsigned>((b.max_down + threads * kRsChunks - 1) / (threads * kRsChunks)),
                              static_cast<unsigned>(nu)),
                         dim3(threads), span_lds, stream, t, bs, w.down);
      SNF_HIP_CHECK(hipGetLastError());
    }
  }
  hipLaunchKernelGGL(pitch_stats_kernel, dim3(static_cast<unsigned>(b.n_utts)), dim3(256), 0, stream,
                     t, b, w.down, w.ub);
  SNF_HIP_CHECK(hipGetLastError());
  static const int trace = getenv("SNF_PITCH_TRACE") ? atoi(getenv("SNF_PITCH_TRACE")) : 0;  // developer knob
  auto stage_done = [&](const char* name, int index) -> int {
    if (!trace) return 0;
    SNF_HIP_CHECK(hipStreamSynchronize(stream));
    fprintf(stderr, "[snf pitch] %s done\n", name);
    return trace == index ? 1 : 0;
  };
  if (stage_done("resample + stats", 1)) return SNF_OK;
  const int S = t.num_states, L = t.num_lags, S4 = (S + 3) & ~3;
  if (t.full_len >= 65536) return set_error(SNF_E_RUNTIME, "pitch: a correlation window of more than 65535 samples");
  hipLaunchKernelGGL(pitch_frame_meta_kernel, dim3(static_cast<unsigned>((b.total_frames + 255) / 256)),
                     dim3(256), 0, stream, t, b, w.ub, w.frame_meta);
  SNF_HIP_CHECK(hipGetLastError());
  {
    const int WL = (t.full_len + 16 + 3) & ~3, LN = (L + t.ar_quad_taps + 3) & ~3;
    const size_t lds = sizeof(float) * (static_cast<size_t>(t.ar_groups) * t.ar_quad_taps * 64 +
                                        t.ar_groups * 16 +
                                        static_cast<size_t>(kNccfWaves) * 4 * nccf_layout(WL, LN, L).pitch);
    if (lds > 160 * 1024) return set_error(SNF_E_RUNTIME, "pitch lag tables do not fit in LDS");
    if (lds > 64 * 1024)
      SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pitch_nccf_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    const int64_t n_sets = (b.total_frames + 3) / 4;
    int64_t blocks = (n_sets + kNccfWaves - 1) / kNccfWaves;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride: the taps are staged once per workgroup
    hipLaunchKernelGGL(pitch_nccf_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kNccfWaves * 64), lds,
                       stream, t, b, w.down, w.frame_meta, w.nccf_res, w.pov_nccf, w.anp);
    SNF_HIP_CHECK(hipGetLastError());
    if (stage_done("nccf", 2)) return SNF_OK;
  }
  {
    // Small batches: several waves per utterance (viterbi_forward_team) - one wave per utterance leaves most
    // SIMDs with a single wave whose latency is the kernel's time.  SNF_PITCH_TEAM=1|2|4 forces a form.
    const char* knob = getenv("SNF_PITCH_TEAM");   // (read per call: the tests run every form in one process)
    const int forced = knob ? atoi(knob) : 0;
    int team = forced == 1 || forced == 2 || forced == 4 ? forced
               : b.n_utts <= kTeam4Utts ? 4 : b.n_utts <= kTeam2Utts ? 2 : 1;
    if (S > 512 || S < 128) team = 1;
    if (team > 1) {
      const size_t lds = sizeof(float) * (4 * static_cast<size_t>(S4) + kFwdPad + team * (kQueueFloats + 1));
      auto kern = team == 4 ? pitch_viterbi_team_kernel<4> : pitch_viterbi_team_kernel<2>;
      hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(b.n_utts)), dim3(team * 64), lds, stream, t, b,
                         w.nccf_res, w.anp, w.ub, w.backptr, w.states, w.pov_nccf, out);
      SNF_HIP_CHECK(hipGetLastError());
    } else if (S > 128 && S <= 448 && !(getenv("SNF_PITCH_FLAT") && getenv("SNF_PITCH_FLAT")[0] == '0')) {
      // the lane-per-candidate search (round 5): bit-identical to the lane-per-state kernel, 25 % fewer
      // instructions, 11.8 against 14.2 ms per 10 000 utterances; SNF_PITCH_FLAT=0 keeps the old kernel (A/B runs,
      // tests/test_parity_gpu.py::test_pitch_flat_search)
      const size_t lds = ((static_cast<size_t>(S4) * 4 + 15) & ~static_cast<size_t>(15)) +
                         static_cast<size_t>(kFlatWaves) * kFlatWaveBytes;
      if (lds > 64 * 1024)
        SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pitch_viterbi_flat_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      const unsigned blocks = static_cast<unsigned>((b.n_utts + kFlatWaves - 1) / kFlatWaves);
      hipLaunchKernelGGL(pitch_viterbi_flat_kernel, dim3(blocks), dim3(kFlatWaves * 64), lds, stream, t, b,
                         w.nccf_res, w.anp, w.ub, w.backptr, w.states, w.pov_nccf, out);
      SNF_HIP_CHECK(hipGetLastError());
    } else {
      const size_t lds = sizeof(float) * (S4 + static_cast<size_t>(kVitWaves) * (3 * S4 + kFwdPad + kQueueFloats));
      if (lds > 160 * 1024) return set_error(SNF_E_RUNTIME, "pitch state space does not fit in LDS");
      if (lds > 64 * 1024)
        SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pitch_viterbi_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
      const unsigned blocks = static_cast<unsigned>((b.n_utts + kVitWaves - 1) / kVitWaves);
      hipLaunchKernelGGL(pitch_viterbi_kernel, dim3(blocks), dim3(kVitWaves * 64), lds, stream, t, b,
                         w.nccf_res, w.anp, w.ub, w.backptr, w.states, w.pov_nccf, out);
      SNF_HIP_CHECK(hipGetLastError());
    }
    (void)stage_done("viterbi", 3);
  }
  return SNF_OK;
}

}  // namespace snf
