// Kaldi pitch tracker on gfx950: the reference reaches it through
// kaldi.feat.pitch.compute_kaldi_pitch (shennong/processor/pitch_kaldi.py:296-299).
//
// Restates [KALDI-UPSTREAM] pitch-functions.cc (OnlinePitchFeatureImpl::AcceptWaveform /
// InputFinished / RecomputeBacktraces, ComputeCorrelation, ComputeNccf, ComputeLocalCost,
// PitchFrameInfo::ComputeBacktraces) and resample.cc (LinearResample, ArbitraryResample) for the
// offline single-chunk call the reference makes (frames_per_chunk = 0).
//
// Pipeline (three launches per batch):
//   1. pitch_resample_kernel   one thread per downsampled sample: 16 kHz -> 4 kHz windowed-sinc FIR
//   2. pitch_stats_kernel      one workgroup per utterance: signal sum / sum of squares (ballast)
//   3. pitch_track_kernel      one workgroup per utterance: per frame NCCF at the integer lags
//      (batched-lag correlation in LDS) -> sinc resampling to the log-spaced lags -> Viterbi forward
//      step over all states in parallel; then traceback and the POV NCCF of the chosen lags.
// The Viterbi recursion is sequential in time, so an utterance stays on one CU; utterances are the
// parallel axis (10 000+ per launch).
#include <float.h>

#include <cstdlib>

#include "snf_internal.h"

namespace snf {

namespace {

__device__ __forceinline__ int64_t find_utt(const int64_t* __restrict__ offsets, int64_t n,
                                            int64_t g) {
  int64_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

constexpr int kTrackThreads = 512;

}  // namespace

// ---- 1. LinearResample ---------------------------------------------------------------------------
__global__ void pitch_resample_kernel(const PitchDevTables t, const PitchBatch b,
                                      float* __restrict__ down) {
  // blockIdx.y = utterance (no per-thread search), blockIdx.x = 256-sample chunk of its output
  const int64_t u = blockIdx.y;
  const int64_t d0 = b.down_offsets[u], nd = b.down_offsets[u + 1] - d0;
  const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= nd) return;
  const int64_t s0 = b.sample_offsets[u], n = b.sample_offsets[u + 1] - s0;
  const int16_t* __restrict__ w = b.wave + s0;
  const int64_t unit = k / t.rs_out_unit;
  const int wrapped = static_cast<int>(k - unit * t.rs_out_unit);
  const int64_t first_in = t.rs_first[wrapped] + unit * t.rs_in_unit;
  const float* __restrict__ wt = t.rs_w + wrapped * t.rs_max_taps;
  const int ntaps = t.rs_ntaps[wrapped];
  float s = 0.0f;
  for (int i = 0; i < ntaps; ++i) {
    const int64_t j = first_in + i;
    if (j >= 0 && j < n) s += wt[i] * static_cast<float>(w[j]);
  }
  down[d0 + k] = s;
}

// ---- 2. signal statistics for the NCCF ballast -----------------------------------------------------
// stats[u] = {sumsq_phase1, sum_phase1, sumsq_all, sum_all}: Kaldi accumulates float BLAS dot/sum of
// each chunk into doubles; phase 1 = what the resampler emitted before the flush.
__global__ void pitch_stats_kernel(const PitchBatch b, const float* __restrict__ down,
                                   double* __restrict__ stats) {
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
    const double fsq2 = static_cast<float>(d), fs2 = static_cast<float>(e);
    stats[u * 4 + 0] = fsq1;
    stats[u * 4 + 1] = fs1;
    stats[u * 4 + 2] = fsq1 + fsq2;
    stats[u * 4 + 3] = fs1 + fs2;
  }
}

// ---- 3. NCCF + Viterbi ------------------------------------------------------------------------------
namespace {

struct TrackShared {
  float* win;      // [full_len]
  float* nccf;     // [num_lags]      nccf_pitch at the integer lags
  float* norm;     // [num_lags]      e1*e2 (for avg_norm_prod in the recompute pass)
  float* fwd;      // [num_states]
  float* nxt;      // [num_states]
  float* red;      // [16]
  float* part_e2;  // [chunks][num_lags] partial sums of the lag correlation
  float* part_ip;  // [chunks][num_lags]
  float* rep_cost; // [16] Viterbi: best cost of the representative states 0, 32, 64, ...
  int* rep_bp;     // [16] and their backpointers
};

__device__ __forceinline__ float block_min(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float m = red[0];
  for (int i = 1; i < nw; ++i) m = fminf(m, red[i]);
  return m;
}

// loads frame t of the utterance into sh.win, removes the mean of its first win_size samples and
// returns e1 = sum of squares of those samples (every wave computes the two reductions redundantly)
__device__ __forceinline__ float load_frame(const PitchDevTables& t, const float* __restrict__ x,
                                            int64_t nd, int64_t frame, float* win) {
  int64_t start;
  if (t.snip_edges) start = frame * t.win_shift;
  else start = static_cast<int64_t>((static_cast<double>(frame) + 0.5) * t.win_shift) - t.full_len / 2;
  __syncthreads();
  for (int i = threadIdx.x; i < t.full_len; i += blockDim.x) {
    const int64_t k = start + i;
    win[i] = (k >= 0 && k < nd) ? x[k] : 0.0f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float s = 0.0f;
  for (int i = lane; i < t.win_size; i += 64) s += win[i];
  const float neg_mean = -wave_sum_f(s) / static_cast<float>(t.win_size);
  __syncthreads();
  for (int i = threadIdx.x; i < t.full_len; i += blockDim.x) win[i] += neg_mean;
  __syncthreads();
  float e = 0.0f;
  for (int i = lane; i < t.win_size; i += 64) e += win[i] * win[i];
  return wave_sum_f(e);
}

// one forward (Viterbi) pass over all frames; returns with sh.fwd = final normalised forward cost
__device__ void forward_pass(const PitchDevTables& t, const float* __restrict__ x, int64_t nd,
                             int64_t T, int64_t T1, double ms1, double ms2, bool rescale,
                             float new_ballast, int16_t* __restrict__ bp, const TrackShared& sh) {
  const int S = t.num_states, L = t.num_lags, W = t.win_size;
  for (int s = threadIdx.x; s < S; s += blockDim.x) sh.fwd[s] = 0.0f;
  // per-state constants of the lag resampler stay in registers across the frame loop when every
  // thread owns at most one state
  constexpr int kRegTaps = 12;
  const bool reg_taps = S <= static_cast<int>(blockDim.x) && t.ar_max_taps <= kRegTaps;
  float wreg[kRegTaps];
  int my_first = 0, my_n = 0;
  float my_lag = 0.0f;
  if (reg_taps && static_cast<int>(threadIdx.x) < S) {
    my_first = t.ar_first[threadIdx.x];
    my_n = t.ar_n[threadIdx.x];
    my_lag = t.lags[threadIdx.x];
#pragma unroll
    for (int j = 0; j < kRegTaps; ++j)
      wreg[j] = j < my_n ? t.ar_w[threadIdx.x * t.ar_max_taps + j] : 0.0f;
  }
  for (int64_t frame = 0; frame < T; ++frame) {
    const double ms = frame < T1 ? ms1 : ms2;
    const float ballast = static_cast<float>(pow(ms * W, 2.0) * static_cast<double>(t.nccf_ballast));
    const float e1 = load_frame(t, x, nd, frame, sh.win);
    // batched-lag correlation: (lag, sample chunk) pairs over the whole workgroup, then one thread per
    // lag adds the chunk partials in a fixed order (deterministic)
    const int chunks = blockDim.x / L > 0 ? blockDim.x / L : 1;
    const int chunk_len = (W + chunks - 1) / chunks;
    for (int idx = threadIdx.x; idx < L * chunks; idx += blockDim.x) {
      const int l = idx % L, c = idx / L;
      const int i0 = c * chunk_len, i1 = i0 + chunk_len < W ? i0 + chunk_len : W;
      const float* __restrict__ a = sh.win;
      const float* __restrict__ cw = sh.win + t.first_lag + l;
      float e2 = 0.0f, ip = 0.0f;
      for (int i = i0; i < i1; ++i) {
        e2 += cw[i] * cw[i];
        ip += a[i] * cw[i];
      }
      sh.part_e2[c * L + l] = e2;
      sh.part_ip[c * L + l] = ip;
    }
    __syncthreads();
    for (int l = threadIdx.x; l < L; l += blockDim.x) {
      float e2 = 0.0f, ip = 0.0f;
      for (int c = 0; c < chunks; ++c) {
        e2 += sh.part_e2[c * L + l];
        ip += sh.part_ip[c * L + l];
      }
      const float norm = e1 * e2;
      const float den = static_cast<float>(sqrt(static_cast<double>(norm + ballast)));
      sh.nccf[l] = den != 0.0f ? ip / den : 0.0f;
      sh.norm[l] = norm;
    }
    __syncthreads();
    float scale = 1.0f;
    if (rescale) {
      float sum = 0.0f;
      for (int l = 0; l < L; ++l) sum += sh.norm[l];
      const float avg_norm_prod = sum / static_cast<float>(L);
      const float old_ms = static_cast<float>(ms);
      const float old_ballast =
          static_cast<float>(pow(static_cast<double>(old_ms) * W, 2.0) * static_cast<double>(t.nccf_ballast));
      scale = powf((old_ballast + avg_norm_prod) / (new_ballast + avg_norm_prod), 0.5f);
    }
    // ---- Viterbi step.  cost(i, j) = (j - i)^2 * factor + fwd[j]; its argmin is monotone in i (Kaldi's
    // search relies on the same property), so: (1) exact argmin for the representative states
    // 0, 32, 64, ... (32 lanes per representative, strided scan + lane reduction, lowest index wins
    // ties); (2) every other state scans only between the backpointers of its two neighbouring
    // representatives.  No FMA contraction: costs must round like Kaldi's.
    const bool mono = S <= 512 && blockDim.x >= 512;
    if (mono) {
      const int rep = threadIdx.x >> 5, lane32 = threadIdx.x & 31;
      const int i_rep = rep << 5;
      float best = FLT_MAX;
      int best_j = 0;
      if (i_rep < S) {
        const float fi = static_cast<float>(i_rep);
        for (int j = lane32; j < S; j += 32) {
          const float d = static_cast<float>(j) - fi;
          const float c = __fadd_rn(__fmul_rn(d * d, t.inter_frame_factor), sh.fwd[j]);
          if (c < best) { best = c; best_j = j; }
        }
      }
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        const float oc = __shfl_xor(best, off, 32);
        const int oj = __shfl_xor(best_j, off, 32);
        if (oc < best || (oc == best && oj < best_j)) { best = oc; best_j = oj; }
      }
      if (lane32 == 0 && i_rep < S) { sh.rep_cost[rep] = best; sh.rep_bp[rep] = best_j; }
      __syncthreads();
    }
    // sinc-resample the NCCF to the log-spaced lags, local cost, Viterbi step
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
      float v = 0.0f, lag_s;
      if (reg_taps) {
        const float* __restrict__ src = sh.nccf + my_first;
#pragma unroll
        for (int j = 0; j < kRegTaps; ++j)
          if (j < my_n) v += src[j] * wreg[j];
        lag_s = my_lag;
      } else {
        const float* __restrict__ wt = t.ar_w + s * t.ar_max_taps;
        const float* __restrict__ src = sh.nccf + t.ar_first[s];
        const int n = t.ar_n[s];
        for (int j = 0; j < n; ++j) v += src[j] * wt[j];
        lag_s = t.lags[s];
      }
      if (rescale) v *= scale;
      float local = 1.0f - v;
      local += t.soft_min_f0 * lag_s * v;
      const float fs = static_cast<float>(s);
      float best;
      int best_j;
      if (mono) {
        const int k = s >> 5;
        if ((s & 31) == 0) {
          best = sh.rep_cost[k];
          best_j = sh.rep_bp[k];
        } else {
          const int lo = sh.rep_bp[k];
          const int hi = ((k + 1) << 5) < S ? sh.rep_bp[k + 1] : S - 1;
          const float d0 = static_cast<float>(lo) - fs;
          best = __fadd_rn(__fmul_rn(d0 * d0, t.inter_frame_factor), sh.fwd[lo]);
          best_j = lo;
          for (int j = lo + 1; j <= hi; ++j) {
            const float d = static_cast<float>(j) - fs;
            const float c = __fadd_rn(__fmul_rn(d * d, t.inter_frame_factor), sh.fwd[j]);
            if (c < best) { best = c; best_j = j; }
          }
        }
      } else {
        // exact argmin over all states; lowest index wins ties
        best = __fadd_rn(__fmul_rn(fs * fs, t.inter_frame_factor), sh.fwd[0]);
        best_j = 0;
        for (int j = 1; j < S; ++j) {
          const float d = static_cast<float>(j) - fs;
          const float c = __fadd_rn(__fmul_rn(d * d, t.inter_frame_factor), sh.fwd[j]);
          if (c < best) { best = c; best_j = j; }
        }
      }
      sh.nxt[s] = __fadd_rn(best, local);
      bp[frame * S + s] = static_cast<int16_t>(best_j);
    }
    float m = FLT_MAX;
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += blockDim.x) m = fminf(m, sh.nxt[s]);
    m = block_min(m, sh.red);
    for (int s = threadIdx.x; s < S; s += blockDim.x) sh.fwd[s] = sh.nxt[s] + (-m);
    __syncthreads();
  }
}

}  // namespace

__global__ __launch_bounds__(kTrackThreads) void pitch_track_kernel(
    const PitchDevTables t, const PitchBatch b, const float* __restrict__ down,
    const double* __restrict__ stats, int16_t* __restrict__ backptr, int32_t* __restrict__ states,
    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int64_t u = blockIdx.x;
  const int64_t f0 = b.frame_offsets[u], T = b.frame_offsets[u + 1] - f0;
  if (T <= 0) return;
  const int64_t T1 = b.frames_phase1[u];
  const int64_t d0 = b.down_offsets[u], nd = b.down_offsets[u + 1] - d0, nd1 = b.down_phase1[u];
  const float* __restrict__ x = down + d0;
  const int S = t.num_states, L = t.num_lags, W = t.win_size;
  TrackShared sh;
  sh.win = reinterpret_cast<float*>(smem);
  sh.nccf = sh.win + ((t.full_len + 3) & ~3);
  sh.norm = sh.nccf + ((L + 3) & ~3);
  sh.fwd = sh.norm + ((L + 3) & ~3);
  sh.nxt = sh.fwd + ((S + 3) & ~3);
  sh.red = sh.nxt + ((S + 3) & ~3);
  const int chunks = kTrackThreads / L > 0 ? kTrackThreads / L : 1;
  sh.part_e2 = sh.red + 16;
  sh.part_ip = sh.part_e2 + chunks * L;
  sh.rep_cost = sh.part_ip + chunks * L;
  sh.rep_bp = reinterpret_cast<int*>(sh.rep_cost + 16);
  int16_t* __restrict__ bp = backptr + f0 * S;

  const double sq1 = stats[u * 4 + 0], s1 = stats[u * 4 + 1], sq2 = stats[u * 4 + 2],
               s2 = stats[u * 4 + 3];
  const double n1 = static_cast<double>(nd1), n2 = static_cast<double>(nd);
  const double ms1 = nd1 > 0 ? sq1 / n1 - pow(s1 / n1, 2.0) : 0.0;
  const double ms2 = sq2 / n2 - pow(s2 / n2, 2.0);

  forward_pass(t, x, nd, T, T1, ms1, ms2, false, 0.0f, bp, sh);

  // InputFinished(): RecomputeBacktraces when the utterance is shorter than recompute_frame and some
  // frame saw a mean-square energy more than 1 % away from the final one
  if (T < t.recompute_frame && T1 > 0) {
    const double mean = s2 / n2;
    const float ms_final = static_cast<float>(sq2 / n2 - mean * mean);
    const float a = static_cast<float>(ms1);
    const bool approx_equal = (a == ms_final) || (fabsf(a - ms_final) <= 0.01f * (fabsf(a) + fabsf(ms_final)));
    if (!approx_equal) {
      const float new_ballast =
          static_cast<float>(pow(static_cast<double>(ms_final) * W, 2.0) * static_cast<double>(t.nccf_ballast));
      __syncthreads();
      forward_pass(t, x, nd, T, T1, ms1, ms2, true, new_ballast, bp, sh);
    }
  }

  // traceback (sequential chain of dependent loads; L2-resident)
  __syncthreads();
  if (threadIdx.x == 0) {
    int best = 0;
    float bv = sh.fwd[0];
    for (int s = 1; s < S; ++s)
      if (sh.fwd[s] < bv) { bv = sh.fwd[s]; best = s; }
    for (int64_t frame = T - 1; frame >= 0; --frame) {
      states[f0 + frame] = best;
      best = bp[frame * S + best];
    }
  }
  __syncthreads();
  __threadfence_block();

  // output rows: (POV NCCF resampled at the chosen lag, 1 / lag); the POV NCCF (ballast 0) is only
  // needed at the chosen state, so it is recomputed here instead of being stored for all states
  for (int64_t frame = threadIdx.x; frame < T; frame += blockDim.x) {
    const int s = states[f0 + frame];
    int64_t start;
    if (t.snip_edges) start = frame * t.win_shift;
    else start = static_cast<int64_t>((static_cast<double>(frame) + 0.5) * t.win_shift) - t.full_len / 2;
    float sum = 0.0f;
    for (int i = 0; i < W; ++i) {
      const int64_t k = start + i;
      sum += (k >= 0 && k < nd) ? x[k] : 0.0f;
    }
    const float neg_mean = -sum / static_cast<float>(W);
    float e1 = 0.0f;
    for (int i = 0; i < W; ++i) {
      const int64_t k = start + i;
      const float v = ((k >= 0 && k < nd) ? x[k] : 0.0f) + neg_mean;
      e1 += v * v;
    }
    const float* __restrict__ wt = t.ar_w + s * t.ar_max_taps;
    const int first = t.ar_first[s], n = t.ar_n[s];
    float pov = 0.0f;
    for (int j = 0; j < n; ++j) {
      const int lag = t.first_lag + first + j;
      float e2 = 0.0f, ip = 0.0f;
      for (int i = 0; i < W; ++i) {
        const int64_t ka = start + i, kc = ka + lag;
        const float va = ((ka >= 0 && ka < nd) ? x[ka] : 0.0f) + neg_mean;
        const float vc = ((kc >= 0 && kc < nd && i + lag < t.full_len) ? x[kc] : 0.0f) + neg_mean;
        e2 += vc * vc;
        ip += va * vc;
      }
      const float norm = e1 * e2;
      const float den = static_cast<float>(sqrt(static_cast<double>(norm + 0.0f)));
      const float nccf = den != 0.0f ? ip / den : 0.0f;
      pov += nccf * wt[j];
    }
    out[(f0 + frame) * 2 + 0] = pov;
    out[(f0 + frame) * 2 + 1] = 1.0f / t.lags[s];
  }
}


// ---- 3b. wave-per-utterance tracker ---------------------------------------------------------------
// The Viterbi recursion is sequential over the frames of one utterance but utterances are
// independent, and one frame is only ~30 kflop: a 512-thread workgroup per utterance spends its time
// in workgroup barriers (11 per frame).  Here ONE wavefront owns an utterance (no workgroup barrier in
// the frame loop, 16 utterances in flight per CU); the arithmetic of every frame is the same as in
// pitch_track_kernel above (same summation trees for the frame mean / energy, same Viterbi costs and
// tie-breaks), only the lag correlation adds its four 25-sample partials in a quad tree.
namespace {

constexpr int kWaveTrackWaves = 8;   // utterances (= wavefronts) per workgroup
constexpr int kWaveMaxTaps = 12;

struct WaveShared {
  float* win;    // [full_len]
  float* nccf;   // [num_lags]
  float* norm;   // [num_lags]
  float* fwd;    // [num_states]
  float* nxt;    // [num_states]
  int* bpw;      // [num_states]   backpointers of the current frame
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Cross-lane steps on the VALU (DPP + v_readlane); __shfl_xor lowers to ds_bpermute_b32, an LDS round
// trip per step.  quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E; row_ror:4 = 0x124, row_ror:8 = 0x128
// (rotations reach the other three quads of a 16-lane row).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf,
                                                               0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
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
// (cost, index) argmin step: lower cost wins, lower index on ties
__device__ __forceinline__ void argmin_take(float& c, int& j, float oc, int oj) {
  if (oc < c || (oc == c && oj < j)) { c = oc; j = oj; }
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
  float fj = static_cast<float>(lo);
  {
    const float d = fj - fi;
    best = __fadd_rn(__fmul_rn(d * d, factor), fwd[lo]);
  }
  float best_f = fj;
  for (int j = lo + 1; j <= hi; j += 4) {
    const float f0 = fwd[j], f1 = fwd[j + 1], f2 = fwd[j + 2], f3 = fwd[j + 3];
    const float d0 = (fj + 1.0f) - fi, d1 = (fj + 2.0f) - fi, d2 = (fj + 3.0f) - fi, d3 = (fj + 4.0f) - fi;
    const float c0 = __fadd_rn(__fmul_rn(d0 * d0, factor), f0);
    const float c1 = __fadd_rn(__fmul_rn(d1 * d1, factor), f1);
    const float c2 = __fadd_rn(__fmul_rn(d2 * d2, factor), f2);
    const float c3 = __fadd_rn(__fmul_rn(d3 * d3, factor), f3);
    if (c0 < best) { best = c0; best_f = fj + 1.0f; }
    if (c1 < best) { best = c1; best_f = fj + 2.0f; }
    if (c2 < best) { best = c2; best_f = fj + 3.0f; }
    if (c3 < best) { best = c3; best_f = fj + 4.0f; }
    fj += 4.0f;
  }
  best_j = static_cast<int>(best_f);
}

constexpr int kLagGroup = 5;  // lags per work item of the correlation
constexpr int kCorrChunks = 4;  // window quarters per lag group (the 4 lanes of a quad)
constexpr int kFwdPad = 36;    // FLT_MAX entries behind the forward costs (unclamped scan steps)
constexpr int kLongRange = 8;  // candidate ranges at least this long are scanned by a 16-lane row

__device__ void forward_pass_wave(const PitchDevTables& t, const float* __restrict__ x, int64_t nd,
                                  int64_t T, int64_t T1, double ms1, double ms2, bool rescale,
                                  float new_ballast, int16_t* __restrict__ bp,
                                  float* __restrict__ pov_nccf, const WaveShared& sh,
                                  const float* __restrict__ taps, const int* __restrict__ st_first,
                                  const float* __restrict__ st_lag, const int lane) {
  const int S = t.num_states, L = t.num_lags, W = t.win_size;
  for (int s = lane; s < S; s += 64) sh.fwd[s] = 0.0f;
  for (int s = S + lane; s < S + kFwdPad; s += 64) sh.fwd[s] = FLT_MAX;  // scan read-ahead padding
  for (int i = lane; i < 8; i += 64) sh.win[t.full_len + i] = 0.0f;  // read-ahead padding
  for (int i = lane; i < kWaveMaxTaps; i += 64) sh.nccf[L + i] = 0.0f;  // (taps are zero-padded)
  const float ballast1 = static_cast<float>(pow(ms1 * W, 2.0) * static_cast<double>(t.nccf_ballast));
  const float ballast2 = static_cast<float>(pow(ms2 * W, 2.0) * static_cast<double>(t.nccf_ballast));
  const int chunk_len = (W + kCorrChunks - 1) / kCorrChunks;
  const int groups = (L + kLagGroup - 1) / kLagGroup;
  const int passes = (groups * kCorrChunks + 63) >> 6;
  const float factor = t.inter_frame_factor;
  for (int64_t frame = 0; frame < T; ++frame) {
    const double ms = frame < T1 ? ms1 : ms2;
    const float ballast = frame < T1 ? ballast1 : ballast2;
    // ---- frame window, mean removal, e1 ----------------------------------------------------------
    int64_t start;
    if (t.snip_edges) start = frame * t.win_shift;
    else start = static_cast<int64_t>((static_cast<double>(frame) + 0.5) * t.win_shift) - t.full_len / 2;
    wave_sync();
    for (int i = lane; i < t.full_len; i += 64) {
      const int64_t k = start + i;
      sh.win[i] = (k >= 0 && k < nd) ? x[k] : 0.0f;
    }
    wave_sync();
    float sm = 0.0f;
    for (int i = lane; i < W; i += 64) sm += sh.win[i];
    const float neg_mean = -wave_sum_v(sm) / static_cast<float>(W);
    wave_sync();
    for (int i = lane; i < t.full_len; i += 64) sh.win[i] += neg_mean;
    wave_sync();
    float e = 0.0f;
    for (int i = lane; i < W; i += 64) e += sh.win[i] * sh.win[i];
    const float e1 = wave_sum_v(e);
    // ---- lag correlation.  Work item = (group of 5 consecutive lags, quarter of the window); the
    // 5 x 5 (lag, sample) blocks reuse 9 window values from registers.  The four quarters of a group
    // sit in the four lanes of a quad and are added as (q0 + q1) + (q2 + q3). ----------------------
    for (int pass = 0; pass < passes; ++pass) {
      const int item = (pass << 6) + lane;
      const int g = item >> 2, c = item & 3;
      const int l0 = g * kLagGroup;
      float e2[kLagGroup], ip[kLagGroup];
#pragma unroll
      for (int d = 0; d < kLagGroup; ++d) e2[d] = ip[d] = 0.0f;
      if (g < groups) {
        const int i0 = c * chunk_len, i1 = i0 + chunk_len < W ? i0 + chunk_len : W;
        const float* __restrict__ a = sh.win;
        const float* __restrict__ cw = sh.win + t.first_lag + l0;
        int i = i0;
        for (; i + kLagGroup <= i1; i += kLagGroup) {
          float av[kLagGroup], cv[2 * kLagGroup - 1], sq[2 * kLagGroup - 1];
#pragma unroll
          for (int u = 0; u < kLagGroup; ++u) av[u] = a[i + u];
#pragma unroll
          for (int u = 0; u < 2 * kLagGroup - 1; ++u) cv[u] = cw[i + u];
#pragma unroll
          for (int u = 0; u < 2 * kLagGroup - 1; ++u) sq[u] = cv[u] * cv[u];
#pragma unroll
          for (int u = 0; u < kLagGroup; ++u)
#pragma unroll
            for (int d = 0; d < kLagGroup; ++d) {
              e2[d] += sq[u + d];
              ip[d] += av[u] * cv[u + d];
            }
        }
        for (; i < i1; ++i) {  // window quarters that are not a multiple of 5 samples
          const float ai = a[i];
#pragma unroll
          for (int d = 0; d < kLagGroup; ++d) {
            const float vc = cw[i + d];
            e2[d] += vc * vc;
            ip[d] += ai * vc;
          }
        }
      }
#pragma unroll
      for (int d = 0; d < kLagGroup; ++d) {
        e2[d] += dpp_f<0xB1>(e2[d]);
        ip[d] += dpp_f<0xB1>(ip[d]);
        e2[d] += dpp_f<0x4E>(e2[d]);
        ip[d] += dpp_f<0x4E>(ip[d]);
      }
      // every lane of the quad holds the totals: lane c finishes lag l0 + c (lane 0 also l0 + 4)
#pragma unroll
      for (int d = 0; d < kLagGroup; ++d) {
        const int l = l0 + d;
        if (g < groups && l < L && (d & 3) == c) {
          const float norm = e1 * e2[d];
          const float den = static_cast<float>(sqrt(static_cast<double>(norm + ballast)));
          sh.nccf[l] = den != 0.0f ? ip[d] / den : 0.0f;
          sh.norm[l] = norm;
          if (!rescale) {
            // the POV feature uses the NCCF without ballast; only the lags around the state chosen by
            // the traceback will be read back
            const float den0 = static_cast<float>(sqrt(static_cast<double>(norm + 0.0f)));
            pov_nccf[frame * L + l] = den0 != 0.0f ? ip[d] / den0 : 0.0f;
          }
        }
      }
    }
    wave_sync();
    float scale = 1.0f;
    if (rescale) {
      float sum = 0.0f;
      for (int l = 0; l < L; ++l) sum += sh.norm[l];
      const float avg_norm_prod = sum / static_cast<float>(L);
      const float old_ms = static_cast<float>(ms);
      const float old_ballast =
          static_cast<float>(pow(static_cast<double>(old_ms) * W, 2.0) * static_cast<double>(t.nccf_ballast));
      scale = powf((old_ballast + avg_norm_prod) / (new_ballast + avg_norm_prod), 0.5f);
    }
    // ---- local cost of every state (NCCF resampled at its lag) -> nxt ------------------------------
#pragma unroll 2
    for (int s = lane; s < S; s += 64) {
      float v = 0.0f;
      const float* __restrict__ src = sh.nccf + st_first[s];
      const float4* __restrict__ wt = reinterpret_cast<const float4*>(taps + s * kWaveMaxTaps);
#pragma unroll
      for (int j = 0; j < kWaveMaxTaps / 4; ++j) {  // weights beyond the state's taps are zero
        const float4 wq = wt[j];
        v += src[4 * j] * wq.x;
        v += src[4 * j + 1] * wq.y;
        v += src[4 * j + 2] * wq.z;
        v += src[4 * j + 3] * wq.w;
      }
      if (rescale) v *= scale;
      float local = 1.0f - v;
      local += t.soft_min_f0 * st_lag[s] * v;
      sh.nxt[s] = local;
    }
    // ---- Viterbi step.  cost(i, j) = (j - i)^2 * factor + fwd[j]; its argmin is monotone in i
    // (Kaldi's own search relies on it).  Level 1: exact argmin of the states 0, 32, 64, ... (4 lanes
    // per state, strided scan, lowest index wins ties).  Then the strides 16, 8, 4, 2, 1: every new
    // state scans only between the backpointers of its two already known neighbours. -----------------
    {
      const int rep = lane >> 2, sub = lane & 3;
      const int i_rep = rep << 5;
      float best = FLT_MAX;
      int best_j = 0x7fffffff;
      if (i_rep < S) {
        const float fi = static_cast<float>(i_rep);
        // (the forward costs are padded with FLT_MAX up to a multiple of 32 states)
        float fj = static_cast<float>(sub), best_f = 0.0f;
        for (int j = sub; j < S; j += 32) {
          float ff[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) ff[u] = sh.fwd[j + 4 * u];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const float d = (fj + static_cast<float>(4 * u)) - fi;
            const float c = __fadd_rn(__fmul_rn(d * d, factor), ff[u]);
            if (c < best) { best = c; best_f = fj + static_cast<float>(4 * u); }
          }
          fj += 32.0f;
        }
        best_j = static_cast<int>(best_f);
      }
      quad_argmin(best, best_j);
      wave_sync();  // local costs (nxt) written above are read below
      if (sub == 0 && i_rep < S) {
        sh.bpw[i_rep] = best_j;
        sh.nxt[i_rep] = __fadd_rn(best, sh.nxt[i_rep]);
      }
    }
    for (int h = 16; h >= 1; h >>= 1) {
      wave_sync();
      const int count = (S - h + 2 * h - 1) / (2 * h);  // states h, 3h, 5h, ... < S
      for (int m0 = 0; m0 < count; m0 += 64) {
        const int m = m0 + lane;
        const bool active = m < count;
        const int i = h + 2 * h * (active ? m : 0);
        const int lo = sh.bpw[i - h];
        const int hi = i + h < S ? sh.bpw[i + h] : S - 1;
        const bool is_long = active && hi - lo >= kLongRange;
        float best = FLT_MAX;
        int best_j = lo;
        if (active && !is_long) scan_range(sh.fwd, lo, hi, static_cast<float>(i), factor, best, best_j);
        // a jump of the backpointer function makes one state of every level scan a long range: those
        // are searched four at a time, one per 16-lane row (16 candidates per step + a row argmin)
        unsigned long long pending = __ballot(is_long);
        while (pending) {
          int src[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            src[r] = pending ? __ffsll(static_cast<long long>(pending)) - 1 : -1;
            if (pending) pending &= pending - 1;
          }
          const int row = lane >> 4, sub = lane & 15;
          int ri = 0, rlo = 0, rhi = -1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (src[r] >= 0) {
              const int a = __builtin_amdgcn_readlane(i, src[r]);
              const int b = __builtin_amdgcn_readlane(lo, src[r]);
              const int c = __builtin_amdgcn_readlane(hi, src[r]);
              if (row == r) { ri = a; rlo = b; rhi = c; }
            }
          }
          const float fi = static_cast<float>(ri);
          float cb = FLT_MAX;
          int cj = 0x7fffffff;
          for (int j = rlo + sub; j <= rhi; j += 16) {
            const float c = trans_cost(j, fi, factor, sh.fwd[j]);
            if (c < cb) { cb = c; cj = j; }
          }
          quad_argmin(cb, cj);
          argmin_take(cb, cj, dpp_f<0x124>(cb), dpp_i<0x124>(cj));
          argmin_take(cb, cj, dpp_f<0x128>(cb), dpp_i<0x128>(cj));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (src[r] >= 0) {
              const float c = lane_f(cb, 16 * r);
              const int jn = __builtin_amdgcn_readlane(cj, 16 * r);
              if (lane == src[r]) { best = c; best_j = jn; }
            }
          }
        }
        if (active) {
          sh.bpw[i] = best_j;
          sh.nxt[i] = __fadd_rn(best, sh.nxt[i]);
        }
      }
    }
    wave_sync();
    float lane_min = FLT_MAX;
    for (int s = lane; s < S; s += 64) {
      lane_min = fminf(lane_min, sh.nxt[s]);
      bp[frame * S + s] = static_cast<int16_t>(sh.bpw[s]);
    }
    const float m = wave_min_f(lane_min);
    for (int s = lane; s < S; s += 64) sh.fwd[s] = sh.nxt[s] + (-m);
    wave_sync();
  }
}

}  // namespace

__global__ __launch_bounds__(kWaveTrackWaves * 64, 4) void pitch_track_wave_kernel(
    const PitchDevTables t, const PitchBatch b, const float* __restrict__ down,
    const double* __restrict__ stats, int16_t* __restrict__ backptr, int32_t* __restrict__ states,
    float* __restrict__ pov_all, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int S = t.num_states, L = t.num_lags, W = t.win_size;
  // resampler taps of every state, shared by the wavefronts of the workgroup
  float* taps = reinterpret_cast<float*>(smem);
  const int S4 = (S + 3) & ~3;
  int* st_first = reinterpret_cast<int*>(taps + ((S * kWaveMaxTaps + 3) & ~3));
  float* st_lag = reinterpret_cast<float*>(st_first + S4);
  for (int i = threadIdx.x; i < S * kWaveMaxTaps; i += blockDim.x) {
    const int s = i / kWaveMaxTaps, j = i - s * kWaveMaxTaps;
    taps[i] = (j < t.ar_max_taps && j < t.ar_n[s]) ? t.ar_w[s * t.ar_max_taps + j] : 0.0f;
  }
  for (int s = threadIdx.x; s < S; s += blockDim.x) {
    st_first[s] = t.ar_first[s];
    st_lag[s] = t.lags[s];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t slot = static_cast<int64_t>(blockIdx.x) * kWaveTrackWaves + wid;
  if (slot >= b.n_utts) return;
  const int64_t u = b.order ? b.order[slot] : slot;
  const int64_t f0 = b.frame_offsets[u], T = b.frame_offsets[u + 1] - f0;
  if (T <= 0) return;
  const int64_t T1 = b.frames_phase1[u];
  const int64_t d0 = b.down_offsets[u], nd = b.down_offsets[u + 1] - d0, nd1 = b.down_phase1[u];
  const float* __restrict__ x = down + d0;
  const int per_wave = ((t.full_len + 8 + 3) & ~3) + ((L + kWaveMaxTaps + 3) & ~3) + ((L + 3) & ~3) +
                       3 * ((S + 3) & ~3) + kFwdPad;
  WaveShared sh;
  sh.win = st_lag + S4 + wid * per_wave;
  sh.nccf = sh.win + ((t.full_len + 8 + 3) & ~3);
  sh.norm = sh.nccf + ((L + kWaveMaxTaps + 3) & ~3);
  sh.fwd = sh.norm + ((L + 3) & ~3);
  sh.nxt = sh.fwd + ((S + 3) & ~3) + kFwdPad;
  sh.bpw = reinterpret_cast<int*>(sh.nxt + ((S + 3) & ~3));
  int16_t* __restrict__ bp = backptr + f0 * S;
  float* __restrict__ pov_nccf = pov_all + f0 * L;

  const double sq1 = stats[u * 4 + 0], s1 = stats[u * 4 + 1], sq2 = stats[u * 4 + 2],
               s2 = stats[u * 4 + 3];
  const double n1 = static_cast<double>(nd1), n2 = static_cast<double>(nd);
  const double ms1 = nd1 > 0 ? sq1 / n1 - pow(s1 / n1, 2.0) : 0.0;
  const double ms2 = sq2 / n2 - pow(s2 / n2, 2.0);

  forward_pass_wave(t, x, nd, T, T1, ms1, ms2, false, 0.0f, bp, pov_nccf, sh, taps, st_first, st_lag, lane);

  if (T < t.recompute_frame && T1 > 0) {
    const double mean = s2 / n2;
    const float ms_final = static_cast<float>(sq2 / n2 - mean * mean);
    const float a = static_cast<float>(ms1);
    const bool approx_equal = (a == ms_final) || (fabsf(a - ms_final) <= 0.01f * (fabsf(a) + fabsf(ms_final)));
    if (!approx_equal) {
      const float new_ballast =
          static_cast<float>(pow(static_cast<double>(ms_final) * W, 2.0) * static_cast<double>(t.nccf_ballast));
      wave_sync();
      forward_pass_wave(t, x, nd, T, T1, ms1, ms2, true, new_ballast, bp, pov_nccf, sh, taps, st_first,
                        st_lag, lane);
    }
  }

  // traceback: best final state (lowest index wins ties), then the chain of backpointers
  wave_sync();
  __threadfence_block();
  {
    float bv = FLT_MAX;
    int best = 0x7fffffff;
    for (int s = lane; s < S; s += 64) {
      const float c = sh.fwd[s];
      if (c < bv) { bv = c; best = s; }
    }
    wave_argmin(bv, best);
    if (lane == 0) {
      for (int64_t frame = T - 1; frame >= 0; --frame) {
        states[f0 + frame] = best;
        best = bp[frame * S + best];
      }
    }
  }
  wave_sync();
  __threadfence_block();

  // output rows: (POV NCCF resampled at the chosen lag, 1 / lag)
  for (int64_t frame = lane; frame < T; frame += 64) {
    const int s = states[f0 + frame];
    const float* __restrict__ wt = t.ar_w + s * t.ar_max_taps;
    const float* __restrict__ src = pov_nccf + frame * L + t.ar_first[s];
    const int n = t.ar_n[s];
    float pov = 0.0f;
    for (int j = 0; j < n; ++j) pov += src[j] * wt[j];
    out[(f0 + frame) * 2 + 0] = pov;
    out[(f0 + frame) * 2 + 1] = 1.0f / t.lags[s];
  }
}

int launch_pitch(const PitchDevTables& t, const PitchBatch& b, float* down, double* stats,
                 int16_t* backptr, int32_t* states, float* pov_nccf, float* out, hipStream_t stream) {
  if (b.total_frames <= 0) return SNF_OK;
  if (t.num_states > 32767) return set_error(SNF_E_RUNTIME, "too many pitch states (delta_pitch too small)");
  if (b.total_down > 0) {
    const int threads = 256;
    // grid.y is limited to 65535: utterances are launched in slices
    for (int64_t u0 = 0; u0 < b.n_utts; u0 += 65535) {
      PitchBatch bs = b;
      bs.sample_offsets = b.sample_offsets + u0;
      bs.down_offsets = b.down_offsets + u0;
      const int64_t nu = b.n_utts - u0 < 65535 ? b.n_utts - u0 : 65535;
      hipLaunchKernelGGL(pitch_resample_kernel,
                         dim3(static_cast<unsigned>((b.max_down + threads - 1) / threads),
                              static_cast<unsigned>(nu)),
                         dim3(threads), 0, stream, t, bs, down);
      SNF_HIP_CHECK(hipGetLastError());
    }
  }
  hipLaunchKernelGGL(pitch_stats_kernel, dim3(static_cast<unsigned>(b.n_utts)), dim3(256), 0, stream,
                     b, down, stats);
  SNF_HIP_CHECK(hipGetLastError());
  const bool wave_path = t.num_states <= 512 && t.ar_max_taps <= kWaveMaxTaps &&
                         !getenv("SNF_PITCH_BLOCK_KERNEL");
  if (wave_path) {
    const int per_wave = ((t.full_len + 8 + 3) & ~3) + ((t.num_lags + kWaveMaxTaps + 3) & ~3) +
                         ((t.num_lags + 3) & ~3) + 3 * ((t.num_states + 3) & ~3) + kFwdPad;
    const size_t lds_w = sizeof(float) * (((t.num_states * kWaveMaxTaps + 3) & ~3) +
                                          2 * ((t.num_states + 3) & ~3) +
                                          static_cast<size_t>(kWaveTrackWaves) * per_wave);
    if (lds_w <= 80 * 1024) {
      if (lds_w > 64 * 1024)
        SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pitch_track_wave_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          static_cast<int>(lds_w)));
      const unsigned blocks = static_cast<unsigned>((b.n_utts + kWaveTrackWaves - 1) / kWaveTrackWaves);
      hipLaunchKernelGGL(pitch_track_wave_kernel, dim3(blocks), dim3(kWaveTrackWaves * 64), lds_w, stream,
                         t, b, down, stats, backptr, states, pov_nccf, out);
      SNF_HIP_CHECK(hipGetLastError());
      return SNF_OK;
    }
  }
  const int chunks = kTrackThreads / t.num_lags > 0 ? kTrackThreads / t.num_lags : 1;
  const size_t lds = sizeof(float) * (((t.full_len + 3) & ~3) + 2 * ((t.num_lags + 3) & ~3) +
                                      2 * ((t.num_states + 3) & ~3) + 16 + 2 * chunks * t.num_lags +
                                      32);
  if (lds > 160 * 1024) return set_error(SNF_E_RUNTIME, "pitch state space does not fit in LDS");
  if (lds > 64 * 1024)
    SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pitch_track_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL(pitch_track_kernel, dim3(static_cast<unsigned>(b.n_utts)), dim3(kTrackThreads),
                     lds, stream, t, b, down, stats, backptr, states, out);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

}  // namespace snf
