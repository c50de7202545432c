// Frame-matrix kernels: RASTA filtering and the PLP tail (reference plp.py:64-146, :548-626),
// delta / delta-delta (reference postprocessor/delta.py:129-131 -> [KALDI-UPSTREAM]
// feature-functions.cc DeltaFeatures) and pitch post-processing (reference
// processor/pitch_kaldi.py:535-537 -> [KALDI-UPSTREAM] pitch-functions.cc OnlineProcessPitch).
//
// These are pure HBM-streaming kernels over [frames, cols] float32 matrices: each row is read once
// (plus an edge-clamped halo that stays in L2) and each output row written once.
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

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float gauss(uint64_t seed, uint64_t idx) {
  const uint64_t h = mix64(seed ^ mix64(idx));
  const float u1 = (static_cast<float>((h >> 40) & 0xFFFFFF) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = static_cast<float>((h >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

constexpr int kMaxBins = 126;
constexpr int kMaxLpc = 63;

}  // namespace

// ------------------------------------------------------------------------------------------------
// RASTA: one thread per (utterance, mel bin), sequential over the utterance's frames.  Log domain,
// float64 IIR state, direct form II transposed exactly like scipy.signal.lfilter, first four frames
// emit exp(0) = 1 and prime the FIR state with zi * x[0] (reference plp.py:87-88, :121-146).
// ------------------------------------------------------------------------------------------------
__global__ void rasta_kernel(float* __restrict__ mel, const BatchArgs b, const int nb) {
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (tid >= b.n_utts * nb) return;
  const int64_t u = tid / nb;
  const int bin = static_cast<int>(tid - u * nb);
  const int64_t f0 = b.frame_offsets[u], f1 = b.frame_offsets[u + 1];
  const double b0 = 0.2, b1 = 0.1, b2 = 0.0, b3 = -0.1, b4 = -0.2, a1 = -0.94;
  double zi3 = b4, zi2 = b3 + zi3, zi1 = b2 + zi2, zi0 = b1 + zi1;
  double z0 = 0, z1 = 0, z2 = 0, z3 = 0;
  float first[4];
  int count = 0;
  for (int64_t t = f0; t < f1; ++t, ++count) {
    float* cell = mel + t * nb + bin;
    const float x = logf(*cell + FLT_EPSILON);
    double y = 0.0;
    if (count < 4) {
      first[count] = x;
      if (count == 3) {
        const double x0 = first[0];
        z0 = zi0 * x0; z1 = zi1 * x0; z2 = zi2 * x0; z3 = zi3 * x0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double xt = first[k];
          z0 = z1 + xt * b1;
          z1 = z2 + xt * b2;
          z2 = z3 + xt * b3;
          z3 = xt * b4;
        }
      }
    } else {
      const double xd = x;
      y = z0 + b0 * xd;
      z0 = (z1 + xd * b1) - y * a1;
      z1 = z2 + xd * b2;
      z2 = z3 + xd * b3;
      z3 = xd * b4;
    }
    *cell = expf(static_cast<float>(y));
  }
}

int launch_rasta(float* mel, const BatchArgs& b, int num_bins, hipStream_t stream) {
  const int64_t total = b.n_utts * num_bins;
  if (total <= 0 || b.total_frames <= 0) return SNF_OK;
  const int threads = 64;
  hipLaunchKernelGGL(rasta_kernel, dim3(static_cast<unsigned>((total + threads - 1) / threads)),
                     dim3(threads), 0, stream, mel, b, num_bins);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

// ------------------------------------------------------------------------------------------------
// PLP tail: equal loudness -> cube-root compression -> IDFT to autocorrelation -> Durbin ->
// LPC to cepstrum -> lifter/scale -> energy -> HTK reorder.  One thread per frame
// (reference plp.py:587-626; Durbin / IDFT bases are [KALDI-UPSTREAM] mel-computations.cc /
// feature-functions.cc wrapped at plp.py:601 and :473).
// ------------------------------------------------------------------------------------------------
__global__ void plp_tail_kernel(const PlpParams p, const BatchArgs b,
                                const float* __restrict__ mel, const double* __restrict__ energy,
                                float* __restrict__ out) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= b.total_frames) return;
  const int nb = p.num_bins, order = p.lpc_order, nc = p.num_ceps;
  int warp_id = 0;
  if (b.utt_warp) warp_id = b.utt_warp[find_utt(b.frame_offsets, b.n_utts, g)];
  const float* __restrict__ eql = p.eql + warp_id * nb;
  float m[kMaxBins + 2];
  for (int i = 0; i < nb; ++i) {
    float v = mel[g * nb + i] * eql[i];
    m[i + 1] = powf(v, p.compress_factor);
  }
  m[0] = m[1];
  m[nb + 1] = m[nb];
  float ac[kMaxLpc + 1], lpc[kMaxLpc], tmp[kMaxLpc], cep[kMaxLpc];
  for (int i = 0; i <= order; ++i) {
    const float* __restrict__ basis = p.idft + i * (nb + 2);
    float s = 0.0f;
    for (int j = 0; j < nb + 2; ++j) s += basis[j] * m[j];
    ac[i] = s;
  }
  // Durbin recursion
  float E = ac[0];
  for (int i = 0; i < order; ++i) lpc[i] = 0.0f;
  for (int i = 0; i < order; ++i) {
    float ki = ac[i + 1];
    for (int j = 0; j < i; ++j) ki += lpc[j] * ac[i - j];
    ki = ki / E;
    float c = 1 - ki * ki;
    if (c < 1.0e-5f) c = 1.0e-5f;
    E *= c;
    tmp[i] = -ki;
    for (int j = 0; j < i; ++j) tmp[j] = lpc[j] - ki * lpc[i - j - 1];
    for (int j = 0; j <= i; ++j) lpc[j] = tmp[j];
  }
  const float res_f = static_cast<float>(-log(1.0 / static_cast<double>(E)));
  const double res = fmax(static_cast<double>(res_f), DBL_EPSILON);
  // LPC -> cepstrum: Python-float (double) accumulation, float32 storage (reference plp.py:149-168)
  for (int i = 0; i < order; ++i) {
    double sum = 0.0;
    for (int j = 0; j < i; ++j)
      sum += static_cast<double>(i - j) * static_cast<double>(lpc[j]) * static_cast<double>(cep[i - j - 1]);
    cep[i] = static_cast<float>(-static_cast<double>(lpc[i]) - sum / static_cast<double>(i + 1));
  }
  float* __restrict__ row = out + g * nc;
  const bool floor_it = p.has_floor;
  for (int c = 0; c < nc; ++c) {
    float v = c == 0 ? static_cast<float>(res) : cep[c - 1];
    if (p.lifter) v *= p.lifter[c];
    if (p.cepstral_scale != 1.0f) v *= p.cepstral_scale;
    if (c == 0 && p.use_energy) {
      // shennong's PLP floors with float64 eps and takes a double log (reference plp.py:191-193); the mel
      // kernels hand over the linear frame energy
      double le = log(fmax(energy[g], DBL_EPSILON));
      if (floor_it && le < p.log_energy_floor) le = p.log_energy_floor;
      v = static_cast<float>(le);
    }
    int oc = c;
    if (p.htk_compat) oc = c == 0 ? nc - 1 : c - 1;
    row[oc] = v;
  }
}

// Same arithmetic, same order, for the usual small shapes (num_bins <= 32, lpc_order <= 16): every
// array has a compile-time bound and stays in registers (the generic kernel above indexes its arrays
// at run time, i.e. from scratch memory), and the 64 mel rows of a workgroup are staged through LDS
// with coalesced loads.
template <int NBMAX, int ORDMAX>
__global__ __launch_bounds__(64) void plp_tail_small_kernel(const PlpParams p, const BatchArgs b,
                                                            const float* __restrict__ mel,
                                                            const double* __restrict__ energy,
                                                            float* __restrict__ out) {
  __shared__ float rows[64 * NBMAX];
  const int nb = p.num_bins, order = p.lpc_order, nc = p.num_ceps;
  const int64_t g0 = static_cast<int64_t>(blockIdx.x) * 64;
  const int64_t limit = b.total_frames * nb;
  for (int i = threadIdx.x; i < 64 * nb; i += 64) {
    const int64_t a = g0 * nb + i;
    rows[i] = a < limit ? mel[a] : 1.0f;
  }
  __syncthreads();
  const int64_t g = g0 + threadIdx.x;
  if (g >= b.total_frames) return;
  int warp_id = 0;
  if (b.utt_warp) warp_id = b.utt_warp[find_utt(b.frame_offsets, b.n_utts, g)];
  const float* __restrict__ eql = p.eql + warp_id * nb;
  float m[NBMAX + 2];
#pragma unroll
  for (int i = 0; i < NBMAX; ++i) {
    m[i + 1] = 0.0f;
    if (i < nb) {
      const float v = rows[threadIdx.x * nb + i] * eql[i];
      m[i + 1] = powf(v, p.compress_factor);
    }
  }
  m[0] = m[1];
  {
    float last = m[1];
#pragma unroll
    for (int i = 1; i <= NBMAX; ++i) if (i == nb) last = m[i];
#pragma unroll
    for (int i = 1; i <= NBMAX + 1; ++i) if (i == nb + 1) m[i] = last;
  }
  float ac[ORDMAX + 1], lpc[ORDMAX], tmp[ORDMAX], cep[ORDMAX];
#pragma unroll
  for (int i = 0; i <= ORDMAX; ++i) {
    ac[i] = 0.0f;
    if (i <= order) {
      const float* __restrict__ basis = p.idft + i * (nb + 2);
      float s = 0.0f;
#pragma unroll
      for (int j = 0; j < NBMAX + 2; ++j)
        if (j < nb + 2) s += basis[j] * m[j];
      ac[i] = s;
    }
  }
  float E = ac[0];
#pragma unroll
  for (int i = 0; i < ORDMAX; ++i) lpc[i] = tmp[i] = cep[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < ORDMAX; ++i) {
    if (i < order) {
      float ki = ac[i + 1];
#pragma unroll
      for (int j = 0; j < ORDMAX; ++j)
        if (j < i) ki += lpc[j] * ac[i - j];
      ki = ki / E;
      float c = 1 - ki * ki;
      if (c < 1.0e-5f) c = 1.0e-5f;
      E *= c;
      tmp[i] = -ki;
#pragma unroll
      for (int j = 0; j < ORDMAX; ++j)
        if (j < i) tmp[j] = lpc[j] - ki * lpc[i - j - 1];
#pragma unroll
      for (int j = 0; j < ORDMAX; ++j)
        if (j <= i) lpc[j] = tmp[j];
    }
  }
  const float res_f = static_cast<float>(-log(1.0 / static_cast<double>(E)));
  const double res = fmax(static_cast<double>(res_f), DBL_EPSILON);
#pragma unroll
  for (int i = 0; i < ORDMAX; ++i) {
    if (i < order) {
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < ORDMAX; ++j)
        if (j < i)
          sum += static_cast<double>(i - j) * static_cast<double>(lpc[j]) *
                 static_cast<double>(cep[i - j - 1]);
      cep[i] = static_cast<float>(-static_cast<double>(lpc[i]) - sum / static_cast<double>(i + 1));
    }
  }
  float* __restrict__ row = out + g * nc;
#pragma unroll
  for (int c = 0; c <= ORDMAX; ++c) {
    if (c < nc) {
      float v = static_cast<float>(res);
#pragma unroll
      for (int k = 1; k <= ORDMAX; ++k) if (k == c) v = cep[k - 1];
      if (p.lifter) v *= p.lifter[c];
      if (p.cepstral_scale != 1.0f) v *= p.cepstral_scale;
      if (c == 0 && p.use_energy) {
        double le = log(fmax(energy[g], DBL_EPSILON));  // (linear frame energy from the mel kernel)
        if (p.has_floor && le < p.log_energy_floor) le = p.log_energy_floor;
        v = static_cast<float>(le);
      }
      int oc = c;
      if (p.htk_compat) oc = c == 0 ? nc - 1 : c - 1;
      row[oc] = v;
    }
  }
}

// x^e for the PLP compression exponent e = float(1/3) (the reference's compress_factor, plp.py:588) and
// x >= 0.  Hardware log2 / exp2 give x^(1/3) to ~1e-6; one Newton step on y^3 = x takes it to float
// round-off; a second one with the residual x - y^3 formed exactly (the roundings of y*y and of (y*y)*y
// recovered with fused multiply-adds) leaves half an ulp; the distance of float(1/3) from 1/3 is the factor
// 1 + (e - 1/3) ln x.  ~28 vector instructions against ~120 for the correctly-rounded powf, within 1-2 ulp
// of it.
__device__ __forceinline__ float pow_third(float x, float e) {
  // (the hardware log2 takes a subnormal for zero: an energy below 1e-30 - not reachable from int16 audio -
  // is scaled by 2^96 on the way in and by 2^(-96 e) on the way out; no branch)
  const bool tiny = x < 1.0e-30f;
  x *= tiny ? 7.9228162514264338e28f : 1.0f;
  const float l2 = __builtin_amdgcn_logf(x);                    // log2 x
  const float y0 = __builtin_amdgcn_exp2f(l2 * 0.33333334f);
  const float r = x * __builtin_amdgcn_rcpf(y0 * y0 * y0);      // x / y0^3 = 1 + O(1e-6)
  const float y1 = y0 * __builtin_fmaf(r, 0.33333334f, 0.66666669f);   // y0 (2 + r) / 3
  const float sq = y1 * y1, sq_err = __builtin_fmaf(y1, y1, -sq);
  const float cu = sq * y1, cu_err = __builtin_fmaf(sq, y1, -cu) + sq_err * y1;  // y1^3 = cu + cu_err
  const float res = (x - cu) - cu_err;
  float y = __builtin_fmaf(res, __builtin_amdgcn_rcpf(3.0f * sq), y1);
  const float d = static_cast<float>(static_cast<double>(e) - 1.0 / 3.0);
  y = __builtin_fmaf(y * d, l2 * 0.69314718f, y);
  y *= tiny ? __builtin_amdgcn_exp2f(-96.0f * e) : 1.0f;
  return x > 0.0f ? y : 0.0f;
}

// The same arithmetic in the same order once more, for ONE exact shape (the reference's defaults: 23 mel
// bins, LPC order 12, 13 cepstra; round 4).  plp_tail_small_kernel<32, 16> predicates every loop of the
// 32 x 16 bound on the run-time sizes: 1.8 x the multiply-adds of the 25 x 13 IDFT, 1.8 x those of the
// Durbin and cepstrum recursions, ~1 200 scalar branches and ~580 single-dword scalar loads with their waits
// (tools/count_isa.py).  Here every bound is a template argument, the IDFT bases, the equal-loudness curve
// and the lifter are staged in LDS once per 256-frame workgroup and read as 16-byte broadcasts, and the mel
// rows arrive through the same coalesced LDS staging.
template <int NB, int ORD, int NC>
__global__ __launch_bounds__(256) void plp_tail_exact_kernel(const PlpParams p, const BatchArgs b,
                                                             const float* __restrict__ mel,
                                                             const double* __restrict__ energy,
                                                             float* __restrict__ out) {
  // Round 5: the tail was bound by the LDS pipe, not by its arithmetic (0.205 ms with a seventh of the
  // instructions removed).  Rows 24 floats apart put the 64 lanes of a row read on 4-8 banks (24 t mod 32), and
  // the 325 IDFT bases came as 86 broadcast ds_read_b128 per wave: ~1 800 LDS clocks per wave, 0.14 ms per CU.
  // Now the row pitch is the odd row length itself (23: lane t reads bank 23 t + i, all different; the staging
  // copy is linear) and the bases and the lifter are read where they are used through the scalar cache
  // (uniform addresses with compile-time offsets: s_load_dwordx8/16, an SGPR operand per multiply-add).
  constexpr int kRowPad = NB;
  static_assert(NB % 2 == 1 && NC <= NB, "odd row pitch, output row inside the mel row's slot");
  __shared__ float rows[256 * kRowPad];
  const float* __restrict__ basis = p.idft;
  const int64_t g0 = static_cast<int64_t>(blockIdx.x) * 256;
  const int64_t limit = b.total_frames * NB;
  for (int i = threadIdx.x; i < 256 * NB; i += 256) {
    const int64_t a = g0 * NB + i;
    rows[i] = a < limit ? mel[a] : 1.0f;
  }
  __syncthreads();
  // (no early exit: the rows leave through LDS behind a barrier, see the end; a thread past the last frame
  // works on the row of ones the staging loop made and stores nothing)
  const bool live = g0 + threadIdx.x < b.total_frames;
  const int64_t g = live ? g0 + threadIdx.x : b.total_frames - 1;
  int warp_id = 0;
  if (b.utt_warp) warp_id = b.utt_warp[find_utt(b.frame_offsets, b.n_utts, g)];
  const float* __restrict__ eql = p.eql + warp_id * NB;
  float m[NB + 2];
  if (p.compress_factor == 0.33333334f && !p.exact_pow) {
#pragma unroll
    for (int i = 0; i < NB; ++i) m[i + 1] = pow_third(rows[threadIdx.x * kRowPad + i] * eql[i], p.compress_factor);
  } else {
#pragma unroll 1
    for (int i = 0; i < NB; ++i) m[i + 1] = powf(rows[threadIdx.x * kRowPad + i] * eql[i], p.compress_factor);
  }
  m[0] = m[1];
  m[NB + 1] = m[NB];
  float ac[ORD + 1], lpc[ORD], tmp[ORD], cep[ORD];
#pragma unroll
  for (int i = 0; i <= ORD; ++i) {
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < NB + 2; ++j) s += basis[i * (NB + 2) + j] * m[j];
    ac[i] = s;
  }
  float E = ac[0];
#pragma unroll
  for (int i = 0; i < ORD; ++i) lpc[i] = tmp[i] = cep[i] = 0.0f;
#pragma unroll
  for (int i = 0; i < ORD; ++i) {
    float ki = ac[i + 1];
#pragma unroll
    for (int j = 0; j < i; ++j) ki += lpc[j] * ac[i - j];
    ki = ki / E;
    float c = 1 - ki * ki;
    if (c < 1.0e-5f) c = 1.0e-5f;
    E *= c;
    tmp[i] = -ki;
#pragma unroll
    for (int j = 0; j < i; ++j) tmp[j] = lpc[j] - ki * lpc[i - j - 1];
#pragma unroll
    for (int j = 0; j <= i; ++j) lpc[j] = tmp[j];
  }
  // (round 5) The reference forms -log(1 / E) and the frame's log-energy in float64 and rounds them to float32
  // (plp.py:601-603, :615-620): a float32 logarithm that is good to an ulp gives the same float32 up to its last
  // bit (1e-7 relative on values of 5-20, the parity tolerance is 1e-4) at a fifth of the instructions of the
  // double one.  The LPC -> cepstrum recursion keeps the reference's double accumulation (plp.py:149-168) with
  // the weights (k + 1) c[k] made once per cepstrum and one fused multiply-add per term: 78 double operations
  // instead of 198 + 12 divisions, the same sums up to the last bit of a double that is rounded to float next.
  const float res_f = logf(E);
  const double res = fmax(static_cast<double>(res_f), DBL_EPSILON);
  double wcep[ORD];
#pragma unroll
  for (int i = 0; i < ORD; ++i) {
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < i; ++j) sum = fma(static_cast<double>(lpc[j]), wcep[i - j - 1], sum);
    cep[i] = static_cast<float>(-static_cast<double>(lpc[i]) - sum * (1.0 / static_cast<double>(i + 1)));
    wcep[i] = static_cast<double>(i + 1) * static_cast<double>(cep[i]);
  }
  // Round 5: the 13 values of a frame used to leave as 13 dword stores per lane, 52 bytes apart from lane to lane
  // - every store instruction touched 52 cache lines, and the kernel took 0.26 ms whatever arithmetic was left in
  // it (ablations without the cube roots, the divisions, the double recursion or the IDFT: 0.24-0.27 ms).  The
  // thread now puts its row where its mel row was (its own slot: nobody else reads it), and the workgroup
  // writes its 256 x 13 floats - one contiguous block of the output - with coalesced stores: 0.21 ms.
  // (Also built: persistent workgroups with the next tile's mel rows requested into registers a tile ahead -
  // 180 registers, 2 waves per SIMD, the same 0.21 ms; capped at 128 / 96 / 80 registers it spills: 0.30-0.33.)
  float* __restrict__ slot = rows + threadIdx.x * kRowPad;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    float v = c == 0 ? static_cast<float>(res) : cep[c > 0 ? c - 1 : 0];
    if (p.lifter) v *= p.lifter[c];
    if (p.cepstral_scale != 1.0f) v *= p.cepstral_scale;
    if (c == 0 && p.use_energy) {
      // (linear frame energy from the mel kernel: a float sum widened to double; the floor of the reference's
      // double logarithm, DBL_EPSILON, is a float32 number too)
      float le = logf(fmaxf(static_cast<float>(energy[g]), 2.220446049250313e-16f));
      if (p.has_floor && le < p.log_energy_floor) le = p.log_energy_floor;
      v = le;
    }
    int oc = c;
    if (p.htk_compat) oc = c == 0 ? NC - 1 : c - 1;
    slot[oc] = v;
  }
  __syncthreads();
  const int64_t left = b.total_frames - g0;
  const int count = static_cast<int>(left < 256 ? left : 256) * NC;
  float* __restrict__ block = out + g0 * NC;
  for (int i = threadIdx.x; i < count; i += 256) block[i] = rows[(i / NC) * kRowPad + i % NC];
}

int launch_plp_tail(const PlpParams& p, const BatchArgs& b, const float* mel, const double* energy,
                    float* out, hipStream_t stream) {
  if (b.total_frames <= 0) return SNF_OK;
  if (p.num_bins > kMaxBins || p.lpc_order > kMaxLpc)
    return set_error(SNF_E_RUNTIME, "PLP: num_bins > 126 or lpc_order > 63 not supported");
  const int threads = 64;
  if (p.num_bins == 23 && p.lpc_order == 12 && p.num_ceps == 13 && !getenv("SNF_PLP_GENERIC_TAIL") &&
      !getenv("SNF_PLP_SMALL_TAIL")) {
    hipLaunchKernelGGL((plp_tail_exact_kernel<23, 12, 13>),
                       dim3(static_cast<unsigned>((b.total_frames + 255) / 256)), dim3(256), 0, stream, p, b, mel,
                       energy, out);
    SNF_HIP_CHECK(hipGetLastError());
    return SNF_OK;
  }
  if (p.num_bins <= 32 && p.lpc_order <= 16 && !getenv("SNF_PLP_GENERIC_TAIL")) {
    hipLaunchKernelGGL((plp_tail_small_kernel<32, 16>),
                       dim3(static_cast<unsigned>((b.total_frames + threads - 1) / threads)),
                       dim3(threads), 0, stream, p, b, mel, energy, out);
    SNF_HIP_CHECK(hipGetLastError());
    return SNF_OK;
  }
  hipLaunchKernelGGL(plp_tail_kernel,
                     dim3(static_cast<unsigned>((b.total_frames + threads - 1) / threads)),
                     dim3(threads), 0, stream, p, b, mel, energy, out);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

// ------------------------------------------------------------------------------------------------
// Deltas: out[:, i*D:(i+1)*D] = sum_j scales_i[j] * in[clamp(t + j)]; one thread per (row, column).
// ------------------------------------------------------------------------------------------------
__global__ void delta_kernel(const DeltaParams p, const float* __restrict__ in, const int D,
                             const int64_t* __restrict__ frame_offsets, const int64_t n_utts,
                             const int64_t total_frames, float* __restrict__ out) {
  const int64_t idx = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= total_frames * D) return;
  const int64_t g = idx / D;
  const int c = static_cast<int>(idx - g * D);
  const int64_t u = find_utt(frame_offsets, n_utts, g);
  const int64_t f0 = frame_offsets[u], f1 = frame_offsets[u + 1];
  const int OD = D * (p.order + 1);
  const float* __restrict__ sc = p.scales;
  for (int i = 0; i <= p.order; ++i) {
    const int dim = p.dims[i], max_off = (dim - 1) / 2;
    float acc = 0.0f;
    for (int j = -max_off; j <= max_off; ++j) {
      int64_t t = g + j;
      t = t < f0 ? f0 : (t >= f1 ? f1 - 1 : t);
      const float s = sc[j + max_off];
      if (s != 0.0f) acc += s * in[t * D + c];
    }
    out[g * OD + static_cast<int64_t>(i) * D + c] = acc;
    sc += dim;
  }
}

// Tiled variant: a workgroup stages kDeltaRows + 2*halo input rows in LDS once (each HBM row is read
// ~1.1x instead of 9x through the caches) and writes its output rows fully coalesced.  The edge clamp
// is per utterance; a clamped neighbour is never farther than the unclamped one, so it is in the tile.
constexpr int kDeltaRows = 256;

__global__ __launch_bounds__(256) void delta_tiled_kernel(
    const DeltaParams p, const float* __restrict__ in, const int D, const int halo,
    const int64_t* __restrict__ frame_offsets, const int64_t n_utts, const int64_t total_frames,
    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* row_lo = reinterpret_cast<int*>(smem);   // [kDeltaRows] first row of the utterance (tile-relative)
  int* row_hi = row_lo + kDeltaRows;            // [kDeltaRows] last row of the utterance (tile-relative)
  float* scales = reinterpret_cast<float*>(row_hi + kDeltaRows);  // [n_scales]
  float* tile = scales + ((p.n_scales + 3) & ~3);                 // [(kDeltaRows + 2 halo), D]
  const int64_t g0 = static_cast<int64_t>(blockIdx.x) * kDeltaRows;
  const int64_t t0 = g0 - halo;
  const int tile_rows = kDeltaRows + 2 * halo;
  const int64_t first = t0 * D, limit = total_frames * D;
  for (int i = threadIdx.x; i < tile_rows * D; i += blockDim.x) {
    const int64_t a = first + i;
    tile[i] = (a >= 0 && a < limit) ? in[a] : 0.0f;
  }
  for (int i = threadIdx.x; i < p.n_scales; i += blockDim.x) scales[i] = p.scales[i];
  if (threadIdx.x < kDeltaRows) {
    const int64_t g = g0 + threadIdx.x;
    if (g < total_frames) {
      const int64_t u = find_utt(frame_offsets, n_utts, g);
      const int64_t lo = frame_offsets[u] - t0, hi = frame_offsets[u + 1] - 1 - t0;
      row_lo[threadIdx.x] = lo < 0 ? 0 : static_cast<int>(lo);  // rows outside the tile are never
      row_hi[threadIdx.x] = hi > tile_rows - 1 ? tile_rows - 1 : static_cast<int>(hi);  // reached
    }
  }
  __syncthreads();
  const int OD = D * (p.order + 1);
  const int rows_here = static_cast<int>(total_frames - g0 < kDeltaRows ? total_frames - g0 : kDeltaRows);
  // element e = (row r, column c); advanced without divisions
  int r = threadIdx.x / D, c = threadIdx.x - r * D;
  const int dr = blockDim.x / D, dc = blockDim.x - dr * D;
  float* __restrict__ obase = out + g0 * OD;
  for (; r < rows_here; r += dr, c += dc) {
    if (c >= D) { c -= D; ++r; if (r >= rows_here) break; }
    const int lo = row_lo[r], hi = row_hi[r], centre = r + halo;
    const float* __restrict__ sc = scales;
    float* __restrict__ orow = obase + static_cast<int64_t>(r) * OD + c;
    for (int i = 0; i <= p.order; ++i) {
      const int max_off = i * p.window;
      float acc = 0.0f;
      for (int j = -max_off; j <= max_off; ++j) {
        int t = centre + j;
        t = t < lo ? lo : (t > hi ? hi : t);
        const float s = sc[j + max_off];
        if (s != 0.0f) acc += s * tile[t * D + c];
      }
      orow[i * D] = acc;
      sc += 2 * max_off + 1;
    }
  }
}

// The same tile with the order and window known at compile time (the reference's defaults, order 2 /
// window 2, and order 1): the 2*halo+1 clamped neighbours of an element are read from LDS ONCE for all
// the orders, the composite scales sit in scalar registers, the tap loops are unrolled.  Same products
// in the same order as the generic kernel above.
template <int ORDER, int WINDOW>
__global__ __launch_bounds__(256) void delta_tiled_fixed_kernel(
    const DeltaParams p, const float* __restrict__ in, const int D,
    const int64_t* __restrict__ frame_offsets, const int64_t n_utts, const int64_t total_frames,
    float* __restrict__ out) {
  constexpr int kHalo = ORDER * WINDOW;
  constexpr int kTaps = 2 * kHalo + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* row_lo = reinterpret_cast<int*>(smem);
  int* row_hi = row_lo + kDeltaRows;
  float* tile = reinterpret_cast<float*>(row_hi + kDeltaRows);  // [(kDeltaRows + 2 kHalo), D]
  const int64_t g0 = static_cast<int64_t>(blockIdx.x) * kDeltaRows;
  const int64_t t0 = g0 - kHalo;
  constexpr int tile_rows = kDeltaRows + 2 * kHalo;
  const int64_t first = t0 * D, limit = total_frames * D;
  for (int i = threadIdx.x; i < tile_rows * D; i += blockDim.x) {
    const int64_t a = first + i;
    tile[i] = (a >= 0 && a < limit) ? in[a] : 0.0f;
  }
  if (threadIdx.x < kDeltaRows) {
    const int64_t g = g0 + threadIdx.x;
    if (g < total_frames) {
      const int64_t u = find_utt(frame_offsets, n_utts, g);
      const int64_t lo = frame_offsets[u] - t0, hi = frame_offsets[u + 1] - 1 - t0;
      row_lo[threadIdx.x] = lo < 0 ? 0 : static_cast<int>(lo);
      row_hi[threadIdx.x] = hi > tile_rows - 1 ? tile_rows - 1 : static_cast<int>(hi);
    }
  }
  // composite scales of every order, concatenated like DeltaParams::scales (uniform: scalar loads)
  constexpr int kScales = (ORDER + 1) * (ORDER * WINDOW + 1);  // sum of 2 i WINDOW + 1, i <= ORDER
  float sc[kScales];
#pragma unroll
  for (int i = 0; i < kScales; ++i) sc[i] = p.scales[i];
  __syncthreads();
  const int OD = D * (ORDER + 1);
  const int rows_here = static_cast<int>(total_frames - g0 < kDeltaRows ? total_frames - g0 : kDeltaRows);
  int r = threadIdx.x / D, c = threadIdx.x - r * D;
  const int dr = blockDim.x / D, dc = blockDim.x - dr * D;
  float* __restrict__ obase = out + g0 * OD;
  for (; r < rows_here; r += dr, c += dc) {
    if (c >= D) { c -= D; ++r; if (r >= rows_here) break; }
    const int lo = row_lo[r], hi = row_hi[r], centre = r + kHalo;
    float x[kTaps];
#pragma unroll
    for (int j = 0; j < kTaps; ++j) {
      int t = centre + j - kHalo;
      t = t < lo ? lo : (t > hi ? hi : t);
      x[j] = tile[t * D + c];
    }
    float* __restrict__ orow = obase + static_cast<int64_t>(r) * OD + c;
    int soff = 0;
#pragma unroll
    for (int i = 0; i <= ORDER; ++i) {
      const int max_off = i * WINDOW;
      float acc = 0.0f;
#pragma unroll
      for (int j = -max_off; j <= max_off; ++j) {
        const float s = sc[soff + j + max_off];
        if (s != 0.0f) acc += s * x[j + kHalo];
      }
      orow[i * D] = acc;
      soff += 2 * max_off + 1;
    }
  }
}

// The reference's defaults (order 2, window 2) as a flat stream: the [T, D] input and the [T, 3 D] output
// are both contiguous.  A tile of rows is staged with 16-byte loads; thread i computes the elements
// (row, column) i, i + 256, ... for all three orders from ONE read of the 9 clamped neighbours
// (consecutive lanes -> consecutive columns: conflict-free; the division by D is by a compile-time
// constant) into an LDS image of the output block, which is then copied out as dwordx4: loads and
// stores are fully coalesced 16-byte accesses (the per-(row, column) form above writes 4-byte elements in runs of D).  The utterance of a
// tile's first row comes from a table built by one thread per tile (a binary search of dependent loads
// at the head of every workgroup costs more than its arithmetic); rows walk forward from it.  Same
// products in the same order as the kernels above.
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int kFlatTileBytes = 28 * 1024;  // input tile + output image of one workgroup
template <int D>
constexpr int flat_rows() {
  // rows per tile: a multiple of 4 (16-byte aligned output blocks) that fits the LDS budget
  int rows = (kFlatTileBytes / 4 - 8 * D - 8) / (4 * D);
  rows &= ~3;
  return rows > 256 ? 256 : rows;
}
// one 32-byte record per tile: the utterance of its first row and the frame offsets of that utterance and
// the next two, so that a workgroup learns the utterance boundaries of its rows from ONE load (a chain of
// dependent loads per row used to be most of a workgroup's life)
__global__ void delta_tile_utt_kernel(const int64_t* __restrict__ frame_offsets, int64_t n_utts,
                                      int64_t total_frames, int rows_per_tile,
                                      int64_t* __restrict__ tile_info) {
  const int64_t tile = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t g = tile * rows_per_tile;
  if (g >= total_frames) return;
  const int64_t u = find_utt(frame_offsets, n_utts, g);
  int64_t* __restrict__ rec = tile_info + 4 * tile;
  rec[0] = u;
  rec[1] = frame_offsets[u];
  rec[2] = frame_offsets[u + 1];
  rec[3] = frame_offsets[u + 2 <= n_utts ? u + 2 : n_utts];
}

template <int D>
__global__ __launch_bounds__(256) void delta_flat_o2w2_kernel(
    const DeltaParams p, const float* __restrict__ in, const int64_t* __restrict__ frame_offsets,
    const int64_t* __restrict__ tile_info, const int64_t n_utts, const int64_t total_frames,
    float* __restrict__ out) {
  constexpr int kRows = flat_rows<D>();
  constexpr int kHalo = 4, OD = 3 * D, kTileRows = kRows + 2 * kHalo;
  constexpr int kTileFloats = (kTileRows * D + 3) & ~3;
  __shared__ __attribute__((aligned(16))) float tile[kTileFloats + 4];
  __shared__ __attribute__((aligned(16))) float image[kRows * OD];
  __shared__ int row_lo[kRows], row_hi[kRows];
  const int64_t g0 = static_cast<int64_t>(blockIdx.x) * kRows;
  const int64_t t0 = g0 - kHalo;
  const int64_t first = t0 * D, limit = total_frames * D;
  // the tile starts at float `first` of the input, which is 16-byte aligned only for some tiles: stage
  // from the aligned float4 at or below it (`skew` floats earlier)
  const int skew = static_cast<int>(((first % 4) + 4) % 4);
  const int64_t base = first - skew;
  // the tile is requested first (registers), the utterance bookkeeping below overlaps its latency
  constexpr int kStage = (kTileFloats / 4 + 1 + 255) / 256;
  float4 staged[kStage];
#pragma unroll
  for (int k = 0; k < kStage; ++k) {
    const int i = threadIdx.x + 256 * k;
    const int64_t a = base + 4 * static_cast<int64_t>(i);
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (i * 4 < kTileFloats + 4) {
      if (a >= 0 && a + 3 < limit) {
        v = *reinterpret_cast<const float4*>(in + a);
      } else {
        if (a >= 0 && a < limit) v.x = in[a];
        if (a + 1 >= 0 && a + 1 < limit) v.y = in[a + 1];
        if (a + 2 >= 0 && a + 2 < limit) v.z = in[a + 2];
        if (a + 3 >= 0 && a + 3 < limit) v.w = in[a + 3];
      }
    }
    staged[k] = v;
  }
  const int rows_here = static_cast<int>(total_frames - g0 < kRows ? total_frames - g0 : kRows);
  bool interior = rows_here == kRows;
  {
    const int64_t* __restrict__ rec = tile_info + 4 * static_cast<int64_t>(blockIdx.x);
    const int64_t u0 = rec[0], o0 = rec[1], o1 = rec[2], o2 = rec[3];
    for (int r = threadIdx.x; r < kRows; r += blockDim.x) {
      const int64_t g = g0 + r;
      if (g < total_frames) {
        int64_t lo_f = o0, hi_f = o1;          // [first, end) frames of the row's utterance
        if (g >= o1) {
          lo_f = o1;
          hi_f = o2;
          if (g >= o2) {                       // more than two boundaries inside the tile (rare)
            int64_t u = u0 + 2;
            while (frame_offsets[u + 1] <= g) ++u;
            lo_f = frame_offsets[u];
            hi_f = frame_offsets[u + 1];
          }
        }
        const int64_t lo = lo_f - t0, hi = hi_f - 1 - t0;
        row_lo[r] = lo < 0 ? 0 : static_cast<int>(lo);
        row_hi[r] = hi > kTileRows - 1 ? kTileRows - 1 : static_cast<int>(hi);
        interior = interior && lo <= r && hi >= r + 2 * kHalo;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kStage; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i * 4 < kTileFloats + 4) reinterpret_cast<float4*>(tile)[i] = staged[k];
  }
  float sc[15];  // scales of the three orders, concatenated like DeltaParams::scales: 1 + 5 + 9
#pragma unroll
  for (int i = 0; i < 15; ++i) sc[i] = p.scales[i];
  // interior tile: every row and its +-4 neighbours lie inside one utterance (3 tiles out of 4 on 3 s
  // utterances) -> no clamping, and a thread walks down one column of a strip of rows with the 9-row
  // window in registers: one LDS read per element instead of nine
  interior = __syncthreads_and(interior);
  const int n_out = rows_here * OD;
  const float* __restrict__ tl = tile + skew;
  if (interior) {
    constexpr int kStrips = 256 / D, kStripRows = (kRows + kStrips - 1) / kStrips;
    const int c = threadIdx.x % D, strip = threadIdx.x / D;
    if (strip < kStrips) {
      const int r0 = strip * kStripRows;
      float v[kStripRows + 2 * kHalo];
#pragma unroll
      for (int i = 0; i < kStripRows + 2 * kHalo; ++i)
        v[i] = r0 + i < kTileRows ? tl[(r0 + i) * D + c] : 0.0f;
#pragma unroll
      for (int rr = 0; rr < kStripRows; ++rr) {
        if (r0 + rr < kRows) {
          float* __restrict__ orow = image + (r0 + rr) * OD + c;
          int soff = 0;
#pragma unroll
          for (int i = 0; i <= 2; ++i) {
            const int max_off = 2 * i;
            float acc = 0.0f;
#pragma unroll
            for (int j = -max_off; j <= max_off; ++j) {
              const float w = sc[soff + j + max_off];
              if (w != 0.0f) acc += w * v[rr + j + kHalo];
            }
            orow[i * D] = acc;
            soff += 2 * max_off + 1;
          }
        }
      }
    }
  }
  // element (row r, column c): its 9 clamped neighbours are read ONCE for the three orders (every lane
  // runs the same taps: no divergence), the three results go to the output image
  for (int idx = threadIdx.x; !interior && idx < rows_here * D; idx += blockDim.x) {
    const int r = idx / D, c = idx - r * D;
    const int lo = row_lo[r], hi = row_hi[r], centre = r + kHalo;
    float x[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      int t = centre + j - kHalo;
      t = t < lo ? lo : (t > hi ? hi : t);
      x[j] = tl[t * D + c];
    }
    float* __restrict__ orow = image + r * OD + c;
    int soff = 0;
#pragma unroll
    for (int i = 0; i <= 2; ++i) {
      const int max_off = 2 * i;
      float acc = 0.0f;
#pragma unroll
      for (int j = -max_off; j <= max_off; ++j) {
        const float w = sc[soff + j + max_off];
        if (w != 0.0f) acc += w * x[j + kHalo];
      }
      orow[i * D] = acc;
      soff += 2 * max_off + 1;
    }
  }
  __syncthreads();
  float* __restrict__ obase = out + g0 * OD;   // (g0 * OD * 4 bytes is a multiple of 16: kRows % 4 == 0)
  for (int e0 = 4 * threadIdx.x; e0 < n_out; e0 += 4 * blockDim.x) {
    if (e0 + 3 < n_out) {
      __builtin_nontemporal_store(*reinterpret_cast<const f4v*>(image + e0), reinterpret_cast<f4v*>(obase + e0));
    } else {
      for (int k = 0; k < 4 && e0 + k < n_out; ++k) obase[e0 + k] = image[e0 + k];
    }
  }
}

int launch_deltas(const DeltaParams& p, const float* in, int in_cols, const int64_t* frame_offsets,
                  int64_t n_utts, int64_t total_frames, float* out, int64_t* tile_info, bool build_info,
                  hipStream_t stream) {
  const int64_t total = total_frames * in_cols;
  if (total <= 0) return SNF_OK;
  const int halo = p.order * p.window;
  const bool aligned16 = (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (p.order == 2 && p.window == 2 && tile_info &&
      (in_cols == 13 || in_cols == 23 || in_cols == 40 || in_cols == 43)) {
    // The tile records are rebuilt whenever the caller asks (`build_info`: its offsets table changed), on
    // whichever path this call then takes: the caller remembers the table, not the path, and a later call
    // with 16-byte aligned pointers must not find records of an older table.
#define SNF_FLAT(D_)                                                                                  \
  do {                                                                                                \
    constexpr int rows = flat_rows<D_>();                                                             \
    const unsigned tiles = static_cast<unsigned>((total_frames + rows - 1) / rows);                   \
    if (build_info)                                                                                   \
      hipLaunchKernelGGL(delta_tile_utt_kernel, dim3((tiles + 255) / 256), dim3(256), 0, stream,      \
                         frame_offsets, n_utts, total_frames, rows, tile_info);                       \
    if (aligned16)                                                                                    \
      hipLaunchKernelGGL((delta_flat_o2w2_kernel<D_>), dim3(tiles), dim3(256), 0, stream, p, in,      \
                         frame_offsets, tile_info, n_utts, total_frames, out);                        \
  } while (0)
    if (in_cols == 13) SNF_FLAT(13);
    else if (in_cols == 23) SNF_FLAT(23);
    else if (in_cols == 40) SNF_FLAT(40);
    else SNF_FLAT(43);
#undef SNF_FLAT
    SNF_HIP_CHECK(hipGetLastError());
    if (aligned16) return SNF_OK;
  }
  const size_t lds = 2 * sizeof(int) * kDeltaRows + sizeof(float) * ((p.n_scales + 3) & ~3) +
                     sizeof(float) * static_cast<size_t>(kDeltaRows + 2 * halo) * in_cols;
  const unsigned tiles = static_cast<unsigned>((total_frames + kDeltaRows - 1) / kDeltaRows);
  if (lds <= 48 * 1024 && in_cols <= 256 && p.window == 2 && (p.order == 2 || p.order == 1)) {
    const size_t lds_fixed = 2 * sizeof(int) * kDeltaRows +
                             sizeof(float) * static_cast<size_t>(kDeltaRows + 2 * halo) * in_cols;
    if (p.order == 2)
      hipLaunchKernelGGL((delta_tiled_fixed_kernel<2, 2>), dim3(tiles), dim3(256), lds_fixed, stream, p,
                         in, in_cols, frame_offsets, n_utts, total_frames, out);
    else
      hipLaunchKernelGGL((delta_tiled_fixed_kernel<1, 2>), dim3(tiles), dim3(256), lds_fixed, stream, p,
                         in, in_cols, frame_offsets, n_utts, total_frames, out);
    SNF_HIP_CHECK(hipGetLastError());
    return SNF_OK;
  }
  if (lds <= 48 * 1024 && in_cols <= 256) {
    hipLaunchKernelGGL(delta_tiled_kernel,
                       dim3(static_cast<unsigned>((total_frames + kDeltaRows - 1) / kDeltaRows)),
                       dim3(256), lds, stream, p, in, in_cols, halo, frame_offsets, n_utts,
                       total_frames, out);
    SNF_HIP_CHECK(hipGetLastError());
    return SNF_OK;
  }
  const int threads = 256;
  hipLaunchKernelGGL(delta_kernel, dim3(static_cast<unsigned>((total + threads - 1) / threads)),
                     dim3(threads), 0, stream, p, in, in_cols, frame_offsets, n_utts, total_frames,
                     out);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

// ------------------------------------------------------------------------------------------------
// Pitch post-processing: one thread per frame.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float nccf_to_pov_feature(float n) {
  n = fminf(fmaxf(n, -1.0f), 1.0f);
  return static_cast<float>(pow(1.0001 - static_cast<double>(n), 0.15) - 1.0);
}
__device__ __forceinline__ float nccf_to_pov(float n) {
  float a = fabsf(n);
  if (a > 1.0f) a = 1.0f;
  const double ad = a;
  // Kaldi calls Exp() with double arguments here -> double overload
  const float r = static_cast<float>(-5.2 + 5.4 * exp(7.5 * (ad - 1.0)) + 4.8 * ad -
                                     2.0 * exp(-10.0 * ad) + 4.2 * exp(20.0 * (ad - 1.0)));
  return static_cast<float>(1.0 / (1.0 + exp(-1.0 * static_cast<double>(r))));
}

__global__ void pitch_post_kernel(const PitchPostParams p, const float* __restrict__ in,
                                  const int64_t* __restrict__ frame_offsets, const int64_t n_utts,
                                  const int64_t total_frames, float* __restrict__ out) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= total_frames) return;
  const int64_t u = find_utt(frame_offsets, n_utts, g);
  const int64_t f0 = frame_offsets[u], f1 = frame_offsets[u + 1];
  const snf_pitch_post_options& o = p.o;
  float* __restrict__ row = out + g * p.ndims;
  int idx = 0;
  const float nccf = in[g * 2], log_pitch = logf(in[g * 2 + 1]);
  if (o.add_pov_feature) row[idx++] = o.pov_scale * nccf_to_pov_feature(nccf) + o.pov_offset;
  if (o.add_normalized_log_pitch) {
    int64_t wb = g - o.normalization_left_context, we = g + o.normalization_right_context + 1;
    if (wb < f0) wb = f0;
    if (we > f1) we = f1;
    double sum_pov = 0.0, sum_lp = 0.0;
    for (int64_t t = wb; t < we; ++t) {
      const float pov = nccf_to_pov(in[t * 2]), lp = logf(in[t * 2 + 1]);
      sum_pov += pov;
      sum_lp += pov * lp;
    }
    const float avg = static_cast<float>(sum_lp / sum_pov);
    row[idx++] = (log_pitch - avg) * o.pitch_scale;
  }
  if (o.add_delta_pitch) {
    // ComputeDeltas(order 1, window w) over the clamped +-w neighbourhood of log-pitch
    const int w = o.delta_window;
    float normalizer = 0.0f;
    for (int j = -w; j <= w; ++j) normalizer += static_cast<float>(j * j);
    const float inv = static_cast<float>(1.0 / normalizer);
    int64_t lo = g - w, hi = g + w;
    if (lo < f0) lo = f0;
    if (hi > f1 - 1) hi = f1 - 1;
    float acc = 0.0f;
    for (int j = -w; j <= w; ++j) {
      int64_t t = g + j;
      t = t < lo ? lo : (t > hi ? hi : t);
      const float s = static_cast<float>(j) * inv;
      if (s != 0.0f) acc += s * logf(in[t * 2 + 1]);
    }
    float noise = 0.0f;
    if (o.delta_pitch_noise_stddev != 0.0f)
      noise = gauss(p.seed, frame_noise_id(g - f0, f1 - f0, __float_as_uint(in[f0 * 2 + 1]))) *
              o.delta_pitch_noise_stddev;  // (keyed per utterance, not per batch row: snf_internal.h)
    row[idx++] = (acc + noise) * o.delta_pitch_scale;
  }
  if (o.add_raw_log_pitch) row[idx++] = log_pitch;
}

// Tiled variant of the same arithmetic: the POV weight and the log-pitch of a frame are evaluated
// ONCE (by the workgroup that needs them, into LDS) instead of once per window position - the POV
// mapping costs four double-precision exponentials, and the +-75-frame normalisation window made the
// kernel above evaluate it 151 times per frame.
constexpr int kPostRows = 256;
__global__ __launch_bounds__(kPostRows) void pitch_post_tiled_kernel(
    const PitchPostParams p, const float* __restrict__ in, const int halo_l, const int halo_r,
    const int64_t* __restrict__ frame_offsets, const int64_t n_utts, const int64_t total_frames,
    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_pov = reinterpret_cast<float*>(smem);        // [kPostRows + halo_l + halo_r]
  float* s_lp = s_pov + (kPostRows + halo_l + halo_r);  // log-pitch
  const int64_t g0 = static_cast<int64_t>(blockIdx.x) * kPostRows;
  const int64_t t0 = g0 - halo_l;
  const int tile = kPostRows + halo_l + halo_r;
  for (int i = threadIdx.x; i < tile; i += blockDim.x) {
    const int64_t t = t0 + i;
    float pov = 0.0f, lp = 0.0f;
    if (t >= 0 && t < total_frames) {
      pov = nccf_to_pov(in[t * 2]);
      lp = logf(in[t * 2 + 1]);
    }
    s_pov[i] = pov;
    s_lp[i] = lp;
  }
  __syncthreads();
  const int64_t g = g0 + threadIdx.x;
  if (g >= total_frames) return;
  const int64_t u = find_utt(frame_offsets, n_utts, g);
  const int64_t f0 = frame_offsets[u], f1 = frame_offsets[u + 1];
  const snf_pitch_post_options& o = p.o;
  float* __restrict__ row = out + g * p.ndims;
  int idx = 0;
  const float nccf = in[g * 2];
  const float log_pitch = s_lp[g - t0];
  if (o.add_pov_feature) row[idx++] = o.pov_scale * nccf_to_pov_feature(nccf) + o.pov_offset;
  if (o.add_normalized_log_pitch) {
    int64_t wb = g - o.normalization_left_context, we = g + o.normalization_right_context + 1;
    if (wb < f0) wb = f0;
    if (we > f1) we = f1;
    double sum_pov = 0.0, sum_lp = 0.0;
    for (int64_t t = wb; t < we; ++t) {
      const float pov = s_pov[t - t0], lp = s_lp[t - t0];
      sum_pov += pov;
      sum_lp += pov * lp;
    }
    const float avg = static_cast<float>(sum_lp / sum_pov);
    row[idx++] = (log_pitch - avg) * o.pitch_scale;
  }
  if (o.add_delta_pitch) {
    const int w = o.delta_window;
    float normalizer = 0.0f;
    for (int j = -w; j <= w; ++j) normalizer += static_cast<float>(j * j);
    const float inv = static_cast<float>(1.0 / normalizer);
    int64_t lo = g - w, hi = g + w;
    if (lo < f0) lo = f0;
    if (hi > f1 - 1) hi = f1 - 1;
    float acc = 0.0f;
    for (int j = -w; j <= w; ++j) {
      int64_t t = g + j;
      t = t < lo ? lo : (t > hi ? hi : t);
      const float sc = static_cast<float>(j) * inv;
      if (sc != 0.0f) acc += sc * s_lp[t - t0];
    }
    float noise = 0.0f;
    if (o.delta_pitch_noise_stddev != 0.0f)
      noise = gauss(p.seed, frame_noise_id(g - f0, f1 - f0, __float_as_uint(in[f0 * 2 + 1]))) *
              o.delta_pitch_noise_stddev;  // (keyed per utterance, not per batch row: snf_internal.h)
    row[idx++] = (acc + noise) * o.delta_pitch_scale;
  }
  if (o.add_raw_log_pitch) row[idx++] = log_pitch;
}

int launch_pitch_post(const PitchPostParams& p, const float* in, const int64_t* frame_offsets,
                      int64_t n_utts, int64_t total_frames, float* out, hipStream_t stream) {
  if (total_frames <= 0) return SNF_OK;
  {
    int halo_l = p.o.normalization_left_context, halo_r = p.o.normalization_right_context;
    if (p.o.delta_window > halo_l) halo_l = p.o.delta_window;
    if (p.o.delta_window > halo_r) halo_r = p.o.delta_window;
    if (halo_l < 0) halo_l = 0;
    if (halo_r < 0) halo_r = 0;
    const size_t lds = sizeof(float) * 2 * static_cast<size_t>(kPostRows + halo_l + halo_r);
    if (lds <= 48 * 1024) {
      hipLaunchKernelGGL(pitch_post_tiled_kernel,
                         dim3(static_cast<unsigned>((total_frames + kPostRows - 1) / kPostRows)),
                         dim3(kPostRows), lds, stream, p, in, halo_l, halo_r, frame_offsets, n_utts,
                         total_frames, out);
      SNF_HIP_CHECK(hipGetLastError());
      return SNF_OK;
    }
  }
  const int threads = 128;
  hipLaunchKernelGGL(pitch_post_kernel,
                     dim3(static_cast<unsigned>((total_frames + threads - 1) / threads)),
                     dim3(threads), 0, stream, p, in, frame_offsets, n_utts, total_frames, out);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

// ------------------------------------------------------------------------------------------------
// energy VAD, CMVN, sliding-window CMVN
// ------------------------------------------------------------------------------------------------
// VAD ([KALDI-UPSTREAM] ivector/voice-activity-detection.cc ComputeVadEnergy; reference
// postprocessor/vad.py:182-185): per-utterance threshold from the mean of column 0, then the
// proportion test over a +-context window, one thread per frame.
__global__ void vad_threshold_kernel(const snf_vad_options o, const float* __restrict__ in,
                                     const int D, const int64_t* __restrict__ frame_offsets,
                                     float* __restrict__ thr) {
  const int64_t u = blockIdx.x;
  const int64_t f0 = frame_offsets[u], T = frame_offsets[u + 1] - f0;
  double s = 0.0;
  for (int64_t t = threadIdx.x; t < T; t += blockDim.x) s += in[(f0 + t) * D];
  __shared__ double red[16];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0;
    for (unsigned i = 0; i < (blockDim.x >> 6); ++i) tot += red[i];
    float v = o.energy_threshold;
    if (o.energy_mean_scale != 0.0f && T > 0)
      v += o.energy_mean_scale * static_cast<float>(tot) / static_cast<float>(T);
    thr[u] = v;
  }
}

__global__ void vad_kernel(const snf_vad_options o, const float* __restrict__ in, const int D,
                           const int64_t* __restrict__ frame_offsets, const int64_t n_utts,
                           const int64_t total_frames, const float* __restrict__ thr,
                           float* __restrict__ out) {
  const int64_t g = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (g >= total_frames) return;
  const int64_t u = find_utt(frame_offsets, n_utts, g);
  const int64_t f0 = frame_offsets[u], f1 = frame_offsets[u + 1];
  const float th = thr[u];
  int num = 0, den = 0;
  for (int64_t t = g - o.frames_context; t <= g + o.frames_context; ++t)
    if (t >= f0 && t < f1) {
      ++den;
      if (in[t * D] > th) ++num;
    }
  out[g] = (static_cast<float>(num) >= static_cast<float>(den) * o.proportion_threshold) ? 1.0f : 0.0f;
}

int launch_vad(const snf_vad_options& o, const float* in, int in_cols, const int64_t* frame_offsets,
               int64_t n_utts, int64_t total_frames, float* thr_scratch, float* out,
               hipStream_t stream) {
  if (total_frames <= 0) return SNF_OK;
  hipLaunchKernelGGL(vad_threshold_kernel, dim3(static_cast<unsigned>(n_utts)), dim3(256), 0, stream,
                     o, in, in_cols, frame_offsets, thr_scratch);
  hipLaunchKernelGGL(vad_kernel, dim3(static_cast<unsigned>((total_frames + 255) / 256)), dim3(256),
                     0, stream, o, in, in_cols, frame_offsets, n_utts, total_frames, thr_scratch, out);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

// CMVN statistics of every utterance ([KALDI-UPSTREAM] transform/cmvn.cc AccCmvnStats): one
// workgroup per utterance, deterministic reduction; stats[u] = [2, D+1] doubles.
__global__ __launch_bounds__(256) void cmvn_stats_kernel(const float* __restrict__ in, const int D,
                                                         const int64_t* __restrict__ frame_offsets,
                                                         const float* __restrict__ weights,
                                                         double* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* part = reinterpret_cast<double*>(smem);  // [R][2 D + 1]
  const int64_t u = blockIdx.x;
  const int64_t f0 = frame_offsets[u], T = frame_offsets[u + 1] - f0;
  const int W = 2 * D + 1;
  const int R = blockDim.x / D > 0 ? blockDim.x / D : 1;  // rows processed in parallel
  const int r = threadIdx.x / D, c = threadIdx.x - r * D;
  if (r < R) {
    double s = 0.0, q = 0.0, n = 0.0;
    for (int64_t t = r; t < T; t += R) {
      const float w = weights ? weights[f0 + t] : 1.0f;
      if (w != 0.0f) {
        const float x = in[(f0 + t) * D + c];
        s += static_cast<double>(x * w);
        q += static_cast<double>(x * x * w);
        n += static_cast<double>(w);
      }
    }
    part[r * W + c] = s;
    part[r * W + D + c] = q;
    if (c == 0) part[r * W + 2 * D] = n;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < W; e += blockDim.x) {
    double tot = 0.0;
    for (int k = 0; k < R; ++k) tot += part[k * W + e];
    double* st = stats + u * 2 * (D + 1);
    if (e < D) st[e] = tot;                       // row 0: sums
    else if (e < 2 * D) st[(D + 1) + (e - D)] = tot;  // row 1: sums of squares
    else { st[D] = tot; st[(D + 1) + D] = 0.0; }  // count; stats(1, D) unused
  }
}

// more than 256 columns (spectrograms): every thread owns columns c, c + 256, ... over all the rows
__global__ __launch_bounds__(256) void cmvn_stats_wide_kernel(const float* __restrict__ in, const int D,
                                                              const int64_t* __restrict__ frame_offsets,
                                                              const float* __restrict__ weights,
                                                              double* __restrict__ stats) {
  const int64_t u = blockIdx.x;
  const int64_t f0 = frame_offsets[u], T = frame_offsets[u + 1] - f0;
  double* st = stats + u * 2 * (D + 1);
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    double s = 0.0, q = 0.0;
    for (int64_t t = 0; t < T; ++t) {
      const float w = weights ? weights[f0 + t] : 1.0f;
      if (w != 0.0f) {
        const float x = in[(f0 + t) * D + c];
        s += static_cast<double>(x * w);
        q += static_cast<double>(x * x * w);
      }
    }
    st[c] = s;
    st[(D + 1) + c] = q;
  }
  if (threadIdx.x == 0) {
    double n = 0.0;
    for (int64_t t = 0; t < T; ++t) {
      const float w = weights ? weights[f0 + t] : 1.0f;
      if (w != 0.0f) n += static_cast<double>(w);
    }
    st[D] = n;
    st[(D + 1) + D] = 0.0;
  }
}

int launch_cmvn_stats(const float* in, int in_cols, const int64_t* frame_offsets,
                      const float* weights, int64_t n_utts, double* stats, hipStream_t stream) {
  if (n_utts <= 0) return SNF_OK;
  if (in_cols > 256) {
    hipLaunchKernelGGL(cmvn_stats_wide_kernel, dim3(static_cast<unsigned>(n_utts)), dim3(256), 0, stream,
                       in, in_cols, frame_offsets, weights, stats);
    SNF_HIP_CHECK(hipGetLastError());
    return SNF_OK;
  }
  const int R = 256 / in_cols > 0 ? 256 / in_cols : 1;
  const size_t lds = sizeof(double) * R * (2 * in_cols + 1);
  hipLaunchKernelGGL(cmvn_stats_kernel, dim3(static_cast<unsigned>(n_utts)), dim3(256), lds, stream,
                     in, in_cols, frame_offsets, weights, stats);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

// out = in * scale + offset with two roundings (Kaldi MulColsVec then AddVecToRows);
// norm[group][2][D] = {offset, scale} in float.
__global__ void cmvn_apply_kernel(const float* __restrict__ in, const int D,
                                  const int64_t* __restrict__ frame_offsets, const int64_t u0,
                                  const int32_t* __restrict__ group,
                                  const float* __restrict__ norm, const int scale_it,
                                  float* __restrict__ out) {
  // blockIdx.y = utterance (no per-element search), blockIdx.x = 256-element chunk of its block
  const int64_t u = u0 + blockIdx.y;
  const int64_t e0 = frame_offsets[u] * D, ne = (frame_offsets[u + 1] - frame_offsets[u]) * D;
  const int64_t k = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= ne) return;
  const int64_t idx = e0 + k;
  const int c = static_cast<int>(k % D);
  const float* __restrict__ nm = norm + static_cast<int64_t>(group ? group[u] : 0) * 2 * D;
  float x = in[idx];
  if (scale_it) x = __fmul_rn(x, nm[D + c]);
  out[idx] = __fadd_rn(x, nm[c]);
}

int launch_cmvn_apply(const float* in, int in_cols, const int64_t* frame_offsets, int64_t n_utts,
                      int64_t max_frames, const int32_t* group, const float* norm, int scale_it,
                      float* out, hipStream_t stream) {
  const int64_t per_utt = max_frames * in_cols;
  if (per_utt <= 0 || n_utts <= 0) return SNF_OK;
  for (int64_t u0 = 0; u0 < n_utts; u0 += 65535) {  // grid.y limit
    const int64_t nu = n_utts - u0 < 65535 ? n_utts - u0 : 65535;
    hipLaunchKernelGGL(cmvn_apply_kernel,
                       dim3(static_cast<unsigned>((per_utt + 255) / 256), static_cast<unsigned>(nu)),
                       dim3(256), 0, stream, in, in_cols, frame_offsets, u0, group, norm, scale_it, out);
    SNF_HIP_CHECK(hipGetLastError());
  }
  return SNF_OK;
}

// Sliding-window CMN ([KALDI-UPSTREAM] feat/feature-functions.cc SlidingWindowCmnInternal): one thread
// per (utterance, column) walks the frames with Kaldi's incremental double-precision window sums.
__global__ void sliding_cmvn_kernel(const snf_sliding_cmvn_options o, const float* __restrict__ in,
                                    const int D, const int64_t* __restrict__ frame_offsets,
                                    const int64_t n_utts, float* __restrict__ out) {
  const int64_t tid = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (tid >= n_utts * D) return;
  const int64_t u = tid / D;
  const int d = static_cast<int>(tid - u * D);
  const int64_t f0 = frame_offsets[u], T = frame_offsets[u + 1] - f0;
  const float* __restrict__ x = in + f0 * D + d;
  float* __restrict__ y = out + f0 * D + d;
  double cur_sum = 0.0, cur_sumsq = 0.0;
  int64_t last_start = -1, last_end = -1;
  for (int64_t t = 0; t < T; ++t) {
    int64_t ws, we;
    if (o.center) { ws = t - (o.cmn_window / 2); we = ws + o.cmn_window; }
    else { ws = t - o.cmn_window; we = t + 1; }
    if (ws < 0) { we -= ws; ws = 0; }
    if (!o.center && we > t) we = (t + 1 > o.min_window) ? t + 1 : o.min_window;
    if (we > T) { ws -= (we - T); we = T; if (ws < 0) ws = 0; }
    if (last_start == -1) {
      for (int64_t r = ws; r < we; ++r) {
        const double v = x[r * D];
        cur_sum += v;
        cur_sumsq += v * v;
      }
    } else {
      if (ws > last_start) {
        const double v = x[last_start * D];
        cur_sum += -1.0 * v;
        if (o.normalize_variance) cur_sumsq += -1.0 * v * v;
      }
      if (we > last_end) {
        const double v = x[last_end * D];
        cur_sum += 1.0 * v;
        if (o.normalize_variance) cur_sumsq += 1.0 * v * v;
      }
    }
    const int64_t wf = we - ws;
    last_start = ws;
    last_end = we;
    double r = static_cast<double>(x[t * D]) + (-1.0 / static_cast<double>(wf)) * cur_sum;
    if (o.normalize_variance) {
      if (wf == 1) r = 0.0;
      else {
        double v = cur_sumsq * (1.0 / static_cast<double>(wf));
        v += (-1.0 / (static_cast<double>(wf) * static_cast<double>(wf))) * cur_sum * cur_sum;
        if (v < 1.0e-10) v = 1.0e-10;
        r *= pow(v, -0.5);
      }
    }
    y[t * D] = static_cast<float>(r);
  }
}

int launch_sliding_cmvn(const snf_sliding_cmvn_options& o, const float* in, int in_cols,
                        const int64_t* frame_offsets, int64_t n_utts, float* out,
                        hipStream_t stream) {
  const int64_t total = n_utts * in_cols;
  if (total <= 0) return SNF_OK;
  hipLaunchKernelGGL(sliding_cmvn_kernel, dim3(static_cast<unsigned>((total + 63) / 64)), dim3(64), 0,
                     stream, o, in, in_cols, frame_offsets, n_utts, out);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

// Column-wise concatenation of two feature blocks per utterance (reference Features.concatenate,
// features.py:386-437: the longer side is trimmed to the shorter one within the caller's tolerance):
// out[u][t] = [a[u][t], b[u][t]] for t < rows_out(u).  A workgroup owns kConcatRows consecutive output rows: one
// lane per row finds the row's utterance (the only search; it used to be one per ELEMENT) and leaves where the row
// starts in `a` and `b` in LDS; then the 256 lanes walk the rows' elements in output order - coalesced stores, and
// loads that are contiguous wherever consecutive rows belong to one utterance.  HBM-bound by design:
// 4 (ca + cb) bytes read + as many written per row.
constexpr int kConcatRows = 32;
__global__ __launch_bounds__(256) void concat_columns_kernel(
    const float* __restrict__ a, const int ca, const int64_t* __restrict__ off_a, const float* __restrict__ bm,
    const int cb, const int64_t* __restrict__ off_b, const int64_t n_utts, float* __restrict__ out,
    const int64_t* __restrict__ off_o, const int64_t total_rows) {
  __shared__ int64_t row_a[kConcatRows], row_b[kConcatRows];
  const int co = ca + cb;
  const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kConcatRows;
  const int rows = static_cast<int>(total_rows - row0 < kConcatRows ? total_rows - row0 : kConcatRows);
  if (static_cast<int>(threadIdx.x) < rows) {
    const int64_t g = row0 + threadIdx.x;
    const int64_t u = find_utt(off_o, n_utts, g);
    const int64_t t = g - off_o[u];
    row_a[threadIdx.x] = (off_a[u] + t) * ca;
    row_b[threadIdx.x] = (off_b[u] + t) * cb - ca;   // (indexed with the output column)
  }
  __syncthreads();
  float* __restrict__ dst = out + row0 * co;
  const int count = rows * co;
  for (int i = threadIdx.x; i < count; i += 256) {
    const int r = i / co, c = i - r * co;
    dst[i] = c < ca ? a[row_a[r] + c] : bm[row_b[r] + c];
  }
}

int launch_concat_columns(const float* a, int cols_a, const int64_t* d_off_a, const float* b, int cols_b,
                          const int64_t* d_off_b, int64_t n_utts, float* out, const int64_t* d_off_out,
                          int64_t total_rows, hipStream_t stream) {
  if (total_rows <= 0) return SNF_OK;
  hipLaunchKernelGGL(concat_columns_kernel, dim3(static_cast<unsigned>((total_rows + kConcatRows - 1) / kConcatRows)),
                     dim3(256), 0, stream, a, cols_a, d_off_a, b, cols_b, d_off_b, n_utts, out, d_off_out,
                     total_rows);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

// Features.validate's data check (reference features.py:170-215 `is_valid`: "data contains non-finite
// numbers") for a block that is still in HBM: the number of NaN / +-Inf among n floats.  An exponent
// field of all ones is the test; 16-byte loads over the aligned body, one atomic per workgroup.
__global__ void __launch_bounds__(256) count_nonfinite_kernel(const float* __restrict__ x, uint64_t n,
                                                              unsigned long long* __restrict__ count) {
  const uint64_t n4 = n >> 2;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4* x4 = reinterpret_cast<const u32x4*>(x);
  unsigned bad = 0;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
  for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const u32x4 v = __builtin_nontemporal_load(x4 + i);
    bad += ((v.x & 0x7F800000u) == 0x7F800000u) + ((v.y & 0x7F800000u) == 0x7F800000u) +
           ((v.z & 0x7F800000u) == 0x7F800000u) + ((v.w & 0x7F800000u) == 0x7F800000u);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const unsigned v = __float_as_uint(x[(n4 << 2) + threadIdx.x]);
    bad += (v & 0x7F800000u) == 0x7F800000u;
  }
  if (__builtin_amdgcn_ballot_w64(bad != 0) == 0) return;  // (the usual case: nothing to add)
  atomicAdd(count, static_cast<unsigned long long>(bad));
}

int launch_count_nonfinite(const float* x, uint64_t n, unsigned long long* d_count, hipStream_t stream) {
  if (n == 0) return SNF_OK;
  const uint64_t want = (n / 4 + 255) / 256;
  const unsigned blocks = static_cast<unsigned>(want < 1 ? 1 : (want > 256 * 16 ? 256 * 16 : want));
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3(blocks), dim3(256), 0, stream, x, n, d_count);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

}  // namespace snf
