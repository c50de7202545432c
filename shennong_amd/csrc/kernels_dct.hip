// MFCC tail for plans the register-resident kernels cover only up to the log-mel energies (round 6): more
// than 16 cepstra (Kaldi's "hires" MFCC: 40 bins, 40 cepstra) or more than 64 mel bins.  The filterbank kernel
// writes [log energy |] log-mel rows to a scratch, this kernel forms the cepstra:
// [KALDI-UPSTREAM] MfccComputer::Compute (feature-mfcc.cc; reached by the reference at
// shennong/processor/base.py:429-431): feature = DCT x log-mel; x lifter; c0 := log energy if use_energy;
// htk_compat: c0 moves to the last column (x sqrt 2 when it is not the energy).
// One wavefront owns 64 consecutive frames: their rows come in through LDS (coalesced), lane f walks the row of
// frame f (row pitch odd or padded to odd: conflict-free), the DCT coefficients are wave-uniform (LDS broadcasts of the
// transposed matrix), eight cepstra at a time; a lane writes the cepstra of its frame straight to memory.  HBM-bound by design:
// 4 (num_bins + use_energy + num_ceps) bytes per frame.
#include <float.h>

#include "snf_internal.h"

namespace snf {

namespace {

__global__ __launch_bounds__(1024) void mfcc_dct_kernel(
    const float* __restrict__ in, const int in_cols, const int mel_col, const int num_bins, const int num_ceps,
    const float* __restrict__ dct_t /* [num_bins][num_ceps8] */, const int num_ceps8,
    const float* __restrict__ lifter, const int use_energy, const int htk_compat, const int64_t total_frames,
    float* __restrict__ out, const int out_cols, const int in_pitch) {
  extern __shared__ __attribute__((aligned(16))) float dct_lds[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  // the transposed DCT matrix first (every lane reads the same eight coefficients: LDS broadcasts; scalar loads
  // of them cost a round trip per mel bin and eight cepstra)
  float* __restrict__ dt = dct_lds;
  const int dt_floats = num_bins * num_ceps8;
  for (int i = threadIdx.x; i < dt_floats; i += blockDim.x) dt[i] = dct_t[i];
  // ... and the lifter behind it (a load per cepstrum and tile from memory is a round trip each)
  float* __restrict__ lf = dct_lds + ((dt_floats + 3) & ~3);
  for (int i = threadIdx.x; i < num_ceps8; i += blockDim.x) lf[i] = (lifter && i < num_ceps) ? lifter[i] : 1.0f;
  __syncthreads();
  // (one tile of 64 rows per wave and nothing else: the LDS of a CU holds 12-16 waves' tiles of 41-float rows; with
  // a second tile for the rows on their way out it held 4 waves per CU and the kernel took 1.2 ms per 2.98 M frames)
  float* __restrict__ tin = lf + num_ceps8 + wid * (64 * in_pitch + 64 * 17);
  float* __restrict__ tout = tin + 64 * in_pitch;   // sixteen cepstra of the 64 frames on their way out
  const int64_t n_tiles = (total_frames + 63) >> 6;
  for (int64_t tile = static_cast<int64_t>(blockIdx.x) * n_waves + wid; tile < n_tiles;
       tile += static_cast<int64_t>(gridDim.x) * n_waves) {
    const int64_t f0 = tile << 6;
    const int nfr = static_cast<int>(total_frames - f0 < 64 ? total_frames - f0 : 64);
    // rows in: nfr x in_cols contiguous floats -> LDS rows of pitch in_pitch (odd: a lane per row, no bank conflict)
    const float* __restrict__ src = in + f0 * in_cols;
    if (in_pitch == in_cols) {
      for (int i = lane; i < nfr * in_cols; i += 64) tin[i] = src[i];
    } else {
      for (int i = lane; i < nfr * in_cols; i += 64) {
        const int r = i / in_cols, c = i - r * in_cols;
        tin[r * in_pitch + c] = src[i];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const float* __restrict__ row = tin + (lane < nfr ? lane : 0) * in_pitch;
    const float energy = use_energy ? row[0] : 0.0f;
    float* __restrict__ obase = out + f0 * out_cols;
    for (int c0 = 0; c0 < num_ceps; c0 += 16) {   // (the matrix rows are padded to whole sixteens)
      float acc[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = 0.0f;
#pragma unroll 2
      for (int m = 0; m < num_bins; ++m) {
        const float x = row[mel_col + m];
        const float4* __restrict__ d = reinterpret_cast<const float4*>(dt + m * num_ceps8 + c0);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 dq = d[q4];
          acc[4 * q4] += dq.x * x;
          acc[4 * q4 + 1] += dq.y * x;
          acc[4 * q4 + 2] += dq.z * x;
          acc[4 * q4 + 3] += dq.w * x;
        }
      }
      // lifter, c0 := energy, then through LDS: 16 cepstra of a frame leave as one 64-byte stretch (a dword per
      // lane and row, 40 scattered stores per tile, kept the L2 busy with 119 M four-byte writes per launch)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float v = acc[k] * lf[c0 + k];
        if (c0 + k == 0) {
          if (use_energy) v = energy;
          else if (htk_compat) v = static_cast<float>(static_cast<double>(v) * 1.4142135623730950488016887);
        }
        tout[lane * 17 + k] = v;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      // element i = lane + 64 t of the 64 x 16 block: row i / 16, cepstrum c0 + i % 16 -> its column (htk: c0 last)
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int r = 4 * t + (lane >> 4), k = lane & 15, c = c0 + k;
        if (r < nfr && c < num_ceps) {
          const int oc = htk_compat ? (c == 0 ? num_ceps - 1 : c - 1) : c;
          obase[static_cast<int64_t>(r) * out_cols + oc] = tout[r * 17 + k];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

}  // namespace

// `in`: [total_frames][in_cols] = [log energy (use_energy) | num_bins log-mel energies]; `dct_t`: the DCT matrix
// transposed, [num_bins][num_ceps8] with num_ceps8 = num_ceps rounded up to 16 (zero columns behind the cepstra)
int launch_mfcc_dct(const float* in, int in_cols, int num_bins, int num_ceps, const float* dct_t, const float* lifter,
                    int use_energy, int htk_compat, int64_t total_frames, float* out, int out_cols,
                    hipStream_t stream) {
  if (total_frames <= 0) return SNF_OK;
  const int num_ceps8 = (num_ceps + 15) & ~15;
  const int in_pitch = in_cols | 1;   // an odd row pitch: a lane per row, no bank conflict
  // one workgroup per CU, as many waves as tiles fit beside the matrix in its LDS (14 for 40 bins + energy)
  const size_t dt_bytes = sizeof(float) * (((static_cast<size_t>(num_bins) * num_ceps8 + 3) & ~static_cast<size_t>(3)) + num_ceps8);
  const size_t tile_bytes = sizeof(float) * 64 * static_cast<size_t>(in_pitch + 17);
  const size_t budget = 159 * 1024;
  if (dt_bytes + tile_bytes > budget) return set_error(SNF_E_RUNTIME, "mfcc_dct: rows too wide for LDS");
  int n_waves = static_cast<int>((budget - dt_bytes) / tile_bytes);
  if (n_waves > 16) n_waves = 16;
  const size_t lds = dt_bytes + n_waves * tile_bytes;
  const int64_t tiles = (total_frames + 63) / 64;
  int64_t blocks = (tiles + n_waves - 1) / n_waves;
  if (blocks > 256) blocks = 256;
  if (lds > 64 * 1024)
    SNF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfcc_dct_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL(mfcc_dct_kernel, dim3(static_cast<unsigned>(blocks)), dim3(n_waves * 64), lds, stream, in,
                     in_cols, use_energy ? 1 : 0, num_bins, num_ceps, dct_t, num_ceps8, lifter, use_energy,
                     htk_compat, total_frames, out, out_cols, in_pitch);
  SNF_HIP_CHECK(hipGetLastError());
  return SNF_OK;
}

}  // namespace snf
