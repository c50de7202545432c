// µbench (round 4): does a wave's f32 MFMA work hide behind the vector instructions of the other waves
// of its SIMD?  The 512-point kernel keeps the vector pipe busy ~65 % of the time with 4 waves per SIMD
// and the matrix pipe ~8 %; moving the first register FFT-16 pass to v_mfma_f32_16x16x4_f32 (a dense
// real DFT-16, exact f32) only pays if the two pipes overlap.  Per iteration every wave issues NV plain
// v_fma_f32 (8 independent chains) and NM matrix instructions (4 accumulators), evenly interleaved.
//   KIND 0: v_mfma_f32_16x16x4_f32 (32 clocks per SIMD)   KIND 1: v_mfma_f32_4x4x1_16b_f32 (8 clocks)
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_coissue.hip -o scratch/ub_coissue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kIters = 2000;

// GV vector instructions (a multiple of 8), then one matrix instruction; repeated NG times per iteration
template <int GV, int NG, int KIND, bool WITH_M>
__global__ __launch_bounds__(1024) void k_mix(float* out, float seed) {
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x + i;
    b[i] = 0.5f * a[i] + 1.0f;
  }
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, seed, seed, seed};
  const float ma = seed * 0.001f, mb = 1.0f + seed * 0.002f;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  const i32x4 h8a = {0x3c003c00, 0x3c003c00, 0x3c003c00, 0x3c003c00}, h8b = {0x38003800, 0x38003800, 0x38003800, 0x38003800};
  const i32x2 h4a = {0x3c003c00, 0x3c003c00}, h4b = {0x38003800, 0x38003800};
  i32x4 iacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) iacc[i] = i32x4{1, 2, 3, 4};
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int v = 0; v < GV / 8; ++v)
        asm volatile(
            "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %10, %11\n"
            "v_fma_f32 %3, %3, %11, %12\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %13, %14\n"
            "v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %15, %8"
            : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
            : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
      if (WITH_M) {
        if (KIND == 0)
          asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(ma), "v"(mb));
        else if (KIND == 1)
          asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(ma), "v"(mb));
        else if (KIND == 2)
          asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(h8a), "v"(h8b));
        else if (KIND == 3)
          asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(iacc[g & 3]) : "v"(h8a), "v"(h8b));
        else
          asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[g & 3]) : "v"(h4a), "v"(h4b));
      }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + iacc[i][0] + iacc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int GV, int NG, int KIND, bool WITH_M>
static int run(const char* label, float* d_out, int waves_per_simd) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int threads = waves_per_simd * 256;
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<GV, NG, KIND, WITH_M>), dim3(256), dim3(threads), 0, 0, d_out, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  // clocks one SIMD spends per iteration of ONE wave (x waves_per_simd = per round of all its waves)
  const double clk = best * 1e-3 * 2.4e9 / kIters / waves_per_simd;
  printf("%-44s w/SIMD %d  %.3f ms  %.0f clk per wave-iteration (%d valu + %d mfma)\n", label, waves_per_simd,
         best, clk, GV * NG, WITH_M ? NG : 0);
  return 0;
}

int main() {
  float* d_out;
  CK(hipMalloc(&d_out, 256 * 1024 * sizeof(float)));
  for (int w : {4, 2}) {
    run<24, 8, 2, true>("192 fma + 8 mfma16x16x32_f16 (1 per 24)", d_out, w);
    run<0, 8, 2, true>("8 mfma16x16x32_f16 alone", d_out, w);
    run<8, 24, 2, true>("192 fma + 24 mfma16x16x32_f16 (1 per 8)", d_out, w);
    run<24, 8, 3, true>("192 fma + 8 mfma16x16x64_i8 (1 per 24)", d_out, w);
    run<0, 8, 3, true>("8 mfma16x16x64_i8 alone", d_out, w);
    run<24, 8, 4, true>("192 fma + 8 mfma16x16x16_f16 (1 per 24)", d_out, w);
    run<0, 8, 4, true>("8 mfma16x16x16_f16 alone", d_out, w);
    run<8, 24, 4, true>("192 fma + 24 mfma16x16x16_f16 (1 per 8)", d_out, w);
    run<24, 8, 0, false>("192 fma", d_out, w);
    run<24, 8, 0, true>("192 fma + 8 mfma16x16x4 (1 per 24)", d_out, w);
    run<8, 8, 0, true>("64 fma + 8 mfma16x16x4 (1 per 8)", d_out, w);
    run<0, 8, 0, true>("8 mfma16x16x4 alone", d_out, w);
    run<24, 8, 1, true>("192 fma + 8 mfma4x4x1 (1 per 24)", d_out, w);
    run<0, 8, 1, true>("8 mfma4x4x1 alone", d_out, w);
    run<16, 16, 0, true>("256 fma + 16 mfma16x16x4 (1 per 16)", d_out, w);
    run<16, 16, 0, false>("256 fma", d_out, w);
  }
  CK(hipFree(d_out));
  return 0;
}
