// which conditions allow > 1 VALU instr per 4 clk per SIMD on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 32768;
// MODE 0: 8 independent fma chains; 1: 1 dependent chain (8 instr serial); 2: 2 chains; 3: fmac e32 (VOP2) 8 independent
// 4: mix add/sub/mul/fmac VOP2 8 independent; 5: 8 indep v_fma with 3 distinct vgpr sources
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, float seed) {
  extern __shared__ char smem[];
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float c = 1.0001f + seed * 1e-9f, d = 0.0001f + seed * 1e-9f;
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 1) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
                   "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 2) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
                   "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 3) {
      asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                   "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 4) {
      asm volatile("v_add_f32 %0, %0, %8\n v_sub_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_fmac_f32 %3, %8, %9\n"
                   "v_add_f32 %4, %4, %9\n v_sub_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_fmac_f32 %7, %9, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 5) {  // butterfly-like: outputs depend on pairs
      asm volatile("v_add_f32 %0, %0, %1\n v_sub_f32 %1, %0, %1\n v_add_f32 %2, %2, %3\n v_sub_f32 %3, %2, %3\n"
                   "v_add_f32 %4, %4, %5\n v_sub_f32 %5, %4, %5\n v_add_f32 %6, %6, %7\n v_sub_f32 %7, %6, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (r == 12345.678f) out[0] = r + smem[0];
}
template <int MODE>
int run(const char* name, int waves_per_simd) {
  float* out;
  CHECK(hipMalloc(&out, 4));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  // one wave per block; LDS request sets the residency: 160 KB / (4 * waves_per_simd) per block
  const int lds = 160 * 1024 / (4 * waves_per_simd) - 512;
  const int blocks = 256 * 4 * waves_per_simd;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), lds, 0, out, 1.0f);
  CHECK(hipDeviceSynchronize());
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), lds, 0, out, 1.0f);
  hipEventRecord(e1);
  CHECK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = double(blocks) * ITERS * 8;
  printf("%-22s waves/SIMD %d  %8.3f ms  %7.1f G wave-instr/s  %.2f cyc/instr/SIMD @2.4GHz\n", name, waves_per_simd, ms,
         winstr / (ms * 1e-3) / 1e9, ms * 1e-3 * 2.4e9 / (winstr / 1024.0));
  hipFree(out);
  return 0;
}
int main() {
  for (int w : {1, 2, 3, 4, 5, 6, 8}) {
    run<0>("fma x8 indep", w);
    run<1>("fma dependent", w);
    run<4>("add/sub/mul/fmac mix", w);
    run<5>("butterfly pairs", w);
  }
  return 0;
}
