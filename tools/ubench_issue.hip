// µbench: VALU issue rate against waves per SIMD and instruction-level parallelism inside a wave.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 2048;
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * .5f, b3 = a3 * .5f, b4 = a4 * .5f, b5 = a5 * .5f, b6 = a6 * .5f, b7 = a7 * .5f;
  for (int i = 0; i < ITERS; ++i) {
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define INB "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7)
    // 8 independent streams
    if (MODE == 0) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %10\n v_add_f32 %3, %3, %11\n v_add_f32 %4, %4, %12\n v_add_f32 %5, %5, %13\n v_add_f32 %6, %6, %14\n v_add_f32 %7, %7, %15" : OPS : INB);
    // one dependent chain
    if (MODE == 1) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %9\n v_add_f32 %0, %0, %10\n v_add_f32 %0, %0, %11\n v_add_f32 %0, %0, %12\n v_add_f32 %0, %0, %13\n v_add_f32 %0, %0, %14\n v_add_f32 %0, %0, %15" : OPS : INB);
    // 2 interleaved chains
    if (MODE == 2) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %0, %0, %10\n v_add_f32 %1, %1, %11\n v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %13\n v_add_f32 %0, %0, %14\n v_add_f32 %1, %1, %15" : OPS : INB);
    // 8 independent v_fma with 3 vgpr sources (VOP3, 8 bytes)
    if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %9, %10\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %11, %12\n v_fma_f32 %4, %4, %12, %13\n v_fma_f32 %5, %5, %13, %14\n v_fma_f32 %6, %6, %14, %15\n v_fma_f32 %7, %7, %15, %8" : OPS : INB);
    // mix: 6 plain + 1 dpp + 1 cvt (the kernel's rough proportions)
    if (MODE == 4) asm volatile("v_add_f32 %0, %0, %8\n v_fma_f32 %1, %1, %9, %10\n v_sub_f32 %2, %2, %10\n v_add_f32_dpp %3, %11, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n v_mul_f32 %4, %4, %12\n v_fma_f32 %5, %5, %13, %14\n v_cvt_f32_i32 %6, %14\n v_add_f32 %7, %7, %15" : OPS : INB);
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (r == 12345.678f) out[0] = r;
}
template <int MODE> void run(const char* name, float* out, int waves_per_simd) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // one or two workgroups per CU; waves_per_simd * 4 waves per CU
  int blocks = 256, threads = 256 * waves_per_simd;
  if (waves_per_simd == 8) { blocks = 512; threads = 1024; }
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr_per_simd = double(waves_per_simd) * ITERS * 8;
  printf("%-28s waves/SIMD %d  %7.3f ms  %.2f clk/instr/SIMD  %.2f clk/instr/wave @2.2GHz\n", name, waves_per_simd, ms,
         ms * 1e-3 * 2.2e9 / winstr_per_simd, ms * 1e-3 * 2.2e9 / (ITERS * 8.0));
}
int main() {
  float* out; hipMalloc(&out, 4);
  for (int w : {1, 2, 3, 4, 8}) {
    run<0>("8 independent v_add", out, w); run<1>("1 dependent chain", out, w); run<2>("2 chains", out, w);
    run<3>("8 independent v_fma vop3", out, w); run<4>("mix 6 plain+dpp+cvt", out, w);
  }
  return 0;
}
