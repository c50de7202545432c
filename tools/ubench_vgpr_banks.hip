// µbench: VGPR bank conflicts?  v_add_f32 / v_fma_f32 with sources in the same bank (reg % 4) vs different banks
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 16384;
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float* out, float seed) {
  // explicit registers: v10..v41 initialised, results to v50..v57 (never read in the loop)
  asm volatile(
      "v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n"
      "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n"
      "v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n"
      :: "v"(seed) : "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27");
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0)  // different banks
      asm volatile("v_add_f32 v50, v10, v11\n v_add_f32 v51, v12, v13\n v_add_f32 v52, v14, v15\n v_add_f32 v53, v16, v17\n"
                   "v_add_f32 v54, v18, v19\n v_add_f32 v55, v20, v21\n v_add_f32 v56, v22, v23\n v_add_f32 v57, v24, v25\n"
                   ::: "v50","v51","v52","v53","v54","v55","v56","v57");
    if (MODE == 1)  // same bank (stride 4)
      asm volatile("v_add_f32 v50, v10, v14\n v_add_f32 v51, v11, v15\n v_add_f32 v52, v12, v16\n v_add_f32 v53, v13, v17\n"
                   "v_add_f32 v54, v18, v22\n v_add_f32 v55, v19, v23\n v_add_f32 v56, v20, v24\n v_add_f32 v57, v21, v25\n"
                   ::: "v50","v51","v52","v53","v54","v55","v56","v57");
    if (MODE == 2)  // fma, three banks
      asm volatile("v_fma_f32 v50, v10, v11, v12\n v_fma_f32 v51, v13, v14, v15\n v_fma_f32 v52, v16, v17, v18\n v_fma_f32 v53, v19, v20, v21\n"
                   "v_fma_f32 v54, v22, v23, v24\n v_fma_f32 v55, v25, v26, v27\n v_fma_f32 v56, v11, v12, v13\n v_fma_f32 v57, v14, v15, v16\n"
                   ::: "v50","v51","v52","v53","v54","v55","v56","v57");
    if (MODE == 3)  // fma, all three sources in one bank
      asm volatile("v_fma_f32 v50, v10, v14, v18\n v_fma_f32 v51, v11, v15, v19\n v_fma_f32 v52, v12, v16, v20\n v_fma_f32 v53, v13, v17, v21\n"
                   "v_fma_f32 v54, v14, v18, v22\n v_fma_f32 v55, v15, v19, v23\n v_fma_f32 v56, v16, v20, v24\n v_fma_f32 v57, v17, v21, v25\n"
                   ::: "v50","v51","v52","v53","v54","v55","v56","v57");
    if (MODE == 4)  // add where dst bank == src bank
      asm volatile("v_add_f32 v50, v10, v11\n v_add_f32 v50, v12, v13\n v_add_f32 v50, v14, v15\n v_add_f32 v50, v16, v17\n"
                   "v_add_f32 v50, v18, v19\n v_add_f32 v50, v20, v21\n v_add_f32 v50, v22, v23\n v_add_f32 v50, v24, v25\n"
                   ::: "v50");
    if (MODE == 5)  // VOP2 with same register twice
      asm volatile("v_mul_f32 v50, v10, v10\n v_mul_f32 v51, v11, v11\n v_mul_f32 v52, v12, v12\n v_mul_f32 v53, v13, v13\n"
                   "v_mul_f32 v54, v14, v14\n v_mul_f32 v55, v15, v15\n v_mul_f32 v56, v16, v16\n v_mul_f32 v57, v17, v17\n"
                   ::: "v50","v51","v52","v53","v54","v55","v56","v57");
  }
  float r; asm volatile("v_add_f32 %0, v50, v57" : "=v"(r));
  if (r == 12345.678f) out[0] = r;
}
template <int MODE> void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 4;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f); hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = double(blocks) * 4 * ITERS * 8;
  printf("%-40s %7.3f ms  %.2f clk/wave-instr/SIMD @2.2GHz\n", name, ms, ms * 1e-3 * 2.2e9 / (winstr / 1024.0));
}
int main() {
  float* out; hipMalloc(&out, 4);
  run<0>("v_add 2 src different banks", out); run<1>("v_add 2 src same bank", out);
  run<2>("v_fma 3 src different banks", out); run<3>("v_fma 3 src same bank", out);
  run<4>("v_add same dst", out); run<5>("v_mul same reg twice", out);
  run<0>("v_add 2 src different banks (again)", out);
  return 0;
}
