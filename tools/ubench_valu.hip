// Developer microbenchmark (not part of the product): VALU issue rates on gfx950 that drive the
// design of kernels_fbank512.hip.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float c = 1.0001f, d = 0.0001f;
  const float2v pc = {c, c}, pd = {d, d};
  for (int i = 0; i < ITERS; ++i) {
    if (MODE == 0) {  // v_fma_f32
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
    } else if (MODE == 1) {  // v_add_f32
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
    } else if (MODE == 2) {  // v_pk_fma_f32
      asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
    } else if (MODE == 3) {  // v_pk_add_f32
      asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                   "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pd));
    } else if (MODE == 4) {  // v_pk_mul_f32
      asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
    } else if (MODE == 5) {  // v_add_f32 with DPP row_ror:1 on src0
      asm volatile("v_add_f32_dpp %0, %0, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %2, %2, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %4, %4, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                   "v_add_f32_dpp %6, %6, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
    } else if (MODE == 6) {  // v_mul_f32
      asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                   "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (MODE == 7) {  // v_cvt_f32_i32 + v_bfe_i32 mix
      asm volatile("v_cvt_f32_i32 %0, %1\n v_cvt_f32_i32 %1, %2\n v_cvt_f32_i32 %2, %3\n v_cvt_f32_i32 %3, %4\n"
                   "v_cvt_f32_i32 %4, %5\n v_cvt_f32_i32 %5, %6\n v_cvt_f32_i32 %6, %7\n v_cvt_f32_i32 %7, %0\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 8) {  // v_log_f32
      asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                   "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
  if (r == 12345.678f) out[0] = r;
}

template <int MODE>
int run(const char* name, int lane_ops_per_instr) {
  float* out;
  CHECK(hipMalloc(&out, 4));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  CHECK(hipDeviceSynchronize());
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f);
  hipEventRecord(e1);
  CHECK(hipDeviceSynchronize());
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = double(blocks) * 4 /*waves*/ * ITERS * 8;
  const double per_simd_cycles = ms * 1e-3 * 2.4e9 / (winstr / 1024.0);
  printf("%-28s %8.3f ms  %7.2f G wave-instr/s  %.2f cyc/wave-instr/SIMD @2.4GHz  %.1f T lane-ops/s\n", name, ms,
         winstr / (ms * 1e-3) / 1e9, per_simd_cycles, winstr * 64 * lane_ops_per_instr / (ms * 1e-3) / 1e12);
  hipFree(out);
  return 0;
}

int main() {
  run<0>("v_fma_f32", 1);
  run<1>("v_add_f32", 1);
  run<6>("v_mul_f32", 1);
  run<2>("v_pk_fma_f32", 2);
  run<3>("v_pk_add_f32", 2);
  run<4>("v_pk_mul_f32", 2);
  run<5>("v_add_f32_dpp row_ror", 1);
  run<7>("v_cvt_f32_i32", 1);
  run<8>("v_log_f32", 1);
  return 0;
}
