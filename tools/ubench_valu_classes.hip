// µbench: issue rate of the VALU instruction classes used by fbank512_kernel (8 independent streams per
// wave, 4 waves per SIMD).  Developer tool.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 2048;
#define REP8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float* out, float seed, int sel) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0 * 0.5f, b1 = a1 * 0.5f, b2 = a2 * .5f, b3 = a3 * .5f, b4 = a4 * .5f, b5 = a5 * .5f, b6 = a6 * .5f, b7 = a7 * .5f;
  const float c = 1.0001f, d = 0.0001f;
  const unsigned long long mask = sel ? 0x5555555555555555ull : 0xAAAAAAAAAAAAAAAAull;
  for (int i = 0; i < ITERS; ++i) {
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define INB "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(b6), "v"(b7)
    if (MODE == 0) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %2, %2, %10\n v_add_f32 %3, %3, %11\n v_add_f32 %4, %4, %12\n v_add_f32 %5, %5, %13\n v_add_f32 %6, %6, %14\n v_add_f32 %7, %7, %15" : OPS : INB);
    if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %8, %16\n v_fma_f32 %1, %1, %9, %16\n v_fma_f32 %2, %2, %10, %16\n v_fma_f32 %3, %3, %11, %16\n v_fma_f32 %4, %4, %12, %16\n v_fma_f32 %5, %5, %13, %16\n v_fma_f32 %6, %6, %14, %16\n v_fma_f32 %7, %7, %15, %16" : OPS : INB, "v"(d));
    if (MODE == 2) asm volatile("v_fmac_f32 %0, %8, %16\n v_fmac_f32 %1, %9, %16\n v_fmac_f32 %2, %10, %16\n v_fmac_f32 %3, %11, %16\n v_fmac_f32 %4, %12, %16\n v_fmac_f32 %5, %13, %16\n v_fmac_f32 %6, %14, %16\n v_fmac_f32 %7, %15, %16" : OPS : INB, "v"(d));
    if (MODE == 3) asm volatile("v_mul_f32 %0, 0x3f317218, %0\n v_mul_f32 %1, 0x3f317218, %1\n v_mul_f32 %2, 0x3f317218, %2\n v_mul_f32 %3, 0x3f317218, %3\n v_mul_f32 %4, 0x3f317218, %4\n v_mul_f32 %5, 0x3f317218, %5\n v_mul_f32 %6, 0x3f317218, %6\n v_mul_f32 %7, 0x3f317218, %7" : OPS);
    if (MODE == 4) asm volatile("v_mov_b32_dpp %0, %8 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %9 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %10 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %11 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %12 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %13 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %14 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %15 row_mirror row_mask:0xf bank_mask:0xf" : OPS : INB);
    if (MODE == 5) asm volatile("v_add_f32_dpp %0, %8, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %9, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %10, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %11, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %12, %4 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %13, %5 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %14, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %15, %7 row_ror:1 row_mask:0xf bank_mask:0xf" : OPS : INB);
    if (MODE == 6) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %1, sext(%9) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_i32_sdwa %2, sext(%10) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %3, sext(%11) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_i32_sdwa %4, sext(%12) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %5, sext(%13) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_i32_sdwa %6, sext(%14) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n v_cvt_f32_i32_sdwa %7, sext(%15) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : OPS : INB);
    if (MODE == 7) asm volatile("v_cndmask_b32 %0, %0, %8, %16\n v_cndmask_b32 %1, %1, %9, %16\n v_cndmask_b32 %2, %2, %10, %16\n v_cndmask_b32 %3, %3, %11, %16\n v_cndmask_b32 %4, %4, %12, %16\n v_cndmask_b32 %5, %5, %13, %16\n v_cndmask_b32 %6, %6, %14, %16\n v_cndmask_b32 %7, %7, %15, %16" : OPS : INB, "s"(mask));
    if (MODE == 8) asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7" : OPS);
    if (MODE == 9) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %9\n v_mov_b32 %2, %10\n v_mov_b32 %3, %11\n v_mov_b32 %4, %12\n v_mov_b32 %5, %13\n v_mov_b32 %6, %14\n v_mov_b32 %7, %15" : OPS : INB);
    if (MODE == 10) asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %9\n v_mul_lo_u32 %2, %2, %10\n v_mul_lo_u32 %3, %3, %11\n v_mul_lo_u32 %4, %4, %12\n v_mul_lo_u32 %5, %5, %13\n v_mul_lo_u32 %6, %6, %14\n v_mul_lo_u32 %7, %7, %15" : OPS : INB);
    if (MODE == 11) asm volatile("v_med3_f32 %0, %0, %8, %16\n v_med3_f32 %1, %1, %9, %16\n v_med3_f32 %2, %2, %10, %16\n v_med3_f32 %3, %3, %11, %16\n v_med3_f32 %4, %4, %12, %16\n v_med3_f32 %5, %5, %13, %16\n v_med3_f32 %6, %6, %14, %16\n v_med3_f32 %7, %7, %15, %16" : OPS : INB, "v"(d));
    if (MODE == 12) asm volatile("v_dot2c_i32_i16 %0, 0x10001, %8\n v_dot2c_i32_i16 %1, 0x10001, %9\n v_dot2c_i32_i16 %2, 0x10001, %10\n v_dot2c_i32_i16 %3, 0x10001, %11\n v_dot2c_i32_i16 %4, 0x10001, %12\n v_dot2c_i32_i16 %5, 0x10001, %13\n v_dot2c_i32_i16 %6, 0x10001, %14\n v_dot2c_i32_i16 %7, 0x10001, %15" : OPS : INB);
    if (MODE == 13) asm volatile("v_fmamk_f32 %0, %0, 0x3f317218, %8\n v_fmamk_f32 %1, %1, 0x3f317218, %9\n v_fmamk_f32 %2, %2, 0x3f317218, %10\n v_fmamk_f32 %3, %3, 0x3f317218, %11\n v_fmamk_f32 %4, %4, 0x3f317218, %12\n v_fmamk_f32 %5, %5, 0x3f317218, %13\n v_fmamk_f32 %6, %6, 0x3f317218, %14\n v_fmamk_f32 %7, %7, 0x3f317218, %15" : OPS : INB);
    // dependent chain: 8 adds on ONE accumulator
    if (MODE == 14) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %0, %0, %9\n v_add_f32 %0, %0, %10\n v_add_f32 %0, %0, %11\n v_add_f32 %0, %0, %12\n v_add_f32 %0, %0, %13\n v_add_f32 %0, %0, %14\n v_add_f32 %0, %0, %15" : OPS : INB);
    // 2 interleaved dependent chains
    if (MODE == 15) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_add_f32 %0, %0, %10\n v_add_f32 %1, %1, %11\n v_add_f32 %0, %0, %12\n v_add_f32 %1, %1, %13\n v_add_f32 %0, %0, %14\n v_add_f32 %1, %1, %15" : OPS : INB);
    if (MODE == 16) asm volatile("v_fma_f32 %0, %8, %9, %10\n v_fma_f32 %1, %9, %10, %11\n v_fma_f32 %2, %10, %11, %12\n v_fma_f32 %3, %11, %12, %13\n v_fma_f32 %4, %12, %13, %14\n v_fma_f32 %5, %13, %14, %15\n v_fma_f32 %6, %14, %15, %8\n v_fma_f32 %7, %15, %8, %9" : OPS : INB);
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (r == 12345.678f) out[0] = r;
}
template <int MODE> void run(const char* name, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 4;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0f, 1);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = double(blocks) * 4 * ITERS * 8;
  printf("%-34s %7.3f ms  %.2f clk/wave-instr/SIMD @2.2GHz\n", name, ms, ms * 1e-3 * 2.2e9 / (winstr / 1024.0));
}
int main() {
  float* out; hipMalloc(&out, 4);
  run<0>("v_add_f32 e32", out); run<1>("v_fma_f32 (vop3, 3 vgpr)", out); run<2>("v_fmac_f32 e32", out);
  run<3>("v_mul_f32 literal", out); run<4>("v_mov_b32_dpp row_mirror", out); run<5>("v_add_f32_dpp row_ror", out);
  run<6>("v_cvt_f32_i32_sdwa", out); run<7>("v_cndmask_b32 e64 (sgpr mask)", out); run<8>("v_log_f32", out);
  run<9>("v_mov_b32", out); run<10>("v_mul_lo_u32", out); run<11>("v_med3_f32", out); run<12>("v_dot2c_i32_i16", out);
  run<13>("v_fmamk_f32", out); run<14>("dependent chain v_add (1 acc)", out); run<15>("2 interleaved chains", out);
  run<16>("v_fma_f32 3 distinct src regs", out);
  return 0;
}
