// µbench (round 3): does a long straight-line loop body issue as fast as a short one?
// The 512-point kernel is ~1000 instructions (7-9 KB) of straight-line code per frame set and every wave
// streams all of it once per iteration.  Same number of v_fma_f32 (VOP3, 8 bytes) / v_fmac_f32 (VOP2, 4
// bytes) executed per wave, body sizes from 64 B to 64 KB, 4 / 6 / 8 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_ifetch.hip -o scratch/ubif
#include <hip/hip_runtime.h>
#include <cstdio>

extern __shared__ char dyn_lds[];
constexpr long kTotal = 1 << 18;  // instructions per wave

// BODY = instructions per loop iteration (multiple of 8); VOP2 = 4-byte encodings
template <int BODY, bool VOP2>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0 * 0.5f, b1 = a1 * 0.5f;
  if (seed == 77.0f) dyn_lds[threadIdx.x] = 1;
#pragma unroll 1
  for (long it = 0; it < kTotal / BODY; ++it) {
    if (VOP2)
      asm volatile(".rept %10\n v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                   "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n .endr"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(b0), "v"(b1), "n"(BODY / 8));
    else
      asm volatile(".rept %10\n v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                   "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n .endr"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(b0), "v"(b1), "n"(BODY / 8));
  }
  float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (r == 12345.678f) out[0] = r;
}

template <int BODY, bool VOP2>
void run(float* out) {
  printf("body %6d instr (%6d B) %s:", BODY, BODY * (VOP2 ? 4 : 8), VOP2 ? "v_fmac (4 B)" : "v_fma  (8 B)");
  for (int w : {1, 2, 4, 6, 8}) {
    const size_t lds = (160 * 1024 / w) & ~size_t(255);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<BODY, VOP2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BODY, VOP2>), dim3(256 * w), dim3(256), lds, 0, out, 1.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<BODY, VOP2>), dim3(256 * w), dim3(256), lds, 0, out, 1.0f);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("  w%d %.3f ns", w, best * 1e6 / (double(w) * kTotal));
  }
  printf("   per wave-instruction and SIMD\n");
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  run<8, false>(out);
  run<128, false>(out);
  run<512, false>(out);
  run<1024, false>(out);
  run<2048, false>(out);
  run<3072, false>(out);
  run<8, true>(out);
  run<1024, true>(out);
  run<2048, true>(out);
  run<4096, true>(out);
  run<6144, true>(out);
  return 0;
}
