// µbench (round 3): what a vector-memory load instruction costs on the texture path (TA / L1) of a CU in
// the access pattern of the 512-point kernel: a wave = 4 frames x 16 lanes, frames 320 bytes apart,
// lane l of a frame reads 4 bytes at 4 l + 64 j (13 instructions per frame set), the data streamed once
// from HBM with the 60 % overlap between frames served by L1 / L2.  Against: the same bytes as 8-byte typed
// loads, and as 16-byte loads of contiguous lanes (4 instructions per set).
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_vmem.hip -o scratch/ubvm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ f32x2 tbuf2(i32x4, int, int, int, int) __asm("llvm.amdgcn.raw.tbuffer.load.v2f32");
__device__ float tbuf1(i32x4, int, int, int, int) __asm("llvm.amdgcn.raw.tbuffer.load.f32");
__device__ f32x4 tbuf4(i32x4, int, int, int, int) __asm("llvm.amdgcn.raw.tbuffer.load.v4f32");

// MODE 0: 13 x global_load_dword; 1: 13 x typed xy; 2: 13 x (typed xy + typed x); 3: 13 x typed xyzw;
// 4: 4 x global_load_dwordx4 (lane l of a frame: 16 bytes at 16 l + 256 j); 5: 13 x global_load_dwordx2
template <int MODE>
__global__ __launch_bounds__(1024) void k(const short* __restrict__ wave, long long total_frames, float* out) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l = lane & 15, q = lane >> 4;
  const long long n_sets = total_frames / 4, stride = (long long)gridDim.x * 16;
  float acc = 0.0f;
  for (long long set = (long long)blockIdx.x * 16 + wid; set < n_sets; set += stride) {
    const long long st = (set * 4 + q) * 160;  // first sample of the lane's frame
    if (MODE == 0) {
      const int* p = reinterpret_cast<const int*>(wave + st) + l;
      int r[13];
#pragma unroll
      for (int j = 0; j < 13; ++j) r[j] = p[16 * j];
#pragma unroll
      for (int j = 0; j < 13; ++j) acc += __int_as_float(r[j] & 0x3fffffff);
    } else if (MODE == 5) {
      const int2* p = reinterpret_cast<const int2*>(wave + st - 2 + 2 * l);
      int2 r[13];
#pragma unroll
      for (int j = 0; j < 13; ++j) r[j] = *reinterpret_cast<const int2*>(reinterpret_cast<const char*>(p) + 64 * j);
#pragma unroll
      for (int j = 0; j < 13; ++j) acc += __int_as_float((r[j].x ^ r[j].y) & 0x3fffffff);
    } else if (MODE == 4) {
      const int4* p = reinterpret_cast<const int4*>(wave + st) + l;
      int4 r[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j] = p[16 * j];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += __int_as_float((r[j].x ^ r[j].y ^ r[j].z ^ r[j].w) & 0x3fffffff);
    } else {
      const long long st0 = set * 4 * 160;
      const unsigned long long base = reinterpret_cast<unsigned long long>(wave + st0);
      i32x4 rs;
      rs[0] = __builtin_amdgcn_readfirstlane((int)base);
      rs[1] = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffff));
      rs[2] = 1 << 30;
      rs[3] = 0x00020000;
      const int voff = (int)(st - st0) * 2 + 4 * l;
      if (MODE == 1) {
        f32x2 r[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) r[j] = tbuf2(rs, voff + 64 * j, 0, 5 | (3 << 4), 0);
#pragma unroll
        for (int j = 0; j < 13; ++j) acc += r[j][0] + r[j][1];
      } else if (MODE == 2) {
        f32x2 r[13];
        float s[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) {
          r[j] = tbuf2(rs, voff + 64 * j, 0, 5 | (3 << 4), 0);
          s[j] = tbuf1(rs, (j == 0 && l == 0 ? voff : voff - 2) + 64 * j, 0, 2 | (3 << 4), 0);
        }
#pragma unroll
        for (int j = 0; j < 13; ++j) acc += r[j][0] + r[j][1] + s[j];
      } else {
        f32x4 r[13];
#pragma unroll
        for (int j = 0; j < 13; ++j) r[j] = tbuf4(rs, (j == 0 && l == 0 ? voff : voff - 4) + 64 * j, 0, 12 | (3 << 4), 0);
#pragma unroll
        for (int j = 0; j < 13; ++j) acc += r[j][0] + r[j][1] + r[j][2] + r[j][3];
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
void run(const char* name, const short* w, long long frames, float* out, int n_instr) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wgs : {256, 512}) {
    hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(1024), 0, 0, w, frames, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(1024), 0, 0, w, frames, out);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double sets_per_cu = frames / 4.0 / 256.0;
    printf("%-44s %d waves/CU: %.3f ms = %.2f TB/s of new samples, %.1f ns per load instruction and CU\n", name,
           wgs / 256 * 16, best, frames * 320.0 / best * 1e-9, best * 1e6 / (sets_per_cu * n_instr));
  }
}

int main() {
  const long long frames = 2980000;
  const long long samples = frames * 160 + 4096;
  short* w;
  hipMalloc(&w, samples * 2);
  hipMemset(w, 1, samples * 2);
  w += 1024;  // (the shifted 8-byte loads of the first frame start two samples lower)
  float* out;
  hipMalloc(&out, 4);
  run<0>("13 x global_load_dword (round-2 kernel)", w, frames, out, 13);
  run<1>("13 x typed 16_16", w, frames, out, 13);
  run<2>("13 x (typed 16_16 + typed 16)", w, frames, out, 26);
  run<3>("13 x typed 16_16_16_16", w, frames, out, 13);
  run<5>("13 x global_load_dwordx2 (2-byte aligned)", w, frames, out, 13);
  run<4>("4 x global_load_dwordx4, contiguous lanes", w, frames, out, 4);
  return 0;
}
