// µbench: 16-point FFT in registers, scalar f32 vs packed f32 (v_pk_*), 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }
__device__ __forceinline__ void dft4(float2 a0, float2 a1, float2 a2, float2 a3, float2& o0, float2& o1, float2& o2, float2& o3) {
  const float2 s0 = cadd(a0, a2), s1 = csub(a0, a2), s2 = cadd(a1, a3), s3 = mul_mi(csub(a1, a3));
  o0 = cadd(s0, s2); o1 = cadd(s1, s3); o2 = csub(s0, s2); o3 = csub(s1, s3);
}
__device__ __forceinline__ void fft16(float2 (&v)[16]) {
  constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, r2 = 0.70710678118654752440f;
  float2 t[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) dft4(v[q], v[q + 4], v[q + 8], v[q + 12], t[0][q], t[1][q], t[2][q], t[3][q]);
  t[1][1] = cmul(t[1][1], make_float2(c1, -s1));
  t[1][2] = make_float2((t[1][2].x + t[1][2].y) * r2, (t[1][2].y - t[1][2].x) * r2);
  t[1][3] = cmul(t[1][3], make_float2(s1, -c1));
  t[2][1] = make_float2((t[2][1].x + t[2][1].y) * r2, (t[2][1].y - t[2][1].x) * r2);
  t[2][2] = mul_mi(t[2][2]);
  t[2][3] = make_float2((t[2][3].y - t[2][3].x) * r2, -(t[2][3].x + t[2][3].y) * r2);
  t[3][1] = cmul(t[3][1], make_float2(s1, -c1));
  t[3][2] = make_float2((t[3][2].y - t[3][2].x) * r2, -(t[3][2].x + t[3][2].y) * r2);
  t[3][3] = cmul(t[3][3], make_float2(-c1, s1));
#pragma unroll
  for (int m = 0; m < 4; ++m) dft4(t[m][0], t[m][1], t[m][2], t[m][3], v[m], v[4 + m], v[8 + m], v[12 + m]);
}
// ---- packed version: complex = v2f (x, y) in an aligned register pair --------------------------------
// a + (-i) b = (a.x + b.y, a.y - b.x): one v_pk_add_f32 with op_sel swapping the halves of b and neg_hi
__device__ __forceinline__ v2f padd(v2f a, v2f b) { return a + b; }
__device__ __forceinline__ v2f psub(v2f a, v2f b) { return a - b; }
__device__ __forceinline__ v2f padd_mi(v2f a, v2f b) {  // a + (-i) b
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ v2f psub_mi(v2f a, v2f b) {  // a - (-i) b = (a.x - b.y, a.y + b.x)
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// a * w: t = (a.x, a.x) * (w.x, w.y); r = (a.y, a.y) * (-w.y, w.x) + t
__device__ __forceinline__ v2f pcmul(v2f a, v2f w) {
  v2f t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
__device__ __forceinline__ void pdft4(v2f a0, v2f a1, v2f a2, v2f a3, v2f& o0, v2f& o1, v2f& o2, v2f& o3) {
  const v2f s0 = a0 + a2, s1 = a0 - a2, s2 = a1 + a3, d = a1 - a3;
  o0 = s0 + s2; o2 = s0 - s2; o1 = padd_mi(s1, d); o3 = psub_mi(s1, d);
}
__device__ __forceinline__ void pfft16(v2f (&v)[16], const v2f* tw) {  // tw: W16^1, W16^2, W16^3, W16^6, W16^9 in registers
  v2f t[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) pdft4(v[q], v[q + 4], v[q + 8], v[q + 12], t[0][q], t[1][q], t[2][q], t[3][q]);
  t[1][1] = pcmul(t[1][1], tw[0]); t[1][2] = pcmul(t[1][2], tw[1]); t[1][3] = pcmul(t[1][3], tw[2]);
  t[2][1] = pcmul(t[2][1], tw[1]);
  { v2f z = {0.f, 0.f}; t[2][2] = padd_mi(z, t[2][2]); }
  t[2][3] = pcmul(t[2][3], tw[3]);
  t[3][1] = pcmul(t[3][1], tw[2]); t[3][2] = pcmul(t[3][2], tw[3]); t[3][3] = pcmul(t[3][3], tw[4]);
#pragma unroll
  for (int m = 0; m < 4; ++m) pdft4(t[m][0], t[m][1], t[m][2], t[m][3], v[m], v[4 + m], v[8 + m], v[12 + m]);
}
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters, float seed) {
  if (MODE == 0) {
    float2 z[16];
    for (int i = 0; i < 16; ++i) z[i] = make_float2(seed + threadIdx.x * 0.001f + i, seed - i * 0.5f);
    for (int it = 0; it < iters; ++it) {
      fft16(z);
      for (int i = 0; i < 16; ++i) { z[i].x *= 0.25f; z[i].y *= 0.25f; }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += z[i].x + z[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    v2f z[16];
    for (int i = 0; i < 16; ++i) z[i] = v2f{seed + threadIdx.x * 0.001f + i, seed - i * 0.5f};
    const v2f tw[5] = {{0.92387953f, -0.38268343f}, {0.70710678f, -0.70710678f}, {0.38268343f, -0.92387953f},
                       {-0.70710678f, -0.70710678f}, {-0.92387953f, 0.38268343f}};
    const v2f sc = {0.25f, 0.25f};
    for (int it = 0; it < iters; ++it) {
      pfft16(z, tw);
      for (int i = 0; i < 16; ++i) z[i] *= sc;
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += z[i].x + z[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}
template <int MODE> float run(float* out, int iters, int blocks) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  const int blocks = 256 * 4, iters = 2000;  // 4 blocks of 4 waves per CU = 4 waves per SIMD
  float* out; hipMalloc(&out, blocks * 256 * 4);
  std::vector<float> a(blocks * 256), b(blocks * 256);
  float m0 = run<0>(out, iters, blocks); hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost);
  float m1 = run<1>(out, iters, blocks); hipMemcpy(b.data(), out, b.size() * 4, hipMemcpyDeviceToHost);
  double maxd = 0; for (size_t i = 0; i < a.size(); ++i) maxd = fmax(maxd, fabs(a[i] - b[i]) / fmax(1.0, fabs(a[i])));
  const double ffts = double(blocks) * 4 /*waves*/ * iters;
  printf("scalar fft16: %.3f ms  %.1f clk per wave-fft per SIMD @2.2GHz\n", m0, m0 * 1e-3 * 2.2e9 / (ffts / 1024));
  printf("packed fft16: %.3f ms  %.1f clk per wave-fft per SIMD @2.2GHz\n", m1, m1 * 1e-3 * 2.2e9 / (ffts / 1024));
  printf("max rel diff scalar vs packed %.3g (first %g %g)\n", maxd, a[0], b[0]);
  return 0;
}
