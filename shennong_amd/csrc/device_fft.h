// Device helpers shared by the register-resident FFT kernels (kernels_fbank512.hip, kernels_fbank2048.hip):
// LDS access wrappers, DPP reductions, the counter-based dither generator and the 16-point register FFT.
// Included inside namespace snf; everything is static to the including translation unit.
#ifndef SNF_DEVICE_FFT_H_
#define SNF_DEVICE_FFT_H_

#include <float.h>
#include <hip/hip_runtime.h>

#include <cstdint>

namespace snf {

namespace {

// ln(x) for x >= FLT_EPSILON via the hardware log2 (1 ulp): 2 instructions instead of ~15
__device__ __forceinline__ float fast_log(float x) {
  return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
}
// max(x, FLT_EPSILON) for finite x in one instruction (fmaxf costs a canonicalising v_max first)
__device__ __forceinline__ float floor_eps(float x) {
  return __builtin_amdgcn_fmed3f(x, FLT_EPSILON, FLT_MAX);
}
// acc += f * (value of src in lane + SHIFT of the same 16-lane row, 0 beyond the row): v_fmac_f32 with
// the DPP row shift on its first source (the s_nop covers the VALU-write -> DPP-read hazard, which
// the compiler does not pad inside an asm statement)
template <int SHIFT>
__device__ __forceinline__ void fmac_row_shl(float& acc, float src, float f) {
  asm volatile("s_nop 1\n v_fmac_f32_dpp %0, %1, %2 row_shl:%3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
               : "+v"(acc) : "v"(src), "v"(f), "n"(SHIFT));
}

// On gfx950 ds_read2_b64 runs at half the bandwidth of ds_read_b64 / ds_read_b128 (MI355X_MICROARCH
// LDS table), and hipcc merges adjacent 8-byte LDS loads into it.  The helpers below issue single reads
// through inline asm.  A batch of reads AND the s_waitcnt that completes them form ONE asm statement:
// the compiler treats an asm output as valid the moment the statement ends, so with the wait in a
// later statement it is free to copy (v_mov) an output register before its data has landed - the
// upper lanes of a wave are served last by the LDS pipe, which made exactly the fourth frame of a
// wave read stale values on boxes where the timing lined up.  Outputs are early-clobber: the address
// register is still needed by the later reads of the batch.
typedef __attribute__((address_space(3))) const void* lds_cptr;
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_cptr)p));
}
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// dst[i] = the float2 at byte offset 8 i from `base`, i < 16 (one row of the transpose tile)
__device__ __forceinline__ void read16_b64(const void* base, float2 (&d)[16]) {
  asm volatile(
      "ds_read_b64 %0, %16\n ds_read_b64 %1, %16 offset:8\n ds_read_b64 %2, %16 offset:16\n"
      "ds_read_b64 %3, %16 offset:24\n ds_read_b64 %4, %16 offset:32\n ds_read_b64 %5, %16 offset:40\n"
      "ds_read_b64 %6, %16 offset:48\n ds_read_b64 %7, %16 offset:56\n ds_read_b64 %8, %16 offset:64\n"
      "ds_read_b64 %9, %16 offset:72\n ds_read_b64 %10, %16 offset:80\n ds_read_b64 %11, %16 offset:88\n"
      "ds_read_b64 %12, %16 offset:96\n ds_read_b64 %13, %16 offset:104\n ds_read_b64 %14, %16 offset:112\n"
      "ds_read_b64 %15, %16 offset:120\n s_waitcnt lgkmcnt(0)"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]),
        "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11]), "=&v"(d[12]), "=&v"(d[13]),
        "=&v"(d[14]), "=&v"(d[15])
      : "v"(lds_addr(base))
      : "memory");
}
// the 8 float4 of a twiddle row at `tw` (16-byte aligned) and the 16 float2 of a tile row at `base` in one
// batch: the compiler turns plain 16-byte loads whose first half is unused into ds_read2_b64 (half rate)
__device__ __forceinline__ void read_tw8_row16(const void* tw, const void* base, float4 (&q)[8], float2 (&d)[16]) {
  asm volatile(
      "ds_read_b128 %0, %24\n ds_read_b128 %1, %24 offset:16\n ds_read_b128 %2, %24 offset:32\n"
      "ds_read_b128 %3, %24 offset:48\n ds_read_b128 %4, %24 offset:64\n ds_read_b128 %5, %24 offset:80\n"
      "ds_read_b128 %6, %24 offset:96\n ds_read_b128 %7, %24 offset:112\n"
      "ds_read_b64 %8, %25\n ds_read_b64 %9, %25 offset:8\n ds_read_b64 %10, %25 offset:16\n"
      "ds_read_b64 %11, %25 offset:24\n ds_read_b64 %12, %25 offset:32\n ds_read_b64 %13, %25 offset:40\n"
      "ds_read_b64 %14, %25 offset:48\n ds_read_b64 %15, %25 offset:56\n ds_read_b64 %16, %25 offset:64\n"
      "ds_read_b64 %17, %25 offset:72\n ds_read_b64 %18, %25 offset:80\n ds_read_b64 %19, %25 offset:88\n"
      "ds_read_b64 %20, %25 offset:96\n ds_read_b64 %21, %25 offset:104\n ds_read_b64 %22, %25 offset:112\n"
      "ds_read_b64 %23, %25 offset:120\n s_waitcnt lgkmcnt(0)"
      : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]),
        "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]),
        "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11]), "=&v"(d[12]), "=&v"(d[13]),
        "=&v"(d[14]), "=&v"(d[15])
      : "v"(lds_addr(tw)), "v"(lds_addr(base))
      : "memory");
}
// dst[i] = the float2 at byte offset 128 (7 - i) from `base`, i < 8 (partner rows, reversed)
__device__ __forceinline__ void read8_b64_rev128(const void* base, float2 (&d)[8]) {
  asm volatile(
      "ds_read_b64 %0, %8 offset:896\n ds_read_b64 %1, %8 offset:768\n ds_read_b64 %2, %8 offset:640\n"
      "ds_read_b64 %3, %8 offset:512\n ds_read_b64 %4, %8 offset:384\n ds_read_b64 %5, %8 offset:256\n"
      "ds_read_b64 %6, %8 offset:128\n ds_read_b64 %7, %8\n s_waitcnt lgkmcnt(0)"
      : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]),
        "=&v"(d[7])
      : "v"(lds_addr(base))
      : "memory");
}
// N float4 (= 2 N complex) contiguous from `base`: plain 16-byte LDS loads (the compiler emits
// ds_read_b128 - there is no slower merged form for 128-bit reads - and places the waits itself)
template <int N>
__device__ __forceinline__ void read_quads(const void* base, float4 (&dst)[N]) {
  const float4* __restrict__ q =
      reinterpret_cast<const float4*>(__builtin_assume_aligned(base, 16));  // (rows are 16-byte aligned)
#pragma unroll
  for (int i = 0; i < N; ++i) dst[i] = q[i];
}
// The same with volatile accesses: a quad whose first half is never used (a twiddle row that starts with
// W^0 = 1) is otherwise re-cut by the compiler into 8-byte pieces and merged into ds_read2_b64, which runs at
// half the bandwidth of ds_read_b128 and, 8-byte aligned, is where the 512-point kernel's bank conflicts
// came from (round 4); a volatile 16-byte access stays one ds_read_b128 and is still waited for at first use
template <int N>
__device__ __forceinline__ void read_quads_whole(const void* base, float4 (&dst)[N]) {
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  // (an explicit LDS pointer: a volatile access through a generic pointer becomes a flat load)
  typedef const volatile f32x4v __attribute__((address_space(3))) * lds_quad_ptr;
  lds_quad_ptr q = (lds_quad_ptr)(__builtin_assume_aligned(base, 16));
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const f32x4v v = q[i];
    dst[i] = make_float4(v.x, v.y, v.z, v.w);
  }
}
// counter-based N(0,1) pair for Kaldi's per-frame dither (statistical stand-in for RandGauss(), which
// draws from C rand() and is not reproducible): murmur-style 32-bit finalisers + Box-Muller on the
// hardware log2 / sqrt / sin / cos (v_sin_f32 and v_cos_f32 take revolutions)
__device__ __forceinline__ unsigned fmix32(unsigned h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ float2 gauss_pair(unsigned key_lo, unsigned key_hi, unsigned n,
                                             float k = -1.38629436111989f) {
  // Round 5 (the reference's default costs 1.49 x the dither-0 kernel): 29 issue slots per
  // pair of normals instead of 44.
  //  * one multiply-fold per pair instead of a murmur finaliser: the keys of a frame are finaliser outputs
  //    already (callers), the counter walks a Weyl sequence under them; x ^ x >> 15, then the high and the low
  //    word of the 64-bit product with an odd constant xored (v_mad_u64_u32 + 3 full-rate instructions; the
  //    finaliser was 2 v_mul_lo_u32 + 6);
  //  * the two uniforms are made in the mantissa of a float in [1, 2) - shift + or, no conversion, no scaling:
  //    the radius' uniform is 2 - f in (0, 1] (23 bits: the top 23 of the word, so the radius reaches 5.6 sigma),
  //    the angle goes to v_sin_f32 / v_cos_f32 as it is (they take revolutions and are periodic: sin 2 pi f =
  //    sin 2 pi (f - 1); 16 bits = 65 536 steps, the low 16 of the word);
  //  * Box-Muller itself stays: log2, sqrt, sin, cos on the transcendental unit.
  //  * the amplitude rides on the radius: `k` = -2 ln 2 x dither^2 (dither_scale), so that a sample takes its
  //    noise with one fused multiply-add.
  unsigned x = (key_lo + n * 0x9E3779B1u) ^ key_hi;
  x ^= x >> 15;
  const unsigned long long prod = static_cast<unsigned long long>(x) * 0x85EBCA6Bu;
  const unsigned h = static_cast<unsigned>(prod) ^ static_cast<unsigned>(prod >> 32);
  const float fr = __builtin_bit_cast(float, (h >> 9) | 0x3f800000u);               // [1, 2)
  const float fa = __builtin_bit_cast(float, ((h << 7) & 0x007fff80u) | 0x3f800000u);  // [1, 2), 16 bits
  const float r = __builtin_amdgcn_sqrtf(k * __builtin_amdgcn_logf(2.0f - fr));
  return make_float2(r * __builtin_amdgcn_cosf(fa), r * __builtin_amdgcn_sinf(fa));
}
// k of gauss_pair for N(0, dither^2)
__device__ __forceinline__ float dither_scale(float dither) { return -1.38629436111989f * dither * dither; }
// xe, xo += N(0, dither^2): the form every 512-point / long-frame kernel uses (bit-identical across them)
__device__ __forceinline__ void add_dither_pair(unsigned key_lo, unsigned key_hi, unsigned n, float k, float& xe,
                                                float& xo) {
  unsigned x = (key_lo + n * 0x9E3779B1u) ^ key_hi;
  x ^= x >> 15;
  const unsigned long long prod = static_cast<unsigned long long>(x) * 0x85EBCA6Bu;
  const unsigned h = static_cast<unsigned>(prod) ^ static_cast<unsigned>(prod >> 32);
  const float fr = __builtin_bit_cast(float, (h >> 9) | 0x3f800000u);
  const float fa = __builtin_bit_cast(float, ((h << 7) & 0x007fff80u) | 0x3f800000u);
  const float r = __builtin_amdgcn_sqrtf(k * __builtin_amdgcn_logf(2.0f - fr));
  xe = __builtin_fmaf(r, __builtin_amdgcn_cosf(fa), xe);
  xo = __builtin_fmaf(r, __builtin_amdgcn_sinf(fa), xo);
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// The same ordering point for a tile that only ONE wave touches: the LDS executes the instructions of a wave
// in the order they were issued, so a read behind a write of the same wave needs no s_waitcnt - only the
// compiler has to keep the two in program order (wavefront-scope fences emit no instruction, where the
// workgroup-scope ones above make the wave sit until every LDS store is acknowledged)
__device__ __forceinline__ void wave_lds_order() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the 16 lanes of a DPP row (= one frame), result in every lane of the row
template <int CTRL>
__device__ __forceinline__ float dpp_row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL,
                                                               0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_row_ror_d(double v) {
  const long long bits = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, static_cast<int>(bits), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, static_cast<int>(bits >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, (static_cast<long long>(hi) << 32) |
                                        static_cast<long long>(static_cast<unsigned>(lo)));
}
// v_mov_b32_dpp: lanes whose source lane does not exist keep `old` (BOUND = false) or read 0
template <int CTRL, bool BOUND>
__device__ __forceinline__ float dpp_mov(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old),
                                                               __builtin_bit_cast(int, v), CTRL, 0xf,
                                                               0xf, BOUND));
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short short2v __attribute__((ext_vector_type(2)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // output rows are 4-byte aligned
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_row_ror<0x128>(v);  // row_ror:8
  v += dpp_row_ror<0x124>(v);  // row_ror:4
  v += dpp_row_ror<0x122>(v);  // row_ror:2
  v += dpp_row_ror<0x121>(v);  // row_ror:1
  return v;
}

__device__ __forceinline__ int64_t find_utt(const int64_t* __restrict__ offsets, int64_t n,
                                            int64_t g) {
  int64_t lo = 0, hi = n;
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * (-i)
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// 4-point forward DFT
__device__ __forceinline__ void dft4(float2 a0, float2 a1, float2 a2, float2 a3, float2& o0,
                                     float2& o1, float2& o2, float2& o3) {
  const float2 s0 = cadd(a0, a2), s1 = csub(a0, a2), s2 = cadd(a1, a3), s3 = mul_mi(csub(a1, a3));
  o0 = cadd(s0, s2);
  o1 = cadd(s1, s3);
  o2 = csub(s0, s2);
  o3 = csub(s1, s3);
}

// 16-point forward FFT in registers, natural order in and out (radix-4 DIF x radix-4)
__device__ __forceinline__ void fft16(float2 (&v)[16]) {
  constexpr float c1 = 0.92387953251128675613f;  // cos(pi/8)
  constexpr float s1 = 0.38268343236508977173f;  // sin(pi/8)
  constexpr float r2 = 0.70710678118654752440f;  // sqrt(1/2)
  float2 t[4][4];  // t[m][q]
#pragma unroll
  for (int q = 0; q < 4; ++q) dft4(v[q], v[q + 4], v[q + 8], v[q + 12], t[0][q], t[1][q], t[2][q], t[3][q]);
  // twiddles W16^(q m)
  t[1][1] = cmul(t[1][1], make_float2(c1, -s1));                                   // W^1
  t[1][2] = make_float2((t[1][2].x + t[1][2].y) * r2, (t[1][2].y - t[1][2].x) * r2);  // W^2
  t[1][3] = cmul(t[1][3], make_float2(s1, -c1));                                   // W^3
  t[2][1] = make_float2((t[2][1].x + t[2][1].y) * r2, (t[2][1].y - t[2][1].x) * r2);  // W^2
  t[2][2] = mul_mi(t[2][2]);                                                       // W^4
  t[2][3] = make_float2((t[2][3].y - t[2][3].x) * r2, -(t[2][3].x + t[2][3].y) * r2); // W^6
  t[3][1] = cmul(t[3][1], make_float2(s1, -c1));                                   // W^3
  t[3][2] = make_float2((t[3][2].y - t[3][2].x) * r2, -(t[3][2].x + t[3][2].y) * r2); // W^6
  t[3][3] = cmul(t[3][3], make_float2(-c1, s1));                                   // W^9
#pragma unroll
  for (int m = 0; m < 4; ++m) dft4(t[m][0], t[m][1], t[m][2], t[m][3], v[m], v[4 + m], v[8 + m], v[12 + m]);
}


// ---- the 512- / 256-point kernels' butterflies with their twiddles folded in (Linzer-Feig form, round 4) ----
// A twiddle W = c (1 + i t) costs two fused multiply-adds for u = b (1 + i t) and its real scale c rides
// on the fused multiply-adds of the butterfly that consumes it: a0 +- c u.  Against complex multiply +
// add / subtract that is 6 instead of 8 vector instructions per twiddled radix-2 pair.
__device__ __forceinline__ float2 lf_u(float2 b, float t) {  // b (1 + i t)
  return make_float2(__builtin_fmaf(-t, b.y, b.x), __builtin_fmaf(t, b.x, b.y));
}
__device__ __forceinline__ float2 lf_add(float2 a, float c, float2 u) {  // a + c u
  return make_float2(__builtin_fmaf(c, u.x, a.x), __builtin_fmaf(c, u.y, a.y));
}
// second radix-4 layer of the 16-point FFT with the twiddles W16^(q m) folded into the butterflies
__device__ __forceinline__ void fft16_layer2_lf(float2 (&t)[4][4], float2 (&v)[16]) {
  constexpr float c1 = 0.92387953251128675613f;  // cos(pi/8)
  constexpr float t1 = 0.41421356237309504880f;  // tan(pi/8)
  constexpr float r2 = 0.70710678118654752440f;  // sqrt(1/2)
  dft4(t[0][0], t[0][1], t[0][2], t[0][3], v[0], v[4], v[8], v[12]);
  {  // row 1: W^1 = c1 (1 - i t1), W^2 = r2 (1 - i), W^3 = -i c1 (1 + i t1)
    const float2 b0 = t[1][0], b1 = t[1][1], b2 = t[1][2], b3 = t[1][3];
    const float2 u2 = make_float2(b2.x + b2.y, b2.y - b2.x);
    const float2 s0 = lf_add(b0, r2, u2), s1 = lf_add(b0, -r2, u2);
    const float2 u1 = lf_u(b1, -t1), u3 = lf_u(b3, t1);
    const float2 e = make_float2(u1.x + u3.y, u1.y - u3.x);  // (a1 + a3) / c1
    const float2 f = make_float2(u1.x - u3.y, u1.y + u3.x);  // (a1 - a3) / c1
    v[1] = lf_add(s0, c1, e);
    v[9] = lf_add(s0, -c1, e);
    v[5] = make_float2(__builtin_fmaf(c1, f.y, s1.x), __builtin_fmaf(-c1, f.x, s1.y));
    v[13] = make_float2(__builtin_fmaf(-c1, f.y, s1.x), __builtin_fmaf(c1, f.x, s1.y));
  }
  {  // row 2: W^2 = r2 (1 - i), W^4 = -i, W^6 = -r2 (1 + i)
    const float2 b0 = t[2][0], b1 = t[2][1], b2 = t[2][2], b3 = t[2][3];
    const float2 s0 = make_float2(b0.x + b2.y, b0.y - b2.x), s1 = make_float2(b0.x - b2.y, b0.y + b2.x);
    const float2 u1 = make_float2(b1.x + b1.y, b1.y - b1.x), u3 = make_float2(b3.x - b3.y, b3.y + b3.x);
    const float2 e = csub(u1, u3), f = cadd(u1, u3);
    v[2] = lf_add(s0, r2, e);
    v[10] = lf_add(s0, -r2, e);
    v[6] = make_float2(__builtin_fmaf(r2, f.y, s1.x), __builtin_fmaf(-r2, f.x, s1.y));
    v[14] = make_float2(__builtin_fmaf(-r2, f.y, s1.x), __builtin_fmaf(r2, f.x, s1.y));
  }
  {  // row 3: W^3 = -i c1 (1 + i t1), W^6 = -r2 (1 + i), W^9 = -c1 (1 - i t1)
    const float2 b0 = t[3][0], b1 = t[3][1], b2 = t[3][2], b3 = t[3][3];
    const float2 u2 = make_float2(b2.x - b2.y, b2.y + b2.x);
    const float2 s0 = lf_add(b0, -r2, u2), s1 = lf_add(b0, r2, u2);
    const float2 u1 = lf_u(b1, t1), u3 = lf_u(b3, -t1);
    const float2 e = make_float2(u1.y - u3.x, u1.x + u3.y);  // (a1 + a3) / c1 = (e.x, -e.y)
    const float2 f = make_float2(u1.y + u3.x, u3.y - u1.x);  // (a1 - a3) / c1
    v[3] = make_float2(__builtin_fmaf(c1, e.x, s0.x), __builtin_fmaf(-c1, e.y, s0.y));
    v[11] = make_float2(__builtin_fmaf(-c1, e.x, s0.x), __builtin_fmaf(c1, e.y, s0.y));
    v[7] = make_float2(__builtin_fmaf(c1, f.y, s1.x), __builtin_fmaf(-c1, f.x, s1.y));
    v[15] = make_float2(__builtin_fmaf(-c1, f.y, s1.x), __builtin_fmaf(c1, f.x, s1.y));
  }
}
__device__ __forceinline__ void fft16_lf(float2 (&v)[16]) {
  float2 t[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) dft4(v[q], v[q + 4], v[q + 8], v[q + 12], t[0][q], t[1][q], t[2][q], t[3][q]);
  fft16_layer2_lf(t, v);
}
// The same transform when only the first NZ inputs can be non-zero (a 25 ms frame at 44.1 / 48 kHz fills 9 / 10
// of the 16 element rows of the 2048-point kernel: the rest is the zero padding of the power-of-two window).  The
// first radix-4 layer drops the terms that are zero by construction - an exact zero added or subtracted changes
// nothing but the sign of a zero - instead of adding them (IEEE addition of a constant 0.0f is not folded away).
template <int NZ>
__device__ __forceinline__ void fft16_lf_head(float2 (&v)[16]) {
  float2 t[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (q + 8 >= NZ) {          // a2 = a3 = 0: s0 = s1 = a0, s2 = a1, s3 = -i a1
      const float2 a0 = v[q], a1 = (q + 4 < NZ) ? v[q + 4] : make_float2(0.0f, 0.0f), m1 = mul_mi(a1);
      if (q + 4 < NZ) {
        t[0][q] = cadd(a0, a1);
        t[1][q] = cadd(a0, m1);
        t[2][q] = csub(a0, a1);
        t[3][q] = csub(a0, m1);
      } else {
        t[0][q] = t[1][q] = t[2][q] = t[3][q] = a0;
      }
    } else if (q + 12 >= NZ) {  // a3 = 0: s2 = a1, s3 = -i a1
      const float2 s0 = cadd(v[q], v[q + 8]), s1 = csub(v[q], v[q + 8]), s2 = v[q + 4], s3 = mul_mi(v[q + 4]);
      t[0][q] = cadd(s0, s2);
      t[1][q] = cadd(s1, s3);
      t[2][q] = csub(s0, s2);
      t[3][q] = csub(s1, s3);
    } else {
      dft4(v[q], v[q + 4], v[q + 8], v[q + 12], t[0][q], t[1][q], t[2][q], t[3][q]);
    }
  }
  fft16_layer2_lf(t, v);
}
// 16-point FFT of r[m] * W[m] with the input twiddles W[m] = c[m] (1 + i t[m]) given as (c, t) pairs
// (W[0] = 1): the first radix-4 layer consumes them in its butterflies.  An exact -i is stored as
// (2^-40, -2^40): the products are exact powers of two and the absorbed term is the one a true zero
// cosine would have removed.
__device__ __forceinline__ void fft16_twin(float2 (&v)[16], const float2 (&ct)[16]) {
  float2 t[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float2 u1 = lf_u(v[q + 4], ct[q + 4].y), u2 = lf_u(v[q + 8], ct[q + 8].y), u3 = lf_u(v[q + 12], ct[q + 12].y);
    float2 p0 = v[q];
    if (q != 0) {
      const float2 u0 = lf_u(v[q], ct[q].y);
      p0 = make_float2(ct[q].x * u0.x, ct[q].x * u0.y);
    }
    const float2 p1 = make_float2(ct[q + 4].x * u1.x, ct[q + 4].x * u1.y);
    const float2 s0 = lf_add(p0, ct[q + 8].x, u2), s1 = lf_add(p0, -ct[q + 8].x, u2);
    const float2 s2 = lf_add(p1, ct[q + 12].x, u3), d13 = lf_add(p1, -ct[q + 12].x, u3);
    t[0][q] = cadd(s0, s2);
    t[2][q] = csub(s0, s2);
    t[1][q] = make_float2(s1.x + d13.y, s1.y - d13.x);
    t[3][q] = make_float2(s1.x - d13.y, s1.y + d13.x);
  }
  fft16_layer2_lf(t, v);
}

}  // namespace

}  // namespace snf

#endif  // SNF_DEVICE_FFT_H_
