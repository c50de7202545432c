// Issue cost of the candidate evaluation of the pitch tracker's Viterbi search (cost, compare, select,
// minimum) in its round-2 form and in the table / scalar-ordinal forms, per wave instruction and per
// candidate, on a full chip (every SIMD busy with `waves` waves).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_argmin.hip -o scratch/ubam && scratch/ubam
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int kIters = 8192;

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed, float factor) {
  float b = 1.0e30f + seed, bd = 0.0f, d = seed, f0 = seed * 0.5f, f1 = seed * 0.25f, f2 = seed * 0.125f,
        f3 = seed * 0.0625f;
  int ord = 0, bit = 0, vit = 0;
  float prev = 0.0f;
  for (int it = 0; it < kIters; ++it) {
    if (OP == 0) {  // round 2: 4 candidates, 7 VALU each (compiler's choice of encodings)
      const float d0 = d + 1.0f, d1 = d + 2.0f, d2 = d + 3.0f, d3 = d + 4.0f;
      const float c0 = __fadd_rn(__fmul_rn(d0 * d0, factor), f0), c1 = __fadd_rn(__fmul_rn(d1 * d1, factor), f1);
      const float c2 = __fadd_rn(__fmul_rn(d2 * d2, factor), f2), c3 = __fadd_rn(__fmul_rn(d3 * d3, factor), f3);
      bd = c0 < b ? d0 : bd; b = fminf(b, c0);
      bd = c1 < b ? d1 : bd; b = fminf(b, c1);
      bd = c2 < b ? d2 : bd; b = fminf(b, c2);
      bd = c3 < b ? d3 : bd; b = fminf(b, c3);
      d = d3;
      asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
    }
    if (OP == 1) {  // table form: add, v_cmp_ge -> vcc, v_cndmask_e32 (scalar ordinal), v_min: 4 VALU
      asm volatile(
          "v_mov_b32 %10, %0\n"
          "v_add_f32 %2, %2, %6\n v_cmp_ge_f32 vcc, %2, %0\n v_cndmask_b32 %1, 0, %1, vcc\n v_min_f32 %0, %0, %2\n"
          "v_add_f32 %3, %3, %7\n v_cmp_ge_f32 vcc, %3, %0\n v_cndmask_b32 %1, 1, %1, vcc\n v_min_f32 %0, %0, %3\n"
          "v_add_f32 %4, %4, %8\n v_cmp_ge_f32 vcc, %4, %0\n v_cndmask_b32 %1, 2, %1, vcc\n v_min_f32 %0, %0, %4\n"
          "v_add_f32 %5, %5, %9\n v_cmp_ge_f32 vcc, %5, %0\n v_cndmask_b32 %1, 3, %1, vcc\n v_min_f32 %0, %0, %5\n"
          "v_cmp_ge_f32 vcc, %0, %10\n v_cndmask_b32 %11, %12, %11, vcc\n v_add_u32 %12, 4, %12\n"
          : "+v"(b), "+v"(ord), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)
          : "v"(d), "v"(bd), "v"(d), "v"(bd), "v"(prev), "v"(bit), "v"(vit)
          : "vcc");
    }
    if (OP == 2) {  // the same with e64 compares into a scalar pair and e64 selects
      asm volatile(
          "v_add_f32 %2, %2, %6\n v_cmp_lt_f32 s[20:21], %2, %0\n v_cndmask_b32 %1, %1, %6, s[20:21]\n v_min_f32 %0, %0, %2\n"
          "v_add_f32 %3, %3, %7\n v_cmp_lt_f32 s[22:23], %3, %0\n v_cndmask_b32 %1, %1, %7, s[22:23]\n v_min_f32 %0, %0, %3\n"
          "v_add_f32 %4, %4, %8\n v_cmp_lt_f32 s[20:21], %4, %0\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_min_f32 %0, %0, %4\n"
          "v_add_f32 %5, %5, %9\n v_cmp_lt_f32 s[22:23], %5, %0\n v_cndmask_b32 %1, %1, %9, s[22:23]\n v_min_f32 %0, %0, %5\n"
          : "+v"(b), "+v"(ord), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)
          : "v"(d), "v"(bd), "v"(d), "v"(bd)
          : "s20", "s21", "s22", "s23");
    }
    if (OP == 3) {  // minimum only (block form): add + half a v_min3
      asm volatile(
          "v_add_f32 %1, %1, %5\n v_add_f32 %2, %2, %6\n v_min3_f32 %0, %0, %1, %2\n"
          "v_add_f32 %3, %3, %5\n v_add_f32 %4, %4, %6\n v_min3_f32 %0, %0, %3, %4\n"
          : "+v"(b), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)
          : "v"(d), "v"(bd));
    }
    if (OP == 4) {  // 8 x v_cndmask_b32_e32 on vcc alone
      asm volatile(
          "v_cndmask_b32 %0, %4, %0, vcc\n v_cndmask_b32 %1, %4, %1, vcc\n v_cndmask_b32 %2, %4, %2, vcc\n"
          "v_cndmask_b32 %3, %4, %3, vcc\n v_cndmask_b32 %0, %5, %0, vcc\n v_cndmask_b32 %1, %5, %1, vcc\n"
          "v_cndmask_b32 %2, %5, %2, vcc\n v_cndmask_b32 %3, %5, %3, vcc\n"
          : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)
          : "v"(d), "v"(bd)
          : "vcc");
    }
    if (OP == 5) {  // 8 x v_cmp_ge_f32_e32 -> vcc
      asm volatile(
          "v_cmp_ge_f32 vcc, %0, %1\n v_cmp_ge_f32 vcc, %1, %2\n v_cmp_ge_f32 vcc, %2, %3\n v_cmp_ge_f32 vcc, %3, %0\n"
          "v_cmp_ge_f32 vcc, %0, %2\n v_cmp_ge_f32 vcc, %1, %3\n v_cmp_ge_f32 vcc, %2, %0\n v_cmp_ge_f32 vcc, %3, %1\n"
          :
          : "v"(f0), "v"(f1), "v"(f2), "v"(f3)
          : "vcc");
    }
  }
  if (b + bd + d + f0 + f1 + f2 + f3 + ord + bit + vit + prev == 12345.678f) out[0] = b;
}

template <int OP>
static void run(const char* name, int per_iter_instr, int per_iter_cands, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int waves = 4; waves <= 8; waves += 4) {
    const int blocks = 256 * waves;  // 256-thread workgroups: one wave per SIMD each
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 3.0f, 2.5e-6f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms < best) best = ms;
    }
    const double per_simd_ns = best * 1e6 / (static_cast<double>(kIters) * waves);
    printf("%-46s %d waves/SIMD: %.2f ns per iteration per wave", name, waves, per_simd_ns);
    if (per_iter_instr) printf(", %.2f ns per VALU instruction", per_simd_ns / per_iter_instr);
    if (per_iter_cands) printf(", %.2f ns per candidate", per_simd_ns / per_iter_cands);
    printf("\n");
  }
}

int main() {
  float* out;
  hipMalloc(&out, 64);
  run<0>("round-2 form (7 VALU per candidate)", 28, 4, out);
  run<1>("table form, vcc + e32 select, block ordinal", 20, 4, out);
  run<2>("table form, scalar-pair compare, e64 select", 16, 4, out);
  run<3>("minimum only: add + v_min3 per two", 6, 4, out);
  run<4>("v_cndmask_b32_e32 x 8", 8, 0, out);
  run<5>("v_cmp_ge_f32_e32 x 8", 8, 0, out);
  return 0;
}
