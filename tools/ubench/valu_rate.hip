// Micro-benchmark: issue rate of the VALU ops used by the on-chip LDPC decoder (gfx950).
// Each kernel runs N iterations of 8 independent instructions per wave; 4 waves/SIMD resident.
// Prints cycles per wave-instruction per SIMD (wall clock * clock / instructions issued per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
  for (int i = 0; i < iters; ++i) {
#define RR(n) r##n
    if constexpr (OP == 0) {
#define X(n) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(RR(n)) : "v"(a), "v"(b));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 1) {
#define X(n) asm volatile("v_min_f32 %0, %1, %0" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 2) {
#define X(n) asm volatile("v_med3_f32 %0, %1, %2, %0" : "+v"(RR(n)) : "v"(a), "v"(b));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 3) {
#define X(n) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 4) {
#define X(n) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(RR(n)) : "v"(a), "v"(b));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 5) {
#define X(n) asm volatile("v_add_u32 %0, %1, %0" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 6) {
#define X(n) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(RR(n)));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 7) {
#define X(n) asm volatile("v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %0, %1, %0, vcc" : "+v"(RR(n)) : "v"(a) : "vcc");
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 8) {
#define X(n) asm volatile("v_and_b32 %0, %1, %0" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 9) {
#define X(n) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 10) {
#define X(n) asm volatile("v_min_u32 %0, %1, %0" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 11) {
#define X(n) asm volatile("v_lshl_or_b32 %0, %1, 3, %0" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 12) {
#define X(n) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x78" : "+v"(RR(n)) : "v"(a), "v"(b));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 13) {
#define X(n) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(RR(n)) : "v"(a), "v"(b));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 14) {
#define X(n) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 15) {
#define X(n) asm volatile("v_max_u32 %0, %1, %0" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 16) {
      // compare into distinct SGPR pairs, selects afterwards (the way an unrolled check-node body issues them)
#define X(n) asm volatile("v_cmp_eq_f32 s[%c2:%c3], %1, %0" : : "v"(RR(n)), "v"(a), "n"(20 + 2 * n), "n"(21 + 2 * n) : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30","s31","s32","s33","s34","s35");
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 17) {
#define X(n) asm volatile("v_cndmask_b32 %0, %1, %0, s[%c2:%c3]" : "+v"(RR(n)) : "v"(a), "n"(20 + 2 * n), "n"(21 + 2 * n));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 18) {
#define X(n) asm volatile("v_exp_f32 %0, %0" : "+v"(RR(n)));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 19) {
#define X(n) asm volatile("v_log_f32 %0, %0" : "+v"(RR(n)));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 20) {
#define X(n) asm volatile("v_frexp_mant_f32 %0, %0" : "+v"(RR(n)));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 21) {
#define X(n) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 22) {
#define X(n) asm volatile("v_floor_f32 %0, %0" : "+v"(RR(n)));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 23) {
#define X(n) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(RR(n)));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 24) {
#define X(n) asm volatile("v_min3_f32 %0, %1, %2, %0" : "+v"(RR(n)) : "v"(a), "v"(b));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 25) {
#define X(n) asm volatile("v_add_f32 %0, %1, %0" : "+v"(RR(n)) : "v"(a));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 26) {
#define X(n) asm volatile("v_cmp_gt_u32 vcc, %1, %0\n v_subb_co_u32 %0, vcc, %0, %1, vcc" : "+v"(RR(n)) : "v"(a) : "vcc");
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void kpk(float* out, int iters, float a, float b) {
  f2 r0 = {(float)threadIdx.x, 1.f}, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + 4.f, r5 = r0 + 5.f, r6 = r0 + 6.f, r7 = r0 + 7.f;
  const f2 av = {a, a}, bv = {b, b};
  for (int i = 0; i < iters; ++i) {
    if constexpr (OP == 0) {
#define X(n) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(r##n) : "v"(av));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else if constexpr (OP == 1) {
#define X(n) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r##n) : "v"(av), "v"(bv));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    } else {
#define X(n) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r##n) : "v"(av));
      REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
    }
  }
  const f2 t = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = t.x + t.y;
}

template <int OP>
void runpk(const char* name, float* d) {
  const int iters = 4000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kpk<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.5f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kpk<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("%-28s %8.3f ms  = %5.2f cycles per wave-instruction @%d MHz (nominal; 2 results per lane)\n", name, ms,
         ms * 1e-3 * clk_khz * 1e3 / (4.0 * iters * 32), clk_khz / 1000);
}

template <int OP>
void run(const char* name, float* d, int insts_per_iter) {
  const int iters = 4000, blocks = 256 * 4;   // 4 blocks of 256 threads per CU = 4 waves / SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.5f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x iters x insts
  const double inst_per_simd = 4.0 * iters * insts_per_iter;
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("%-28s %8.3f ms  %6.2f ns/inst/SIMD  = %5.2f cycles @%d MHz (nominal)\n", name, ms,
         ms * 1e6 / inst_per_simd, ms * 1e-3 * clk_khz * 1e3 / inst_per_simd, clk_khz / 1000);
}

int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
  run<0>("v_fma_f32", d, 32); run<1>("v_min_f32", d, 32); run<2>("v_med3_f32", d, 32);
  run<3>("v_cndmask_b32 (vcc)", d, 32); run<4>("v_bfi_b32", d, 32); run<5>("v_add_u32", d, 32);
  run<6>("v_lshlrev_b32", d, 32); run<7>("v_cmp+v_cndmask (pair)", d, 64); run<8>("v_and_b32", d, 32);
  run<9>("v_sub_f32", d, 32); run<10>("v_min_u32", d, 32); run<11>("v_lshl_or_b32", d, 32);
  run<12>("v_bitop3_b32", d, 32); run<13>("v_and_or_b32", d, 32); run<14>("v_xor_b32", d, 32); run<15>("v_max_u32", d, 32);
  run<16>("v_cmp_eq_f32 -> sgpr pair", d, 32); run<17>("v_cndmask_b32 (sgpr pair)", d, 32);
  run<18>("v_exp_f32", d, 32); run<19>("v_log_f32", d, 32); run<20>("v_frexp_mant_f32", d, 32); run<21>("v_ldexp_f32", d, 32);
  run<22>("v_floor_f32", d, 32); run<23>("v_cvt_i32_f32", d, 32); run<24>("v_min3_f32", d, 32); run<25>("v_add_f32", d, 32);
  run<26>("v_cmp_gt_u32+v_subb_co (pair)", d, 64);
  runpk<0>("v_pk_add_f32", d); runpk<1>("v_pk_fma_f32", d); runpk<2>("v_pk_mul_f32", d);
  return 0;
}
