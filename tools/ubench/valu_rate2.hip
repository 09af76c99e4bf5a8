// Micro-benchmark (round 6): issue time of further VALU forms on gfx950, to find which operations of the generated LDPC
// kernel have a cheaper encoding.  valu_rate.hip (round 3) found two classes: v_add_f32 / v_sub_f32 / v_and / v_xor /
// v_add_u32 at ~2.7 nominal cycles per wave instruction and SIMD, most others (v_med3, v_min, v_fma, v_bitop3, shifts,
// v_cndmask with an SGPR-pair mask) at ~4.4.  Same method: 8 independent registers per wave, 32 instructions per loop
// trip, 4 waves per SIMD resident.  The last section checks the VALUE of "s_mov_b64 vcc, sgpr pair; v_cndmask_b32_e32 x 2"
// (no software wait state between the scalar write of vcc and the vector read) against the e64 form.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
#define RR(n) r##n

#define OPS(F)                                                                                      \
  F(0, "v_mov_b32 %0, %1", "v_mov_b32_e32")                                                         \
  F(1, "v_mul_f32 %0, %1, %0", "v_mul_f32_e32")                                                     \
  F(2, "v_max_f32 %0, %1, %0", "v_max_f32_e32")                                                     \
  F(3, "v_cndmask_b32_e32 %0, %1, %0, vcc", "v_cndmask_b32_e32 (vcc, set once by s_mov)")           \
  F(4, "v_sub_u32 %0, %1, %0", "v_sub_u32_e32")                                                     \
  F(5, "v_or_b32 %0, %1, %0", "v_or_b32_e32")                                                       \
  F(6, "v_not_b32 %0, %0", "v_not_b32_e32")                                                         \
  F(7, "v_lshrrev_b32 %0, 1, %0", "v_lshrrev_b32_e32")                                              \
  F(8, "v_ashrrev_i32 %0, 1, %0", "v_ashrrev_i32_e32")                                              \
  F(9, "v_mul_u32_u24 %0, %1, %0", "v_mul_u32_u24_e32")                                             \
  F(10, "v_max_i32 %0, %1, %0", "v_max_i32_e32")                                                    \
  F(11, "v_cmp_eq_f32_e32 vcc, %1, %0", "v_cmp_eq_f32_e32 -> vcc")                                  \
  F(12, "v_cmp_lt_u32_e32 vcc, %1, %0", "v_cmp_lt_u32_e32 -> vcc")                                  \
  F(13, "v_cmp_eq_f32_e64 vcc, |%1|, %0", "v_cmp_eq_f32_e64 |a| -> vcc")                            \
  F(14, "v_addc_co_u32_e32 %0, vcc, %1, %0, vcc", "v_addc_co_u32_e32")                              \
  F(15, "v_add_co_u32_e32 %0, vcc, %1, %0", "v_add_co_u32_e32")                                     \
  F(16, "v_xnor_b32 %0, %1, %0", "v_xnor_b32_e32")                                                  \
  F(17, "v_bfe_u32 %0, %0, 1, 31", "v_bfe_u32")                                                     \
  F(18, "v_perm_b32 %0, %1, %0, %2", "v_perm_b32")                                                  \
  F(19, "v_alignbit_b32 %0, %1, %0, 8", "v_alignbit_b32")                                           \
  F(20, "v_fmac_f32 %0, %1, %2", "v_fmac_f32_e32")                                                  \
  F(21, "v_add_f32_e64 %0, |%1|, %0", "v_add_f32_e64 |a|")                                          \
  F(22, "v_sub_f32_e64 %0, %1, -%0", "v_sub_f32_e64 neg")                                           \
  F(23, "v_add3_u32 %0, %1, %2, %0", "v_add3_u32")                                                  \
  F(24, "v_or3_b32 %0, %1, %2, %0", "v_or3_b32")                                                    \
  F(25, "v_and_b32 %0, 0x7fffffff, %0", "v_and_b32_e32 literal")                                    \
  F(26, "v_add_f32 %0, s20, %0", "v_add_f32_e32 sgpr src0")                                         \
  F(27, "v_med3_f32 %0, %0, s20, 0", "v_med3_f32 v, s, 0")                                          \
  F(28, "v_min_f32 %0, s20, %0", "v_min_f32_e32 sgpr src0")                                         \
  F(29, "v_cndmask_b32_e64 %0, %1, %0, s[20:21]", "v_cndmask_b32_e64 s[20:21]")                     \
  F(30, "v_subrev_f32 %0, %1, %0", "v_subrev_f32_e32")                                              \
  F(31, "v_mul_f32_e64 %0, |%1|, %0", "v_mul_f32_e64 |a|")                                          \
  F(32, "v_max3_f32 %0, %1, %2, %0", "v_max3_f32")                                                  \
  F(33, "v_minimum3_f32 %0, %1, %2, %0", "v_minimum3_f32 (gfx950)")                                 \
  F(34, "v_xor_b32 %0, 0x80000000, %0", "v_xor_b32_e32 literal")                                    \
  F(35, "v_add_u32 %0, 0x12345, %0", "v_add_u32_e32 literal")                                       \
  F(36, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp row_shr:1")    \
  F(37, "v_cvt_f32_u32 %0, %0", "v_cvt_f32_u32_e32")                                               \
  F(38, "v_and_b32 %0, s20, %0", "v_and_b32_e32 sgpr src0")                                         \
  F(39, "v_bfi_b32 %0, s20, %1, %0", "v_bfi_b32 s, v, v")

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
  asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b32 s20, 0x3fc00000\n s_mov_b32 s21, 0x33" : : : "vcc", "s20", "s21");
  for (int i = 0; i < iters; ++i) {
#define F(ID, TXT, NAME)                                                                                      \
    if constexpr (OP == ID) {                                                                                 \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                         \
        asm volatile(TXT : "+v"(r0) : "v"(a), "v"(b) : "vcc", "s20", "s21");                                 \
        asm volatile(TXT : "+v"(r1) : "v"(a), "v"(b) : "vcc", "s20", "s21");                                 \
        asm volatile(TXT : "+v"(r2) : "v"(a), "v"(b) : "vcc", "s20", "s21");                                 \
        asm volatile(TXT : "+v"(r3) : "v"(a), "v"(b) : "vcc", "s20", "s21");                                 \
        asm volatile(TXT : "+v"(r4) : "v"(a), "v"(b) : "vcc", "s20", "s21");                                 \
        asm volatile(TXT : "+v"(r5) : "v"(a), "v"(b) : "vcc", "s20", "s21");                                 \
        asm volatile(TXT : "+v"(r6) : "v"(a), "v"(b) : "vcc", "s20", "s21");                                 \
        asm volatile(TXT : "+v"(r7) : "v"(a), "v"(b) : "vcc", "s20", "s21");                                 \
      }                                                                                                       \
    }
    OPS(F)
#undef F
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
}

// the variable-node exchange as the generated kernel would issue it: per edge one lane mask (an SGPR pair), two values
// selected both ways.  FORM 0: two v_cndmask_b32_e64 on the SGPR pair; FORM 1: s_mov_b64 vcc + two v_cndmask_b32_e32.
template <int FORM>
__global__ __launch_bounds__(256) void kswap(float* out, int iters, float a, float b, unsigned long long m0) {
  float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
  unsigned long long m[4];
  for (int q = 0; q < 4; ++q) m[q] = __builtin_amdgcn_readfirstlane((unsigned)(m0 >> (q * 7))) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(m0 >> (q * 5 + 32))) << 32);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#define SW(x, y)                                                                                                        \
      if constexpr (FORM == 0) {                                                                                         \
        float lo, hi;                                                                                                    \
        asm volatile("v_cndmask_b32_e64 %0, %2, %3, %4\n v_cndmask_b32_e64 %1, %3, %2, %4" : "=&v"(lo), "=&v"(hi) : "v"(x), "v"(y), "s"(m[q])); \
        x = lo; y = hi;                                                                                                  \
      } else {                                                                                                           \
        float lo, hi;                                                                                                    \
        asm volatile("s_mov_b64 vcc, %4\n v_cndmask_b32_e32 %0, %2, %3, vcc\n v_cndmask_b32_e32 %1, %3, %2, vcc" : "=&v"(lo), "=&v"(hi) : "v"(x), "v"(y), "s"(m[q]) : "vcc"); \
        x = lo; y = hi;                                                                                                  \
      }
      SW(r0, r1) SW(r2, r3) SW(r4, r5) SW(r6, r7)
#undef SW
    }
  }
  float* o = out + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 8;
  o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3; o[4] = r4; o[5] = r5; o[6] = r6; o[7] = r7;
}

template <int OP>
void run(const char* name, float* d) {
  const int iters = 4000, blocks = 256 * 4;   // 4 blocks of 256 threads per CU = 4 waves / SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.5f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.5f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double inst_per_simd = 4.0 * iters * 32;
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("%-44s %8.3f ms  %6.2f ns/inst/SIMD  = %5.2f cycles @%d MHz (nominal)\n", name, ms,
         ms * 1e6 / inst_per_simd, ms * 1e-3 * clk_khz * 1e3 / inst_per_simd, clk_khz / 1000);
}

template <int FORM>
double runswap(const char* name, float* d, std::vector<float>* host) {
  const int iters = 4000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned long long m0 = 0x9e3779b97f4a7c15ull;
  hipLaunchKernelGGL(kswap<FORM>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.5f, 0.5f, m0);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kswap<FORM>, dim3(blocks), dim3(256), 0, 0, d, iters + 1, 1.5f, 0.5f, m0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  host->resize((size_t)blocks * 256 * 8);
  hipMemcpy(host->data(), d, host->size() * 4, hipMemcpyDeviceToHost);
  const double inst_per_simd = 4.0 * iters * 32;   // 16 exchanges x 2 selections per loop trip
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("%-44s %8.3f ms  %6.2f ns/selection/SIMD  = %5.2f cycles @%d MHz (nominal)\n", name, ms,
         ms * 1e6 / inst_per_simd, ms * 1e-3 * clk_khz * 1e3 / inst_per_simd, clk_khz / 1000);
  return ms;
}

int main() {
  float* d; hipMalloc(&d, (size_t)256 * 4 * 256 * 8 * 4);
#define F(ID, TXT, NAME) run<ID>(NAME, d);
  OPS(F)
#undef F
  std::vector<float> h0, h1;
  runswap<0>("exchange: 2 x v_cndmask_b32_e64 (sgpr pair)", d, &h0);
  runswap<1>("exchange: s_mov vcc + 2 x v_cndmask_b32_e32", d, &h1);
  size_t bad = 0;
  for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
  printf("exchange forms agree on %zu of %zu values (%s)\n", h0.size() - bad, h0.size(), bad ? "MISMATCH" : "same");
  return bad ? 1 : 0;
}
