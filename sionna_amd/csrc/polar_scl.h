// Shared definitions of the Polar SC / SC-list decoder kernels (polar.hip: generic engine, any list size, SC mode;
// polar_scl_reg.hip: the list engine whose low decoding stages live in registers).
#pragma once
#include "common.h"
#include "options.h"
#include "scl_math.h"
#include <algorithm>
#include <cstdlib>

namespace samd {

// parity of the systematic CRC with generator polynomial g(x) of degree len, zero initial state
// (3GPP 38.212 Sec. 5.1): remainder of u(x) x^len / g(x).  poly = coefficients of x^(len-1)..x^0.
__device__ __forceinline__ uint32_t crc_step(uint32_t reg, uint32_t bit, uint32_t poly, int len) {
  const uint32_t fb = ((reg >> (len - 1)) & 1u) ^ bit;
  reg = (reg << 1) & ((len == 32) ? 0xFFFFFFFFu : ((1u << len) - 1u));
  return fb ? (reg ^ poly) : reg;
}


enum { OP_F = 0, OP_G = 1, OP_LEAF = 2, OP_RATE0 = 3, OP_REP = 4, OP_COMBINE = 5, OP_END = 6, OP_SUBTREE = 7 };
constexpr float kPolarLlrMax = 30.f;

// Metric arithmetic: scl_math.h (float32 operations in a defined order, restated by the CPU oracle
// oracle/polar_scl.c - hard decisions and CRC status are compared bit for bit, tests/test_gpu_polar.py).
// log(1 + e^x) = max(x, 0) + T(|x|) is evaluated ~3000 times per codeword on a few lanes (19 VALU operations).
__device__ __forceinline__ float softplus(float x) { return fmaxf(x, 0.f) + scl_T(fabsf(x)); }
__device__ __forceinline__ float cn_op(float x, float y) {  // polar/decoding.py:684-705
  x = clampf(x, -kPolarLlrMax, kPolarLlrMax);
  y = clampf(y, -kPolarLlrMax, kPolarLlrMax);
  const float lse = fmaxf(x, y) + scl_T(fabsf(x - y));
  return softplus(x + y) - lse;
}

template <typename R>
struct SclArgsT {          // R = float (both engines) or double (precision = "double": generic engine only)
  const R* llr_in;         // [B, n] logits
  R* u_hat;                // [B, k] bits at the information positions of the selected path
  R* crc_status;           // nullable [B]
  const int32_t* ops;      // [num_ops] packed: op | stage<<3 | side<<7 | (a2+2048)<<8
  int num_ops;
  const int32_t* info_pos; // [k]
  const int32_t* iil_inv;  // nullable [k] inverse input interleaver applied before the CRC check
  R* gscratch;             // [grid][L][n - n/2^G] the G top LLR stages of every slot: touched by a handful
                           // of ops per decode, kept in L2 instead of LDS so that more codewords fit on a CU
  unsigned char* gbeta;    // [grid][L][n - n/2^G] partial sums of the same top stages
  const uint32_t* crc_tab; // nullable [k]: remainder contributed by bit i of the CRC-checked sequence (register engine)
  int gstages;             // G
  int batch, n, m, k, L, sc_mode, crc_len;
  uint32_t crc_poly;
  // Polar5GDecoder's rate recovery (decoding.py:2018-2052) as an index in the channel-LLR load (register engine; round 6):
  // llr_in rows then have n_in values, position i of the mother code reads src_a[i] (>= 0: that received value, -1: 0 =
  // punctured, -2: -rm_fill = shortened, known zero) and adds src_b[i] (repetition; null: none).  src_a null: llr_in is [B, n].
  const int32_t* src_a = nullptr;
  const int32_t* src_b = nullptr;
  int n_in = 0;
  R rm_fill = 0;
};
using SclArgs = SclArgsT<float>;

// precision = "double" (reference block.py:25-52): the arithmetic of the reference's own float64 NumPy twin
// (decoding.py:1113-1149: literal log(1 + e^x), log(1 + e^(x+y)) - log(e^x + e^y) on inputs clipped to +-30, libm exp / log) -
// oracle/polar_scl.c precision 1, which is pinned by fixtures generated from that twin (tests/golden/polar_scl_np_golden.npz)
__device__ __forceinline__ double cn_op(double x, double y) {
  x = fmax(fmin(x, (double)kPolarLlrMax), -(double)kPolarLlrMax);
  y = fmax(fmin(y, (double)kPolarLlrMax), -(double)kPolarLlrMax);
  double o = log(1.0 + exp(x + y));
  o -= log(exp(x) + exp(y));
  return o;
}
// (softplus(-l), softplus(+l)) of a clipped LLR: the metric increments of deciding 0 / 1
__device__ __forceinline__ void softplus_pair(float l, float& m0, float& m1) {
  const float tl = scl_T(fabsf(l));                 // shared by softplus(-l) and softplus(l)
  m0 = fmaxf(-l, 0.f) + tl;
  m1 = fmaxf(l, 0.f) + tl;
}
__device__ __forceinline__ void softplus_pair(double l, double& m0, double& m1) {
  m0 = log(1.0 + exp(-l));
  m1 = log(1.0 + exp(l));
}
__device__ __forceinline__ float clamp_llr(float x) { return clampf(x, -kPolarLlrMax, kPolarLlrMax); }
__device__ __forceinline__ double clamp_llr(double x) { return fmax(fmin(x, (double)kPolarLlrMax), -(double)kPolarLlrMax); }


inline int scl_gstages(int n, bool reg_engine = false) {
  // top LLR stages kept in L2 (SAMD_SCL_GSTAGES overrides: 0..5); never more than log2(n) - 2.  The generic engine
  // keeps every other stage in LDS and wants 5 of them out; the register engine has room for one more stage (measured
  // at C5: 4.96 M decodes/s with 4, 4.85 M with 5, 3.4 M with 3 - the 7 KB of list state then cost occupancy)
  int m = 0;
  while ((1 << m) < n) ++m;
  static CachedOpt gstages_opt("SAMD_SCL_GSTAGES");
  int g = (int)gstages_opt.get(reg_engine ? 4 : 5);
  return std::max(0, std::min(g, std::min(5, m - 2)));
}

// polar_scl_reg.hip
bool scl_reg_supported(int n, int list_size, int sc_mode);
int scl_reg_stages(int n, int list_size, int sc_mode);   // register stages R of the engine, -1: generic engine
size_t scl_reg_lds_bytes(int n, int list_size);
int scl_reg_launch(const SclArgs& p, int grid, hipStream_t stream);

}  // namespace samd
