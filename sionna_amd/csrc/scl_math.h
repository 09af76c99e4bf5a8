// Float32 arithmetic of the Polar SC / SC-list decoder (csrc/polar.hip).
//
// The reference (src/sionna/phy/fec/polar/decoding.py:525-723) evaluates path metrics and the boxplus with
// tf.math.softplus / tf.math.reduce_logsumexp / tf.reduce_sum, whose roundings and summation order are not part of
// its contract, while survivor selection compares these metrics exactly - so hard decisions are only reproducible
// if the arithmetic is DEFINED.  The definition (shared with the CPU oracle oracle/polar_scl.c, where it is
// restated independently and checked against float64) uses only IEEE-754 basic operations, fma, rint and ldexp:
//
//   T(a)        = log(1 + e^-a), a >= 0:  t = a * (-log2 e); r = rint(t); f = t - r (exact, |f| <= 1/2);
//                 e = ldexp(1 + f E(f), r) with E of degree 5; T = e Q(e) with Q of degree 8 (log1p(z) / z on
//                 [0, 1]); Horner with fma; |error| < 2e-7 (coefficients: tools/fit_scl_math.py)
//   softplus(x) = max(x, 0) + T(|x|)
//   cn_op(x, y) = softplus(x + y) - (max(x, y) + T(|x - y|)) on inputs clipped to +-30
//
// 19 VALU operations per T, all full rate (the hardware v_exp_f32 / v_log_f32 forms used before were 17 with two
// quarter-rate transcendentals - and differ from any CPU libm in the last bits, which made ~1 % of SCL-8 codewords
// pick a different survivor than the oracle).
#pragma once
#include "common.h"

namespace samd {

// fma(a, b, K) with the constant as the instruction's literal (v_fmaak_f32): the 15 coefficients of T would
// otherwise be hoisted into 15 scalar registers of kernels that are already short of them
template <uint32_t K>
__device__ __forceinline__ float fma_lit(float a, float b) {
  float d;
  asm("v_fmaak_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "n"(K));
  return d;
}

__device__ __forceinline__ float scl_T(float a) {
  const float t = a * -1.44269504f;
  const float r = __builtin_rintf(t);
  const float f = t - r;
  float p = __uint_as_float(0x392209c5u);
  p = fma_lit<0x3aaf8448u>(p, f);
  p = fma_lit<0x3c1d952au>(p, f);
  p = fma_lit<0x3d6357b6u>(p, f);
  p = fma_lit<0x3e75fdf0u>(p, f);
  p = fma_lit<0x3f317218u>(p, f);
  p = __builtin_fmaf(p, f, 1.0f);
  const float e = __builtin_ldexpf(p, (int)r);
  float q = __uint_as_float(0x3ba7f8dcu);
  q = fma_lit<0xbcee2cbcu>(q, e);
  q = fma_lit<0x3d9ec0c1u>(q, e);
  q = fma_lit<0xbe0b497au>(q, e);
  q = fma_lit<0x3e4358e6u>(q, e);
  q = fma_lit<0xbe7e5082u>(q, e);
  q = fma_lit<0x3eaa96bau>(q, e);
  q = fma_lit<0xbeffff46u>(q, e);
  q = fma_lit<0x3f7fffffu>(q, e);
  return e * q;
}

}  // namespace samd
