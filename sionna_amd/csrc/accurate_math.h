// exp / log for float32 arguments evaluated in float64 and rounded once: correctly rounded
// float32 results (up to astronomically rare double-rounding ties).
//
// Why: the reference's boxplus-phi rule phi(x) = log(e^x + 1) - log(e^x - 1) (decoding.py:1120)
// is evaluated literally in float32, clipped at x = 16.635532 ~ ln 2^24 where e^x +- 1 differ
// by one float32 ulp; whether a saturated message maps to exactly 0 (the reference's own
// "all-erasure -> zeros" test, test_ldpc_decoding.py:279-291) depends on exp/log being
// correctly rounded there.  The device libm's expf/logf are 1-2 ulp approximations (measured:
// phi(16.635532) = 5.7e-6 instead of 0), glibc / NumPy on the CPU side are ~0.5 ulp, so the
// HIP kernels carry their own table-free float64 evaluation:
//   exp: x = k ln2 + r, |r| <= ln2/2, degree-11 Taylor, ldexp
//   log: a = 2^e m, m in [sqrt(1/2), sqrt 2), log m = 2 atanh((m-1)/(m+1)), 7 odd terms
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define SAMD_HD __host__ __device__ __forceinline__
#else
#define SAMD_HD static inline
#endif

namespace samd {

SAMD_HD uint32_t bits_of(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
SAMD_HD float float_of(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

// exp(x) for |x| < ~80, result rounded to float32
SAMD_HD float exp_rn_f32(float xf) {
  const double x = (double)xf;
  const double kd = rint(x * 1.4426950408889634074);
  double r = fma(-kd, 6.93147180369123816490e-01, x);        // ln2 split hi/lo (hi has 32 trailing zero bits)
  r = fma(-kd, 1.90821492927058770002e-10, r);
  double p = 1.0 / 39916800.0;
  p = fma(p, r, 1.0 / 3628800.0);
  p = fma(p, r, 1.0 / 362880.0);
  p = fma(p, r, 1.0 / 40320.0);
  p = fma(p, r, 1.0 / 5040.0);
  p = fma(p, r, 1.0 / 720.0);
  p = fma(p, r, 1.0 / 120.0);
  p = fma(p, r, 1.0 / 24.0);
  p = fma(p, r, 1.0 / 6.0);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return (float)ldexp(p, (int)kd);
}

// log(a) for positive normal float32 a, result rounded to float32
SAMD_HD float log_rn_f32(float af) {
  uint32_t u = bits_of(af);
  int e = (int)(u >> 23) - 127;
  uint32_t mb = (u & 0x007FFFFFu) | 0x3F800000u;              // mantissa in [1,2)
  if (mb > 0x3FB504F3u) { mb -= 0x00800000u; e += 1; }        // > sqrt(2): use m/2
  const double m = (double)float_of(mb);
  const double s = (m - 1.0) / (m + 1.0);
  const double s2 = s * s;
  double p = 1.0 / 15.0;
  p = fma(p, s2, 1.0 / 13.0);
  p = fma(p, s2, 1.0 / 11.0);
  p = fma(p, s2, 1.0 / 9.0);
  p = fma(p, s2, 1.0 / 7.0);
  p = fma(p, s2, 1.0 / 5.0);
  p = fma(p, s2, 1.0 / 3.0);
  p = fma(p, s2, 1.0);
  return (float)fma((double)e, 6.93147180559945286227e-01, 2.0 * s * p);
}

// phi of the boxplus-phi check-node rule, float32 semantics of decoding.py:1110-1120
SAMD_HD float phi_f32(float x) {
  x = fminf(fmaxf(x, 8.5e-8f), 16.635532f);
  const float e = exp_rn_f32(x);
  return log_rn_f32(e + 1.f) - log_rn_f32(e - 1.f);
}

}  // namespace samd
