// Node-update arithmetic of the belief-propagation decoders, shared by the HBM-resident engine
// (ldpc_bp_generic.hip) and the on-chip boxplus engine (ldpc5g_onchip_bp.hip): one definition, so both
// engines produce the same bits.  Reference: src/sionna/phy/fec/ldpc/decoding.py:755-1166.
#pragma once
#include "common.h"

namespace samd {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kLargeVal = 100000.f;  // decoding.py:807

__device__ __forceinline__ float sign_nz(float x) { return x < 0.f ? -1.f : 1.f; }  // sign(0) := +1
__device__ __forceinline__ float sgn3(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// phi of the boxplus-phi rule, literal float32 form of decoding.py:1110-1120.
// The upper clip 16.635532 ~ ln 2^24 is where e^x + 1 and e^x - 1 round to neighbouring
// floats: with correctly rounded exp/log (glibc, NumPy, Eigen) the difference of the logs
// is exactly 0 there - the reference's own test demands it ("all-erasure -> zeros",
// test_ldpc_decoding.py:279-291) - but the device libm is a 1-2 ulp approximation and
// returned 5.7e-6.  The saturated point is therefore pinned to the correctly rounded
// value; everywhere else the float32 device functions are used (accurate_math.h keeps the
// float64 evaluation that was measured 1.75x slower on the whole decoder for no gain in
// agreement with the oracle - DESIGN.md "phi conditioning").
//
// exp and log are the device libm's float32 algorithms (v_exp_f32 / v_log_f32 with a two-term
// extended-precision scaling) WITHOUT their special-case handling: after the clip, x is in
// [8.5e-8, 16.64], e^x +- 1 in [2^-23, 3.4e7] - no denormal scaling, overflow, underflow, inf or NaN
// branch can trigger, so dropping them (19 of 43 VALU operations per phi) leaves every bit of the
// result unchanged (tools/ubench/phi_check.hip compares all 1.1e9 floats in [0, 20]).
__device__ __forceinline__ float exp_core_f32(float x) {            // e^x for moderate x
  const float c = __uint_as_float(0x3fb8aa3bu);                      // log2(e), high part
  const float t = x * c;
  float lo = __builtin_fmaf(x, c, -t);
  lo = __builtin_fmaf(x, __uint_as_float(0x32a5705fu), lo);          // low part of log2(e)
  const float r = __builtin_rintf(t);
  const float f = (t - r) + lo;
  return __builtin_ldexpf(__builtin_amdgcn_exp2f(f), (int)r);
}
__device__ __forceinline__ float log_core_f32(float x) {            // ln x for normal positive x
  const float y = __builtin_amdgcn_logf(x);                          // log2
  const float c = __uint_as_float(0x3f317217u);                      // ln 2, high part
  const float r = y * c;
  float lo = __builtin_fmaf(y, c, -r);
  lo = __builtin_fmaf(y, __uint_as_float(0x3377d1cfu), lo);          // low part of ln 2
  return r + lo;
}
__device__ __forceinline__ float phi_fast_f32(float x) {
  x = clampf(x, 8.5e-8f, 16.635532f);
  const float e = exp_core_f32(x);
  const float r = log_core_f32(e + 1.f) - log_core_f32(e - 1.f);
  return (x == 16.635532f) ? 0.f : r;                    // select, not a branch
}

// Two phi evaluations at once: the same operations as phi_fast_f32 per component, with every
// multiply / add / fma issued as a packed-fp32 instruction (v_pk_mul/add/fma_f32: two IEEE results
// per issue slot) - 17 instead of 25 VALU operations per phi, identical bits.
__device__ __forceinline__ f32x2 log_core2_f32(f32x2 x) {
  const f32x2 y = {__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)};
  const float c = __uint_as_float(0x3f317217u), cl = __uint_as_float(0x3377d1cfu);
  const f32x2 c2 = {c, c}, cl2 = {cl, cl};
  const f32x2 r = y * c2;
  f32x2 lo = __builtin_elementwise_fma(y, c2, -r);
  lo = __builtin_elementwise_fma(y, cl2, lo);
  return r + lo;
}
__device__ __forceinline__ f32x2 phi_fast2_f32(float x0, float x1) {
  const f32x2 x = {clampf(x0, 8.5e-8f, 16.635532f), clampf(x1, 8.5e-8f, 16.635532f)};
  const float c = __uint_as_float(0x3fb8aa3bu), cl = __uint_as_float(0x32a5705fu);
  const f32x2 c2 = {c, c}, cl2 = {cl, cl}, one = {1.f, 1.f};
  const f32x2 t = x * c2;
  f32x2 lo = __builtin_elementwise_fma(x, c2, -t);
  lo = __builtin_elementwise_fma(x, cl2, lo);
  const f32x2 r = {__builtin_rintf(t.x), __builtin_rintf(t.y)};
  const f32x2 f = (t - r) + lo;
  const f32x2 e = {__builtin_ldexpf(__builtin_amdgcn_exp2f(f.x), (int)r.x),
                   __builtin_ldexpf(__builtin_amdgcn_exp2f(f.y), (int)r.y)};
  f32x2 res = log_core2_f32(e + one) - log_core2_f32(e - one);
  res.x = (x.x == 16.635532f) ? 0.f : res.x;
  res.y = (x.y == 16.635532f) ? 0.f : res.y;
  return res;
}

// ---- the DEFINED float32 arithmetic of phi (round 3; SAMD_CN_BOXPLUS_PHI) ------------------------------------------
// TensorFlow-CPU evaluates decoding.py:1120 with Eigen's pexp / plog.  Their published algorithm - Cephes expf / logf:
// m = floor(x log2(e) + 1/2), r = x - m ln2 in two parts, degree-5 polynomial, 2^m scaling; frexp to
// [sqrt(1/2), sqrt(2)), degree-8 polynomial in three interleaved parts, e ln2 added last; fused multiply-adds - is one
// fixed sequence of IEEE operations (fma, mul, add, floor, frexp, ldexp: all exact or correctly rounded on gfx950 and on
// any host), so the rule has a bit-level definition: oracle/ldpc_bp.c states it, these functions follow it operation
// for operation, and boxplus-phi soft outputs are compared with array_equal (tests/test_gpu_parity.py).  No hardware
// transcendental (v_exp_f32 / v_log_f32 are 1-ulp approximations without a specification) is involved.  phi(16.635532)
// = 0 and phi(8.5e-8) = 16.6355324 come out exactly, no pinned point.  The price: 14 operations per exp and 23 per log
// against 3 quarter-rate transcendentals - the hardware form stays available as SAMD_CN_BOXPLUS_PHI_FAST.
__device__ __forceinline__ f32x2 spec_exp2_f32(f32x2 x) {
  const f32x2 t = __builtin_elementwise_fma(x, f32x2{1.44269504088896341f, 1.44269504088896341f}, f32x2{0.5f, 0.5f});
  const f32x2 m = {__builtin_floorf(t.x), __builtin_floorf(t.y)};
  f32x2 r = __builtin_elementwise_fma(m, f32x2{-0.693359375f, -0.693359375f}, x);
  r = __builtin_elementwise_fma(m, f32x2{2.12194440e-4f, 2.12194440e-4f}, r);
  const f32x2 z = r * r;
  f32x2 y = {1.9875691500E-4f, 1.9875691500E-4f};
  y = __builtin_elementwise_fma(y, r, f32x2{1.3981999507E-3f, 1.3981999507E-3f});
  y = __builtin_elementwise_fma(y, r, f32x2{8.3334519073E-3f, 8.3334519073E-3f});
  y = __builtin_elementwise_fma(y, r, f32x2{4.1665795894E-2f, 4.1665795894E-2f});
  y = __builtin_elementwise_fma(y, r, f32x2{1.6666665459E-1f, 1.6666665459E-1f});
  y = __builtin_elementwise_fma(y, r, f32x2{5.0000001201E-1f, 5.0000001201E-1f});
  y = __builtin_elementwise_fma(y, z, r);
  y = y + f32x2{1.f, 1.f};
  return f32x2{__builtin_ldexpf(y.x, (int)m.x), __builtin_ldexpf(y.y, (int)m.y)};
}
__device__ __forceinline__ f32x2 spec_log2_f32(f32x2 x) {     // normal positive arguments
  f32x2 f = {__builtin_amdgcn_frexp_mantf(x.x), __builtin_amdgcn_frexp_mantf(x.y)};          // [0.5, 1)
  f32x2 ef = {(float)__builtin_amdgcn_frexp_expf(x.x), (float)__builtin_amdgcn_frexp_expf(x.y)};
  // the specification's "f < sqrt(1/2) ? (e - 1, (f - 1) + f) : (e, f - 1)" with one select per value: k = 2 or 1,
  // f <- f k - 1 (one rounding of an exactly representable result: 2 f - 1 and f - 1 are exact, as are both steps of
  // (f - 1) + f), e <- (e - k) + 1 (small integers) - the same bits in 7 instead of 9 operations per pair
  const f32x2 kk = {f.x < 0.707106781186547524f ? 2.f : 1.f, f.y < 0.707106781186547524f ? 2.f : 1.f};
  f = __builtin_elementwise_fma(f, kk, f32x2{-1.f, -1.f});
  ef = (ef - kk) + f32x2{1.f, 1.f};
  const f32x2 x2 = f * f, x3 = x2 * f;
  f32x2 y = __builtin_elementwise_fma(f32x2{7.0376836292E-2f, 7.0376836292E-2f}, f, f32x2{-1.1514610310E-1f, -1.1514610310E-1f});
  f32x2 y1 = __builtin_elementwise_fma(f32x2{-1.2420140846E-1f, -1.2420140846E-1f}, f, f32x2{1.4249322787E-1f, 1.4249322787E-1f});
  f32x2 y2 = __builtin_elementwise_fma(f32x2{2.0000714765E-1f, 2.0000714765E-1f}, f, f32x2{-2.4999993993E-1f, -2.4999993993E-1f});
  y = __builtin_elementwise_fma(y, f, f32x2{1.1676998740E-1f, 1.1676998740E-1f});
  y1 = __builtin_elementwise_fma(y1, f, f32x2{-1.6668057665E-1f, -1.6668057665E-1f});
  y2 = __builtin_elementwise_fma(y2, f, f32x2{3.3333331174E-1f, 3.3333331174E-1f});
  y = __builtin_elementwise_fma(y, x3, y1);
  y = __builtin_elementwise_fma(y, x3, y2);
  y = y * x3;
  y = __builtin_elementwise_fma(f32x2{-0.5f, -0.5f}, x2, y);
  f = f + y;
  return __builtin_elementwise_fma(ef, f32x2{0.69314718055994530942f, 0.69314718055994530942f}, f);
}
__device__ __forceinline__ f32x2 phi_spec2_f32(float x0, float x1) {
  const f32x2 x = {clampf(x0, 8.5e-8f, 16.635532f), clampf(x1, 8.5e-8f, 16.635532f)};
  const f32x2 e = spec_exp2_f32(x), one = {1.f, 1.f};
  return spec_log2_f32(e + one) - spec_log2_f32(e - one);
}
__device__ __forceinline__ float phi_spec_f32(float x) { return phi_spec2_f32(x, x).x; }

// ---- round 5: a cheaper DEFINED phi (oracle/ldpc_bp.c: phi_expf / phi_logf, the same operations) ---------------------------
// The literal form log(e^x + 1) - log(e^x - 1) stays; exp keeps the Cephes reduction and polynomial but rounds with the
// magic-number addition and scales through the exponent field (14 instead of 17 issue slots per PAIR of values), log is
// table-driven (Tang): p = mantissa in [1, 2), j = its top six bits, (1 / c_j, -ln(1 / c_j) - ln 2) from a 64-entry table,
// r = fma(p, 1 / c_j, -1), ln p by three fma, exponent term by one more - 15 instead of 26 slots per pair and one 8-byte
// table read per value.  The table (csrc/phi_tab.inc = oracle/phi_tab.inc, tools/gen_phi_tables.py) is read through `TAB`:
// from global / constant memory by default (every engine, no set-up), from LDS where a kernel has staged it (PhiTabLds).
// spec_exp2_f32 / spec_log2_f32 above remain the definition of the Polar BP decoder's boxplus.
static __device__ const float2 kPhiTab[64] = {
#include "phi_tab.inc"
};
struct PhiTabGlobal {
  __device__ __forceinline__ float2 get(unsigned byte_off) const {
    return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(kPhiTab) + byte_off);
  }
};
typedef __attribute__((address_space(3))) f32x2 bp_lds_f32x2;
struct PhiTabLds {                    // 512 bytes of LDS at byte offset `base`, filled by phi_tab_stage()
  unsigned base;
  __device__ __forceinline__ float2 get(unsigned byte_off) const {
    const f32x2 v = *(bp_lds_f32x2*)(uintptr_t)(base + byte_off);
    return make_float2(v.x, v.y);
  }
};
constexpr unsigned kPhiNoLds = 0xFFFFFFFFu;
__device__ __forceinline__ void phi_tab_stage(unsigned base, int tid) {          // call with tid = 0..63 (+ a barrier)
  if (tid < 64) *(bp_lds_f32x2*)(uintptr_t)(base + 8u * (unsigned)tid) = f32x2{kPhiTab[tid].x, kPhiTab[tid].y};
}
__device__ __forceinline__ f32x2 phi_exp2_f32(f32x2 x) {
  const f32x2 magic = {12582912.0f, 12582912.0f};
  const f32x2 t = __builtin_elementwise_fma(x, f32x2{1.44269504088896341f, 1.44269504088896341f}, magic);
  const f32x2 m = t - magic;
  f32x2 r = __builtin_elementwise_fma(m, f32x2{-0.693359375f, -0.693359375f}, x);
  r = __builtin_elementwise_fma(m, f32x2{2.12194440e-4f, 2.12194440e-4f}, r);
  const f32x2 z = r * r;
  f32x2 y = {1.9875691500E-4f, 1.9875691500E-4f};
  y = __builtin_elementwise_fma(y, r, f32x2{1.3981999507E-3f, 1.3981999507E-3f});
  y = __builtin_elementwise_fma(y, r, f32x2{8.3334519073E-3f, 8.3334519073E-3f});
  y = __builtin_elementwise_fma(y, r, f32x2{4.1665795894E-2f, 4.1665795894E-2f});
  y = __builtin_elementwise_fma(y, r, f32x2{1.6666665459E-1f, 1.6666665459E-1f});
  y = __builtin_elementwise_fma(y, r, f32x2{5.0000001201E-1f, 5.0000001201E-1f});
  y = __builtin_elementwise_fma(y, z, r);
  y = y + f32x2{1.f, 1.f};
  return f32x2{__uint_as_float(__float_as_uint(y.x) + (__float_as_uint(t.x) << 23)),
               __uint_as_float(__float_as_uint(y.y) + (__float_as_uint(t.y) << 23))};
}
template <class TAB>
__device__ __forceinline__ f32x2 phi_log2_f32(f32x2 x, const TAB& tab) {      // normal positive arguments
  const unsigned b0 = __float_as_uint(x.x), b1 = __float_as_uint(x.y);
  const f32x2 ef = {(float)__builtin_amdgcn_frexp_expf(x.x), (float)__builtin_amdgcn_frexp_expf(x.y)};   // e + 1
  const float2 t0 = tab.get((b0 >> 14) & 0x1F8u), t1 = tab.get((b1 >> 14) & 0x1F8u);
  const f32x2 p = {__uint_as_float((b0 & 0x007FFFFFu) | 0x3F800000u), __uint_as_float((b1 & 0x007FFFFFu) | 0x3F800000u)};
  const f32x2 r = __builtin_elementwise_fma(p, f32x2{t0.x, t1.x}, f32x2{-1.f, -1.f});
  f32x2 t = __builtin_elementwise_fma(r, f32x2{0.333333343f, 0.333333343f}, f32x2{-0.5f, -0.5f});
  t = __builtin_elementwise_fma(t, r, f32x2{1.f, 1.f});
  t = __builtin_elementwise_fma(t, r, f32x2{t0.y, t1.y});
  return __builtin_elementwise_fma(ef, f32x2{0.69314718055994530942f, 0.69314718055994530942f}, t);
}
template <class TAB>
__device__ __forceinline__ f32x2 phi_tab2_f32(float x0, float x1, const TAB& tab) {
  const f32x2 x = {clampf(x0, 8.5e-8f, 16.635532f), clampf(x1, 8.5e-8f, 16.635532f)};
  const f32x2 e = phi_exp2_f32(x), one = {1.f, 1.f};
  f32x2 res = phi_log2_f32(e + one, tab) - phi_log2_f32(e - one, tab);
  res.x = (x.x == 16.635532f) ? 0.f : res.x;
  res.y = (x.y == 16.635532f) ? 0.f : res.y;
  return res;
}

// phi of a rule: the defined arithmetic for SAMD_CN_BOXPLUS_PHI, the hardware transcendentals for ..._PHI_FAST
template <int MODE, class TAB = PhiTabGlobal>
__device__ __forceinline__ f32x2 phi2_f32(float x0, float x1, const TAB& tab = TAB()) {
  if constexpr (MODE == SAMD_CN_BOXPLUS_PHI_FAST) return phi_fast2_f32(x0, x1);
  else return phi_tab2_f32(x0, x1, tab);
}
template <int MODE, class TAB = PhiTabGlobal>
__device__ __forceinline__ float phi1_f32(float x, const TAB& tab = TAB()) {
  if constexpr (MODE == SAMD_CN_BOXPLUS_PHI_FAST) return phi_fast_f32(x);
  else return phi_tab2_f32(x, x, tab).x;
}

// ---- check-node update on one batch column; v[0..d) in CN edge order, in place.
template <int MODE, int MAXD, class TAB = PhiTabGlobal>
__device__ __forceinline__ void cn_update_col(float (&v)[MAXD], int d, float llr_max, float offset, const TAB& tab = TAB()) {
  if constexpr (MODE == SAMD_CN_MINSUM || MODE == SAMD_CN_OFFSET_MINSUM) {
    float sgn[MAXD];
    float node_sign = 1.f, min1 = INFINITY;
#pragma unroll
    for (int i = 0; i < MAXD; ++i)
      if (i < d) {
        const float x = clampf(v[i], -kLargeVal, kLargeVal);
        sgn[i] = sign_nz(x);
        node_sign *= sgn[i];
        v[i] = fabsf(x);
        min1 = fminf(min1, v[i]);
      }
    float min2 = INFINITY, node_sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXD; ++i)
      if (i < d) {
        const float t = v[i] - min1;
        v[i] = (t == 0.f) ? kLargeVal : t;
        min2 = fminf(min2, v[i]);
        node_sum += v[i];
      }
    min2 = min2 + min1;
    node_sum = node_sum - (2.f * kLargeVal - 1.f);
    const float dm = 0.5f * (1.f - sgn3(node_sum));   // 1 <=> unique minimum (decoding.py:872)
    const float min_e = (1.f - dm) * min1 + dm * min2;
#pragma unroll
    for (int i = 0; i < MAXD; ++i)
      if (i < d) {
        float m = (v[i] == kLargeVal) ? min_e : min1;
        m = fmaxf(m - offset, 0.f);
        v[i] = clampf((sgn[i] * node_sign) * m, -llr_max, llr_max);
      }
  } else if constexpr (MODE == SAMD_CN_BOXPLUS_PHI || MODE == SAMD_CN_BOXPLUS_PHI_FAST) {
    // signs as bits: sign_nz(v) = -1 <=> v < 0 (a -0 counts as +), (s_i * node_sign) * q = q with the
    // sign bit s_i ^ node, clip(+-q, +-llr_max) = +-min(q, llr_max) for q >= 0; edges two at a time (phi_fast2)
    unsigned sg[MAXD];
    unsigned node = 0u;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXD; i += 2) {
      if (i + 1 < MAXD && i + 1 < d) {
        sg[i] = (v[i] < 0.f) ? 0x80000000u : 0u;
        sg[i + 1] = (v[i + 1] < 0.f) ? 0x80000000u : 0u;
        node ^= sg[i] ^ sg[i + 1];
        const f32x2 p = phi2_f32<MODE>(fabsf(v[i]), fabsf(v[i + 1]), tab);
        v[i] = p.x; v[i + 1] = p.y;
        sum += p.x;
        sum += p.y;
      } else if (i < d) {
        sg[i] = (v[i] < 0.f) ? 0x80000000u : 0u;
        node ^= sg[i];
        v[i] = phi1_f32<MODE>(fabsf(v[i]), tab);
        sum += v[i];
      }
    }
#pragma unroll
    for (int i = 0; i < MAXD; i += 2) {
      if (i + 1 < MAXD && i + 1 < d) {
        const f32x2 q = phi2_f32<MODE>(-1.f * v[i] + sum, -1.f * v[i + 1] + sum, tab);
        v[i] = __uint_as_float(__float_as_uint(fminf(q.x, llr_max)) ^ (sg[i] ^ node));
        v[i + 1] = __uint_as_float(__float_as_uint(fminf(q.y, llr_max)) ^ (sg[i + 1] ^ node));
      } else if (i < d) {
        const float q = phi1_f32<MODE>(-1.f * v[i] + sum, tab);
        v[i] = __uint_as_float(__float_as_uint(fminf(q, llr_max)) ^ (sg[i] ^ node));
      }
    }
  } else {  // SAMD_CN_BOXPLUS (tanh), decoding.py:1000-1042
    float prod = 1.f;
#pragma unroll
    for (int i = 0; i < MAXD; ++i)
      if (i < d) {
        float t = tanhf(v[i] / 2.f);
        t = (t == 0.f) ? 1e-12f : t;
        v[i] = t;
        prod *= t;
      }
    const float ac = 1.f - 1e-7f;
#pragma unroll
    for (int i = 0; i < MAXD; ++i)
      if (i < d) {
        float e = (1.f / v[i]) * prod;
        e = (fabsf(e) < 1e-7f) ? 0.f : e;
        e = clampf(e, -ac, ac);
        v[i] = clampf(2.f * atanhf(e), -llr_max, llr_max);
      }
  }
}

}  // namespace samd
