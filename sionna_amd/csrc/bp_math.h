// Node-update arithmetic of the belief-propagation decoders, shared by the HBM-resident engine
// (ldpc_bp_generic.hip) and the on-chip boxplus engine (ldpc5g_onchip_bp.hip): one definition, so both
// engines produce the same bits.  Reference: src/sionna/phy/fec/ldpc/decoding.py:755-1166.
#pragma once
#include "common.h"

namespace samd {

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
__device__ __forceinline__ float phi_fast_f32(float x) {
  x = clampf(x, 8.5e-8f, 16.635532f);
  const float e = expf(x);
  const float r = logf(e + 1.f) - logf(e - 1.f);
  return (x == 16.635532f) ? 0.f : r;                    // select, not a branch
}

// ---- check-node update on one batch column; v[0..d) in CN edge order, in place.
template <int MODE, int MAXD>
__device__ __forceinline__ void cn_update_col(float (&v)[MAXD], int d, float llr_max, float offset) {
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
  } else if constexpr (MODE == SAMD_CN_BOXPLUS_PHI) {
    float sgn[MAXD];
    float node_sign = 1.f, sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXD; ++i)
      if (i < d) {
        sgn[i] = sign_nz(v[i]);
        node_sign *= sgn[i];
        v[i] = phi_fast_f32(fabsf(v[i]));
        sum += v[i];
      }
#pragma unroll
    for (int i = 0; i < MAXD; ++i)
      if (i < d) {
        const float e = -1.f * v[i] + sum;
        v[i] = clampf((sgn[i] * node_sign) * phi_fast_f32(e), -llr_max, llr_max);
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
