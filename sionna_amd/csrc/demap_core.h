// Per-symbol LLR computation of a square QAM constellation, shared by the standalone demapper (mapping.hip) and the fused
// LS + LMMSE + demapper kernel (mimo.hip): one function, hence the same bits on both paths.
// Replaces Demapper.call / SymbolLogits2LLRs.call   /root/reference/src/sionna/phy/mapping.py:664-691, 927-967
#pragma once
#include "common.h"

namespace samd {

// Reductions of the 2^NB per-level terms over the two label sets of every bit position at once: r0[p] / r1[p] = op over
// the levels whose label has bit p (LSB = 0) clear / set.  Pairwise tree - 2 L instead of NB L operations - shared by the
// maxima and the sums.
template <int L, class OP>
__device__ __forceinline__ void bit_reduce(const float (&e)[L], float* r0, float* r1, OP op) {
  if constexpr (L == 2) {
    r0[0] = e[0];
    r1[0] = e[1];
  } else {
    float ev = e[0], od = e[1];
#pragma unroll
    for (int k = 1; k < L / 2; ++k) { ev = op(ev, e[2 * k]); od = op(od, e[2 * k + 1]); }
    r0[0] = ev;
    r1[0] = od;
    float pr[L / 2];
#pragma unroll
    for (int k = 0; k < L / 2; ++k) pr[k] = op(e[2 * k], e[2 * k + 1]);
    bit_reduce<L / 2>(pr, r0 + 1, r1 + 1, op);
  }
}

// e^x for x <= 0 / ln x on the hardware transcendentals: 2^(x log2 e) with the product carried to extended precision
// (x up to ~ -100 here: a plain float32 product would cost 1e-5 of relative accuracy), ln x = log2(x) ln 2
__device__ __forceinline__ float demap_exp(float x) {
  const float c = __uint_as_float(0x3fb8aa3bu);                       // log2(e), high part (bp_math.h)
  const float t = x * c;
  return __builtin_amdgcn_exp2f(t + __builtin_fmaf(x, __uint_as_float(0x32a5705fu), __builtin_fmaf(x, c, -t)));
}

// LLRs (logits) of the 2 NB bits of one received symbol ys with noise variance no_v; lev: the 2^NB PAM levels of one axis
// (LDS or constant memory).  llr[2 t + ax]: bit t of axis ax (0 = real), in the reference's output order.
template <int NB, bool MAXLOG>
__device__ __forceinline__ void square_qam_llr(const float2 ys, const float no_v, const float* __restrict__ lev, float (&llr)[2 * NB]) {
  constexpr int L = 1 << NB;
  const auto fmx = [](float a, float b) { return fmaxf(a, b); };
  const auto fad = [](float a, float b) { return a + b; };
  // one reciprocal per symbol instead of 2 L IEEE divisions (r03: the divisions were a quarter of the kernel's
  // instructions): every exponent carries the same factor (1 + delta), so the LLRs - differences of log-sum-exps of
  // the exponents - change by that relative delta <= 2^-23, far inside the 1e-5 bar
    const float inv = 1.f / fmaxf(no_v, 1.17549435e-38f);
#pragma unroll
  for (int ax = 0; ax < 2; ++ax) {
    const float ya = ax == 0 ? ys.x : ys.y;
    float e[L];
#pragma unroll
    for (int j = 0; j < L; ++j) { const float d = ya - lev[j]; e[j] = -(d * d) * inv; }
    float mx0[NB], mx1[NB];
    bit_reduce<L>(e, mx0, mx1, fmx);                                 // per-set maxima, bit p = label bit p (LSB = 0)
    if constexpr (MAXLOG) {
#pragma unroll
      for (int t = 0; t < NB; ++t) llr[2 * t + ax] = mx1[NB - 1 - t] - mx0[NB - 1 - t];
    } else {
      // app: every level's exponential is evaluated ONCE, relative to the axis maximum M, and shared by the NB bit
      // positions: logsumexp over a set = M + log(sum of its shares), and M cancels in the difference of the two
      // sets.  A set whose best member lies more than 80 below M would lose its terms to underflow - that bit
      // position (rare below ~25 dB of SNR) takes the per-set maxima instead.
      const float mall = fmaxf(mx0[NB - 1], mx1[NB - 1]);
      float ex[L];
#pragma unroll
      for (int j = 0; j < L; ++j) ex[j] = demap_exp(e[j] - mall);
      float s0[NB], s1[NB];
      bit_reduce<L>(ex, s0, s1, fad);
#pragma unroll
      for (int t = 0; t < NB; ++t) {
        const int p = NB - 1 - t;
        // sums in [e^-80, 2^(NB-1)]: normal numbers; ln 2 (log2 s1 - log2 s0)
        float r = (__builtin_amdgcn_logf(s1[p]) - __builtin_amdgcn_logf(s0[p])) * 0.693147180559945f;
        if (mall - fminf(mx0[p], mx1[p]) > 80.f) {
          float a0 = 0.f, a1 = 0.f;
#pragma unroll
          for (int j = 0; j < L; ++j) {
            if ((j >> p) & 1) a1 += demap_exp(e[j] - mx1[p]); else a0 += demap_exp(e[j] - mx0[p]);
          }
          r = (__builtin_amdgcn_logf(a1) * 0.693147180559945f + mx1[p]) - (__builtin_amdgcn_logf(a0) * 0.693147180559945f + mx0[p]);
        }
        llr[2 * t + ax] = r;
      }
    }
  }
}

}  // namespace samd
