// QAM / arbitrary-constellation mapper and LLR demapper.
//
// Replaces (reference src/sionna/phy/mapping.py):
//   Mapper.call                      :497-519   bits -> integer label (MSB first) -> LUT
//   Demapper.call                    :664-691   exponents = -|y-c|^2 / max(no, tiny)
//   SymbolLogits2LLRs.call           :927-967   LLR_i = reduce_{c in C_i,1} - reduce_{c in C_i,0}
//                                               reduce = logsumexp ("app") | max ("maxlog")
//
// MI355X design: both are pure streaming kernels (8 B in, 4m B out per symbol).  The
// reference materialises a [..., S, 2^m] distance tensor plus two [..., S, 2^m/2, m]
// gathers in HBM (23.6 GB each at config C2); here one lane owns one symbol, the 2^m
// exponents never leave registers/LDS and the constellation is an LDS-resident LUT.
// Within a wave the m LLRs of a symbol are written as m consecutive floats per lane, i.e.
// each store instruction covers a contiguous 64*m*4-byte span across the wave.
#include "common.h"
#include "bp_math.h"
#include "demap_core.h"

namespace samd {

constexpr int kMaxBits = 10;  // up to 1024 points

template <int MC>   // bits per symbol at compile time (0: run-time m)
__global__ __launch_bounds__(256) void qam_map_kernel(const float* __restrict__ bits,
                                                      const float2* __restrict__ points, int mr,
                                                      int64_t num_symbols, float2* __restrict__ out) {
  const int m = MC ? MC : mr;
  constexpr int MB = MC ? MC : kMaxBits;
  extern __shared__ float2 lut[];
  for (int i = threadIdx.x; i < (1 << m); i += blockDim.x) lut[i] = points[i];
  __syncthreads();
  // The bits of the NEXT symbol are requested before the current symbol is stored: loads and stores share the in-order
  // vmcnt counter, so a load issued after a store cannot be waited for before that store has reached memory (round 3,
  // found on the encoder: 1.08 ms -> 0.69 ms).
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float cur[MB], nxt[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) { cur[i] = (s < num_symbols && i < m) ? bits[s * m + i] : 0.f; nxt[i] = 0.f; }
  for (; s < num_symbols; s += stride) {
    const int64_t s2 = s + stride;
    if (s2 < num_symbols) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
        if (i < m) nxt[i] = bits[s2 * m + i];
    }
    int idx = 0;
#pragma unroll
    for (int i = 0; i < MB; ++i)
      if (i < m) idx = (idx << 1) | ((int)cur[i] & 1);                 // mapping.py:507-511
    out[s] = lut[idx];
#pragma unroll
    for (int i = 0; i < MB; ++i) cur[i] = nxt[i];
  }
}

// Generic demapper: literal per-set reduction with the set's own maximum subtracted
// (tf.reduce_logsumexp semantics).  M is the compile-time number of label bits.
template <int M, bool MAXLOG>
__global__ __launch_bounds__(256) void demap_kernel(const float2* __restrict__ y, const float* __restrict__ no,
                                                    int64_t no_len, const float2* __restrict__ points,
                                                    int64_t num_symbols, int hard_out, float* __restrict__ out,
                                                    const float* __restrict__ prior = nullptr, int64_t prior_len = 0) {
  constexpr int P = 1 << M;
  __shared__ float2 lut[P];
  for (int i = threadIdx.x; i < P; i += blockDim.x) lut[i] = points[i];
  __syncthreads();
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < num_symbols;
       s += (int64_t)gridDim.x * blockDim.x) {
    const float2 ys = y[s];
    const float n0 = fmaxf(no_len == 1 ? no[0] : no[s], 1.17549435e-38f);   // finfo(float32).tiny
    // a-priori term of a point: sum_i log_sigmoid(+-prior_i) (SymbolLogits2LLRs.call, mapping.py:944-958)
    float ls[M][2];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const float p = prior ? prior[prior_len == M ? i : s * M + i] : 0.f;
      ls[i][1] = prior ? (p < 0.f ? p - log1pf(expf(p)) : -log1pf(expf(-p))) : 0.f;
      ls[i][0] = prior ? (-p < 0.f ? -p - log1pf(expf(-p)) : -log1pf(expf(p))) : 0.f;
    }
    // pass 1: per (bit, value) maximum of the exponents
    float mx[M][2];
#pragma unroll
    for (int i = 0; i < M; ++i) { mx[i][0] = -INFINITY; mx[i][1] = -INFINITY; }
    for (int c = 0; c < P; ++c) {
      const float dr = ys.x - lut[c].x, di = ys.y - lut[c].y;
      // |y-c|^2 (mapping.py:672 forms it as abs()**2; the direct sum of squares is the same
      // quantity without the sqrt round trip)
      float e = -(dr * dr + di * di) / n0;
      if (prior) {
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < M; ++i) ps += ls[i][(c >> (M - 1 - i)) & 1];
        e = ps + e;
      }
#pragma unroll
      for (int i = 0; i < M; ++i) {
        const int bit = (c >> (M - 1 - i)) & 1;            // label bit i, MSB first
        if (bit) mx[i][1] = fmaxf(mx[i][1], e); else mx[i][0] = fmaxf(mx[i][0], e);
      }
    }
    float llr[M];
    if constexpr (MAXLOG) {
#pragma unroll
      for (int i = 0; i < M; ++i) llr[i] = mx[i][1] - mx[i][0];
    } else {
      float sm[M][2];
#pragma unroll
      for (int i = 0; i < M; ++i) { sm[i][0] = 0.f; sm[i][1] = 0.f; }
      for (int c = 0; c < P; ++c) {
        const float dr = ys.x - lut[c].x, di = ys.y - lut[c].y;
        float e = -(dr * dr + di * di) / n0;
        if (prior) {
          float ps = 0.f;
#pragma unroll
          for (int i = 0; i < M; ++i) ps += ls[i][(c >> (M - 1 - i)) & 1];
          e = ps + e;
        }
#pragma unroll
        for (int i = 0; i < M; ++i) {
          const int bit = (c >> (M - 1 - i)) & 1;
          if (bit) sm[i][1] += exp_core_f32(e - mx[i][1]); else sm[i][0] += exp_core_f32(e - mx[i][0]);
        }
      }
#pragma unroll
      for (int i = 0; i < M; ++i) llr[i] = (log_core_f32(sm[i][1]) + mx[i][1]) - (log_core_f32(sm[i][0]) + mx[i][0]);
    }
    float* o = out + s * M;
#pragma unroll
    for (int i = 0; i < M; ++i) o[i] = hard_out ? (llr[i] > 0.f ? 1.f : 0.f) : llr[i];
  }
}

// Square-QAM demapper.  For a QAM constellation whose label interleaves the bits of two
// identical Gray-PAM axes (reference mapping.py:108-111: even label bits -> real axis, odd
// -> imaginary), the sum over the 2^(m-1) points of C_{i,b} factorises into (sum over the
// PAM levels with bit b on the bit's own axis) x (sum over ALL levels of the other axis);
// the second factor is common to C_{i,1} and C_{i,0} and cancels in the LLR.  The kernel
// therefore evaluates 2 x 2^(m/2) one-dimensional exponents per symbol instead of 2^m
// two-dimensional ones (64-QAM: 16 distances and 48 exp instead of 64 and 768), which turns
// the demapper from exp-bound into the streaming kernel it should be.  Same per-set
// max-shifted logsumexp as the generic kernel; the cancellation is exact in real
// arithmetic, so the result differs from the reference's 2-D evaluation only by float32
// rounding (checked against the float64 oracle to the 1e-5 bar).
template <int NB, bool MAXLOG>
__global__ __launch_bounds__(256) void demap_square_qam_kernel(const float2* __restrict__ y,
                                                               const float* __restrict__ no, int64_t no_len,
                                                               const float* __restrict__ levels,
                                                               int64_t num_symbols, int hard_out,
                                                               float* __restrict__ out) {
  constexpr int L = 1 << NB;               // PAM levels per axis
  __shared__ float lev[L];
  if (threadIdx.x < L) lev[threadIdx.x] = levels[threadIdx.x];
  __syncthreads();
  // (the next symbol is requested before this one's LLRs are stored: loads and stores share the in-order vmcnt counter)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float2 ynext = s < num_symbols ? y[s] : make_float2(0.f, 0.f);
  float nnext = (s < num_symbols && no_len != 1) ? no[s] : 0.f;
  for (; s < num_symbols; s += stride) {
    const float2 ys = ynext;
    const float ns = nnext;
    if (s + stride < num_symbols) {
      ynext = y[s + stride];
      if (no_len != 1) nnext = no[s + stride];
    }
    float llr[2 * NB];
    square_qam_llr<NB, MAXLOG>(ys, no_len == 1 ? no[0] : ns, lev, llr);
    float* o = out + s * (2 * NB);
#pragma unroll
    for (int i = 0; i < 2 * NB; i += 2) {
      float2 v = make_float2(llr[i], llr[i + 1]);
      if (hard_out) v = make_float2(v.x > 0.f ? 1.f : 0.f, v.y > 0.f ? 1.f : 0.f);
      *reinterpret_cast<float2*>(o + i) = v;              // 8 NB bytes per symbol: 8-byte aligned
    }
  }
}

// SymbolDemapper.call (mapping.py:693-792): normalised log-probabilities (log-softmax) of the constellation points
// for every received symbol, or the index of the most likely point.  One lane per symbol, points in LDS, two passes
// over the points (maximum, then sum of exponentials); prior: log-probabilities [P] or [num_symbols, P].
__global__ __launch_bounds__(256) void symbol_demap_kernel(const float2* __restrict__ y, const float* __restrict__ no,
                                                           int64_t no_len, const float2* __restrict__ points, int P,
                                                           int64_t num_symbols, const float* __restrict__ prior,
                                                           int64_t prior_len, int hard_out, float* __restrict__ out,
                                                           int32_t* __restrict__ out_idx) {
  extern __shared__ float2 lut_dyn[];
  for (int i = threadIdx.x; i < P; i += blockDim.x) lut_dyn[i] = points[i];
  __syncthreads();
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < num_symbols; s += (int64_t)gridDim.x * blockDim.x) {
    const float2 ys = y[s];
    const float n0 = no_len == 1 ? no[0] : no[s];
    const float* pr = prior ? (prior_len == P ? prior : prior + s * P) : nullptr;
    auto expo = [&](int c) -> float {
      const float dr = ys.x - lut_dyn[c].x, di = ys.y - lut_dyn[c].y;
      const float d = sqrtf(dr * dr + di * di);                // tf.abs(y - points), then d**2 (mapping.py:780-783)
      const float e = -(d * d) / n0;
      return pr ? e + pr[c] : e;
    };
    float mx = -INFINITY;
    int arg = 0;
    for (int c = 0; c < P; ++c) {
      const float e = expo(c);
      if (e > mx) { mx = e; arg = c; }                          // first maximum, like tf.argmax
    }
    if (hard_out) { out_idx[s] = arg; continue; }
    float sum = 0.f;
    for (int c = 0; c < P; ++c) sum += expf(expo(c) - mx);
    const float lse = mx + logf(sum);
    for (int c = 0; c < P; ++c) out[s * P + c] = expo(c) - lse;
  }
}

template <bool MAXLOG>
static int launch_demap(int m, dim3 grid, hipStream_t st, const float2* y, const float* no, int64_t no_len,
                        const float2* pts, int64_t ns, int hard, float* out, const float* prior = nullptr,
                        int64_t prior_len = 0) {
#define SAMD_DM(M) case M: hipLaunchKernelGGL((demap_kernel<M, MAXLOG>), grid, dim3(256), 0, st, y, no, no_len, pts, ns, hard, out, prior, prior_len); break
  switch (m) {
    SAMD_DM(1); SAMD_DM(2); SAMD_DM(3); SAMD_DM(4); SAMD_DM(5); SAMD_DM(6); SAMD_DM(7); SAMD_DM(8); SAMD_DM(9); SAMD_DM(10);
    default: set_error("num_bits_per_symbol must be in 1..10"); return SAMD_ERR_UNSUPPORTED;
  }
#undef SAMD_DM
  return SAMD_OK;
}

}  // namespace samd

using namespace samd;

static inline int grid_for(int64_t n, int block) {
  const int64_t g = (n + block - 1) / block;
  return (int)std::min<int64_t>(std::max<int64_t>(g, 1), 256 * 32);
}

extern "C" int samd_qam_map_c64(const float* bits, const float* points, int m, int64_t num_symbols,
                                float* out_symbols, void* stream) {
  SAMD_REQUIRE(bits && points && out_symbols, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= kMaxBits && num_symbols >= 0, "bad m / num_symbols");
  if (num_symbols == 0) return SAMD_OK;
#define SAMD_QM(MC) hipLaunchKernelGGL(qam_map_kernel<MC>, dim3(grid_for(num_symbols, 256)), dim3(256), sizeof(float2) << m, \
                                      (hipStream_t)stream, bits, (const float2*)points, m, num_symbols, (float2*)out_symbols)
  switch (m) {
    case 2: SAMD_QM(2); break;
    case 4: SAMD_QM(4); break;
    case 6: SAMD_QM(6); break;
    case 8: SAMD_QM(8); break;
    default: SAMD_QM(0); break;
  }
#undef SAMD_QM
  return launch_status();
}

extern "C" int samd_square_qam_demap_f32(const float* y, const float* no, int64_t no_len, const float* levels,
                                         int m, int64_t num_symbols, int method, int hard_out, float* out,
                                         void* stream) {
  SAMD_REQUIRE(y && no && levels && out, "null argument");
  SAMD_REQUIRE(num_symbols >= 0 && (no_len == 1 || no_len == num_symbols), "no must be scalar or per symbol");
  SAMD_REQUIRE(method == 0 || method == 1, "method must be 0 (app) or 1 (maxlog)");
  SAMD_REQUIRE(m >= 2 && m <= 10 && m % 2 == 0, "square QAM needs an even m in 2..10");
  if (num_symbols == 0) return SAMD_OK;
  const dim3 grid(grid_for(num_symbols, 256));
  hipStream_t st = (hipStream_t)stream;
#define SAMD_SQ(NB)                                                                                              \
  case NB:                                                                                                       \
    if (method == 1) hipLaunchKernelGGL((demap_square_qam_kernel<NB, true>), grid, dim3(256), 0, st, (const float2*)y, no, no_len, levels, num_symbols, hard_out, out); \
    else hipLaunchKernelGGL((demap_square_qam_kernel<NB, false>), grid, dim3(256), 0, st, (const float2*)y, no, no_len, levels, num_symbols, hard_out, out); \
    break
  switch (m / 2) { SAMD_SQ(1); SAMD_SQ(2); SAMD_SQ(3); SAMD_SQ(4); SAMD_SQ(5); }
#undef SAMD_SQ
  return launch_status();
}

extern "C" int samd_qam_demap_f32(const float* y, const float* no, int64_t no_len, const float* points, int m,
                                  int64_t num_symbols, int method, int hard_out, float* out, void* stream) {
  SAMD_REQUIRE(y && no && points && out, "null argument");
  SAMD_REQUIRE(num_symbols >= 0 && (no_len == 1 || no_len == num_symbols), "no must be scalar or per symbol");
  SAMD_REQUIRE(method == 0 || method == 1, "method must be 0 (app) or 1 (maxlog)");
  if (num_symbols == 0) return SAMD_OK;
  const dim3 grid(grid_for(num_symbols, 256));
  int rc = method == 1
               ? launch_demap<true>(m, grid, (hipStream_t)stream, (const float2*)y, no, no_len, (const float2*)points, num_symbols, hard_out, out)
               : launch_demap<false>(m, grid, (hipStream_t)stream, (const float2*)y, no, no_len, (const float2*)points, num_symbols, hard_out, out);
  if (rc != SAMD_OK) return rc;
  return launch_status();
}

extern "C" int samd_qam_demap_prior_f32(const float* y, const float* no, int64_t no_len, const float* points, int m,
                                        int64_t num_symbols, const float* prior, int64_t prior_len, int method,
                                        int hard_out, float* out, void* stream) {
  SAMD_REQUIRE(y && no && points && prior && out, "null argument");
  SAMD_REQUIRE(num_symbols >= 0 && (no_len == 1 || no_len == num_symbols), "no must be scalar or per symbol");
  SAMD_REQUIRE(prior_len == m || prior_len == num_symbols * m, "prior must be [m] or [num_symbols, m]");
  SAMD_REQUIRE(method == 0 || method == 1, "method must be 0 (app) or 1 (maxlog)");
  if (num_symbols == 0) return SAMD_OK;
  const dim3 grid(grid_for(num_symbols, 256));
  int rc = method == 1 ? launch_demap<true>(m, grid, (hipStream_t)stream, (const float2*)y, no, no_len, (const float2*)points,
                                            num_symbols, hard_out, out, prior, prior_len)
                       : launch_demap<false>(m, grid, (hipStream_t)stream, (const float2*)y, no, no_len, (const float2*)points,
                                             num_symbols, hard_out, out, prior, prior_len);
  if (rc != SAMD_OK) return rc;
  return launch_status();
}

// SymbolLogits2LLRs.call (mapping.py:927-967): LLR_i = reduce_{c: bit i = 1}(logit_c + prior term) - reduce_{c: bit i = 0}(...),
// reduce = logsumexp ("app", with the set's own maximum subtracted like tf.reduce_logsumexp) or max ("maxlog").
// One thread per row of 2^m logits; the per-(bit, value) maxima and sums are formed in two passes over the row.
template <int M, bool MAXLOG>
__global__ __launch_bounds__(256) void logits2llrs_kernel(const float* __restrict__ logits, const float* __restrict__ prior,
                                                          int64_t prior_len, int64_t rows, int hard_out, float* __restrict__ out) {
  constexpr int P = 1 << M;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < rows; s += (int64_t)gridDim.x * blockDim.x) {
    const float* z = logits + s * P;
    float ls[M][2];
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const float p = prior ? prior[prior_len == M ? i : s * M + i] : 0.f;
      ls[i][1] = prior ? (p < 0.f ? p - log1pf(expf(p)) : -log1pf(expf(-p))) : 0.f;
      ls[i][0] = prior ? (-p < 0.f ? -p - log1pf(expf(-p)) : -log1pf(expf(p))) : 0.f;
    }
    auto expo = [&](int c) {
      float e = z[c];
      if (prior) {
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < M; ++i) ps += ls[i][(c >> (M - 1 - i)) & 1];
        e = ps + e;
      }
      return e;
    };
    float mx[M][2];
#pragma unroll
    for (int i = 0; i < M; ++i) { mx[i][0] = -INFINITY; mx[i][1] = -INFINITY; }
    for (int c = 0; c < P; ++c) {
      const float e = expo(c);
#pragma unroll
      for (int i = 0; i < M; ++i) {
        if ((c >> (M - 1 - i)) & 1) mx[i][1] = fmaxf(mx[i][1], e); else mx[i][0] = fmaxf(mx[i][0], e);
      }
    }
    float llr[M];
    if constexpr (MAXLOG) {
#pragma unroll
      for (int i = 0; i < M; ++i) llr[i] = mx[i][1] - mx[i][0];
    } else {
      float sm[M][2];
#pragma unroll
      for (int i = 0; i < M; ++i) { sm[i][0] = 0.f; sm[i][1] = 0.f; }
      for (int c = 0; c < P; ++c) {
        const float e = expo(c);
#pragma unroll
        for (int i = 0; i < M; ++i) {
          if ((c >> (M - 1 - i)) & 1) sm[i][1] += exp_core_f32(e - mx[i][1]); else sm[i][0] += exp_core_f32(e - mx[i][0]);
        }
      }
#pragma unroll
      for (int i = 0; i < M; ++i) llr[i] = (log_core_f32(sm[i][1]) + mx[i][1]) - (log_core_f32(sm[i][0]) + mx[i][0]);
    }
    float* o = out + s * M;
#pragma unroll
    for (int i = 0; i < M; ++i) o[i] = hard_out ? (llr[i] > 0.f ? 1.f : 0.f) : llr[i];
  }
}

template <bool MAXLOG>
static int launch_logits2llrs(int m, dim3 grid, hipStream_t st, const float* logits, const float* prior, int64_t prior_len,
                              int64_t rows, int hard_out, float* out) {
#define SAMD_L2L(MM) case MM: hipLaunchKernelGGL((logits2llrs_kernel<MM, MAXLOG>), grid, dim3(256), 0, st, logits, prior, prior_len, rows, hard_out, out); return SAMD_OK;
  switch (m) {
    SAMD_L2L(1) SAMD_L2L(2) SAMD_L2L(3) SAMD_L2L(4) SAMD_L2L(5) SAMD_L2L(6) SAMD_L2L(7) SAMD_L2L(8)
    default: set_error("num_bits_per_symbol must be in 1..8"); return SAMD_ERR_UNSUPPORTED;
  }
#undef SAMD_L2L
}

extern "C" int samd_symbol_logits2llrs_f32(const float* logits, int m, int64_t rows, const float* prior, int64_t prior_len,
                                           int method, int hard_out, float* out, void* stream) {
  SAMD_REQUIRE(logits && out, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= 8, "num_bits_per_symbol must be in 1..8");
  SAMD_REQUIRE(rows >= 0 && (method == 0 || method == 1), "bad argument");
  SAMD_REQUIRE(!prior || prior_len == m || prior_len == rows * m, "prior must be [m] or [rows, m]");
  if (rows == 0) return SAMD_OK;
  const dim3 grid(grid_for(rows, 256));
  const int rc = method == 1 ? launch_logits2llrs<true>(m, grid, (hipStream_t)stream, logits, prior, prior_len, rows, hard_out, out)
                             : launch_logits2llrs<false>(m, grid, (hipStream_t)stream, logits, prior, prior_len, rows, hard_out, out);
  if (rc != SAMD_OK) return rc;
  return launch_status();
}

// ------------------------------------------------------------------ bit LLRs <-> point logits, moments, PAM -> QAM
// log_sigmoid in the form the demapper kernels above use for the a-priori terms
__device__ __forceinline__ float log_sigmoid_f32(float p) { return p < 0.f ? p - log1pf(expf(p)) : -log1pf(expf(-p)); }

// LLRs2SymbolLogits.call (mapping.py:1043-1058): logit of point c = sum_j log_sigmoid(+-llr_j), + where bit j of the label of
// c (binary representation of c, MSB first) is 1.  A workgroup takes 64 rows: the 2 m log-sigmoids of a row are evaluated
// once into LDS, then every lane forms outputs of the [64, 2^m] tile in storage order (coalesced stores; the output is
// 2^m / m times the input, so the kernel is bound by its writes).  hard_out: the first maximum of the row (tf.argmax),
// from the same sums.
__global__ __launch_bounds__(256) void llrs2logits_kernel(const float* __restrict__ llrs, int m, int64_t rows, int hard_out,
                                                          float* __restrict__ out, int32_t* __restrict__ out_idx) {
  __shared__ float ls[64][8][2];
  const int P = 1 << m;
  for (int64_t row0 = (int64_t)blockIdx.x * 64; row0 < rows; row0 += (int64_t)gridDim.x * 64) {
    const int nr = (int)(rows - row0 < 64 ? rows - row0 : 64);
    __syncthreads();
    for (int e = threadIdx.x; e < nr * m; e += 256) {
      const float l = llrs[row0 * m + e];
      ls[e / m][e % m][1] = log_sigmoid_f32(l);
      ls[e / m][e % m][0] = log_sigmoid_f32(-l);
    }
    __syncthreads();
    if (!hard_out) {
      for (int e = threadIdx.x; e < nr * P; e += 256) {
        const int r = e >> m, c = e & (P - 1);
        float acc = 0.f;
        for (int j = 0; j < m; ++j) acc += ls[r][j][(c >> (m - 1 - j)) & 1];
        out[row0 * P + e] = acc;
      }
    } else if ((int)threadIdx.x < nr) {
      const int r = threadIdx.x;
      float best = -INFINITY;
      int bi = 0;
      for (int c = 0; c < P; ++c) {
        float acc = 0.f;
        for (int j = 0; j < m; ++j) acc += ls[r][j][(c >> (m - 1 - j)) & 1];
        if (acc > best) { best = acc; bi = c; }
      }
      out_idx[row0 + r] = bi;
    }
  }
}

// SymbolLogits2Moments.call (mapping.py:1125-1138): p = softmax(logits), mean = sum_c p_c x_c, var = sum_c p_c |x_c - mean|^2.
// One thread per row of 2^m logits (three passes over a row that stays in L1 / L2, like logits2llrs_kernel); points in LDS.
__global__ __launch_bounds__(256) void logits2moments_kernel(const float* __restrict__ logits, const float2* __restrict__ points,
                                                             int P, int64_t rows, float2* __restrict__ mean,
                                                             float* __restrict__ var) {
  extern __shared__ float2 mom_pts[];
  for (int c = threadIdx.x; c < P; c += 256) mom_pts[c] = points[c];
  __syncthreads();
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < rows; s += (int64_t)gridDim.x * blockDim.x) {
    const float* z = logits + s * P;
    float mx = -INFINITY;
    for (int c = 0; c < P; ++c) mx = fmaxf(mx, z[c]);
    float den = 0.f;
    for (int c = 0; c < P; ++c) den += expf(z[c] - mx);
    float mr = 0.f, mi = 0.f;
    for (int c = 0; c < P; ++c) {
      const float pc = expf(z[c] - mx) / den;
      mr += pc * mom_pts[c].x;
      mi += pc * mom_pts[c].y;
    }
    float v = 0.f;
    for (int c = 0; c < P; ++c) {
      const float pc = expf(z[c] - mx) / den;
      const float dr = mom_pts[c].x - mr, di = mom_pts[c].y - mi;
      v += pc * (dr * dr + di * di);
    }
    mean[s] = make_float2(mr, mi);
    var[s] = v;
  }
}

// PAM2QAM.__call__ with hard_in_out=False (mapping.py:1304-1314), LITERALLY: the P x P matrix pam1_i + pam2_j is flattened
// (index i P + j) and GATHERED with the table t(i, j) = QAM index whose label interleaves the labels of i and j:
// out[i P + j] = pam1[t >> nbh] + pam2[t & (P - 1)].  (For 4- and 16-QAM this places pam1_i + pam2_j at QAM index t(i, j);
// for 64-QAM and above the bit permutation is not its own inverse and the reference's output is this gather - kept.)
__global__ __launch_bounds__(256) void pam2qam_logits_kernel(const float* __restrict__ pam1, const float* __restrict__ pam2,
                                                             int nbh, int64_t rows, float* __restrict__ out) {
  const int P = 1 << nbh, Q = P * P;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < rows * Q; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / Q;
    const int c = (int)(e - r * Q), i = c >> nbh, j = c & (P - 1);
    int t = 0;
    for (int b = 0; b < nbh; ++b)
      t |= (((i >> (nbh - 1 - b)) & 1) << (2 * nbh - 1 - 2 * b)) | (((j >> (nbh - 1 - b)) & 1) << (2 * nbh - 2 - 2 * b));
    out[e] = pam1[r * P + (t >> nbh)] + pam2[r * P + (t & (P - 1))];
  }
}

extern "C" int samd_llrs2symbol_logits_f32(const float* llrs, int m, int64_t rows, int hard_out, float* out, int32_t* out_idx,
                                           void* stream) {
  SAMD_REQUIRE(llrs && (hard_out ? out_idx != nullptr : out != nullptr), "null argument");
  SAMD_REQUIRE(m >= 1 && m <= 8 && rows >= 0, "num_bits_per_symbol must be in 1..8");
  if (rows == 0) return SAMD_OK;
  hipLaunchKernelGGL(llrs2logits_kernel, dim3(grid_for((rows + 63) / 64 * 256, 256)), dim3(256), 0, (hipStream_t)stream, llrs, m,
                     rows, hard_out, out, out_idx);
  return launch_status();
}

extern "C" int samd_symbol_logits2moments_c64(const float* logits, const float* points, int m, int64_t rows, float* mean,
                                              float* var, void* stream) {
  SAMD_REQUIRE(logits && points && mean && var, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= 10 && rows >= 0, "num_bits_per_symbol must be in 1..10");
  if (rows == 0) return SAMD_OK;
  const int P = 1 << m;
  hipLaunchKernelGGL(logits2moments_kernel, dim3(grid_for(rows, 256)), dim3(256), sizeof(float2) * P, (hipStream_t)stream, logits,
                     (const float2*)points, P, rows, (float2*)mean, var);
  return launch_status();
}

extern "C" int samd_pam2qam_logits_f32(const float* pam1, const float* pam2, int num_bits_per_symbol, int64_t rows, float* out,
                                       void* stream) {
  SAMD_REQUIRE(pam1 && pam2 && out, "null argument");
  SAMD_REQUIRE(num_bits_per_symbol >= 2 && num_bits_per_symbol <= 10 && num_bits_per_symbol % 2 == 0 && rows >= 0,
               "num_bits_per_symbol must be even, 2..10");
  if (rows == 0) return SAMD_OK;
  hipLaunchKernelGGL(pam2qam_logits_kernel, dim3(grid_for(rows << num_bits_per_symbol, 256)), dim3(256), 0, (hipStream_t)stream,
                     pam1, pam2, num_bits_per_symbol / 2, rows, out);
  return launch_status();
}

extern "C" int samd_symbol_demap_f32(const float* y, const float* no, int64_t no_len, const float* points, int m,
                                     int64_t num_symbols, const float* prior, int64_t prior_len, int hard_out, float* out,
                                     int32_t* out_idx, void* stream) {
  SAMD_REQUIRE(y && no && points, "null argument");
  SAMD_REQUIRE(hard_out ? out_idx != nullptr : out != nullptr, "output buffer missing");
  SAMD_REQUIRE(m >= 1 && m <= 10, "num_bits_per_symbol must be in 1..10");
  SAMD_REQUIRE(num_symbols >= 0 && (no_len == 1 || no_len == num_symbols), "no must be scalar or per symbol");
  const int P = 1 << m;
  SAMD_REQUIRE(!prior || prior_len == P || prior_len == num_symbols * P, "prior must be [2^m] or [num_symbols, 2^m]");
  if (num_symbols == 0) return SAMD_OK;
  hipLaunchKernelGGL(symbol_demap_kernel, dim3(grid_for(num_symbols, 256)), dim3(256), sizeof(float2) * P, (hipStream_t)stream,
                     (const float2*)y, no, no_len, (const float2*)points, P, num_symbols, prior, prior_len, hard_out, out, out_idx);
  return launch_status();
}
