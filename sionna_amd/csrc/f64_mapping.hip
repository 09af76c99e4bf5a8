// precision = "double" (reference src/sionna/phy/block.py:25-52) for the symbol-domain mapping blocks: float64 variants of
//   SymbolDemapper.call       mapping.py:693-792
//   SymbolLogits2LLRs.call    mapping.py:927-967
//   LLRs2SymbolLogits.call    mapping.py:1043-1058
//   SymbolLogits2Moments.call mapping.py:1125-1138
//   PAM2QAM.__call__ (logits) mapping.py:1304-1314
// The same argument layouts as the float32 entries of csrc/mapping.hip; one lane per row, libm exp / log in double, sums in
// ascending point order.  Held to oracle/mapping.py (float64) at 1e-9 (tests/test_gpu_double.py); the tuned kernels are the
// float32 ones.
#include "common.h"

namespace samd {
namespace {

inline int grid_for64(int64_t n, int block) {
  const int64_t g = (n + block - 1) / block;
  return (int)std::min<int64_t>(std::max<int64_t>(g, 1), 256 * 32);
}

__device__ __forceinline__ double log_sigmoid_f64(double p) { return p < 0.0 ? p - log1p(exp(p)) : -log1p(exp(-p)); }

__global__ __launch_bounds__(256) void symbol_demap64_kernel(const double2* __restrict__ y, const double* __restrict__ no, int64_t no_len,
                                                             const double2* __restrict__ points, int P, int64_t num_symbols,
                                                             const double* __restrict__ prior, int64_t prior_len, int hard_out,
                                                             double* __restrict__ out, int32_t* __restrict__ out_idx) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < num_symbols; s += (int64_t)gridDim.x * blockDim.x) {
    const double2 ys = y[s];
    const double n0 = no_len == 1 ? no[0] : no[s];
    const double* pr = prior ? (prior_len == P ? prior : prior + s * P) : nullptr;
    auto expo = [&](int c) -> double {
      const double dr = ys.x - points[c].x, di = ys.y - points[c].y;
      const double e = -(dr * dr + di * di) / n0;
      return pr ? e + pr[c] : e;
    };
    double mx = -INFINITY;
    int arg = 0;
    for (int c = 0; c < P; ++c) {
      const double e = expo(c);
      if (e > mx) { mx = e; arg = c; }
    }
    if (hard_out) { out_idx[s] = arg; continue; }
    double sum = 0.0;
    for (int c = 0; c < P; ++c) sum += exp(expo(c) - mx);
    const double lse = mx + log(sum);
    for (int c = 0; c < P; ++c) out[s * P + c] = expo(c) - lse;
  }
}

__global__ __launch_bounds__(256) void logits2llrs64_kernel(const double* __restrict__ logits, const double* __restrict__ prior,
                                                            int64_t prior_len, int M, int64_t rows, int maxlog, int hard_out,
                                                            double* __restrict__ out) {
  const int P = 1 << M;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < rows; s += (int64_t)gridDim.x * blockDim.x) {
    const double* z = logits + s * P;
    double ls[8][2];
    for (int i = 0; i < M; ++i) {
      const double p = prior ? prior[prior_len == M ? i : s * M + i] : 0.0;
      ls[i][1] = prior ? log_sigmoid_f64(p) : 0.0;
      ls[i][0] = prior ? log_sigmoid_f64(-p) : 0.0;
    }
    auto expo = [&](int c) {
      double e = z[c];
      if (prior) {
        double ps = 0.0;
        for (int i = 0; i < M; ++i) ps += ls[i][(c >> (M - 1 - i)) & 1];
        e = ps + e;
      }
      return e;
    };
    for (int i = 0; i < M; ++i) {
      double mx[2] = {-INFINITY, -INFINITY}, sm[2] = {0.0, 0.0};
      for (int c = 0; c < P; ++c) { const int b = (c >> (M - 1 - i)) & 1; mx[b] = fmax(mx[b], expo(c)); }
      double llr;
      if (maxlog) {
        llr = mx[1] - mx[0];
      } else {
        for (int c = 0; c < P; ++c) { const int b = (c >> (M - 1 - i)) & 1; sm[b] += exp(expo(c) - mx[b]); }
        llr = (log(sm[1]) + mx[1]) - (log(sm[0]) + mx[0]);
      }
      out[s * M + i] = hard_out ? (llr > 0.0 ? 1.0 : 0.0) : llr;
    }
  }
}

__global__ __launch_bounds__(256) void llrs2logits64_kernel(const double* __restrict__ llrs, int m, int64_t rows, int hard_out,
                                                            double* __restrict__ out, int32_t* __restrict__ out_idx) {
  const int P = 1 << m;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < rows; s += (int64_t)gridDim.x * blockDim.x) {
    double ls[8][2];
    for (int j = 0; j < m; ++j) {
      const double l = llrs[s * m + j];
      ls[j][1] = log_sigmoid_f64(l);
      ls[j][0] = log_sigmoid_f64(-l);
    }
    double best = -INFINITY;
    int bi = 0;
    for (int c = 0; c < P; ++c) {
      double acc = 0.0;
      for (int j = 0; j < m; ++j) acc += ls[j][(c >> (m - 1 - j)) & 1];
      if (hard_out) { if (acc > best) { best = acc; bi = c; } }
      else out[s * P + c] = acc;
    }
    if (hard_out) out_idx[s] = bi;
  }
}

__global__ __launch_bounds__(256) void logits2moments64_kernel(const double* __restrict__ logits, const double2* __restrict__ points, int P,
                                                               int64_t rows, double2* __restrict__ mean, double* __restrict__ var) {
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < rows; s += (int64_t)gridDim.x * blockDim.x) {
    const double* z = logits + s * P;
    double mx = -INFINITY;
    for (int c = 0; c < P; ++c) mx = fmax(mx, z[c]);
    double den = 0.0;
    for (int c = 0; c < P; ++c) den += exp(z[c] - mx);
    double mr = 0.0, mi = 0.0;
    for (int c = 0; c < P; ++c) {
      const double pc = exp(z[c] - mx) / den;
      mr += pc * points[c].x;
      mi += pc * points[c].y;
    }
    double v = 0.0;
    for (int c = 0; c < P; ++c) {
      const double pc = exp(z[c] - mx) / den;
      const double dr = points[c].x - mr, di = points[c].y - mi;
      v += pc * (dr * dr + di * di);
    }
    mean[s] = make_double2(mr, mi);
    var[s] = v;
  }
}

__global__ __launch_bounds__(256) void pam2qam_logits64_kernel(const double* __restrict__ pam1, const double* __restrict__ pam2, int nbh,
                                                               int64_t rows, double* __restrict__ out) {
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

}  // namespace
}  // namespace samd

using namespace samd;

extern "C" int samd_symbol_demap_f64(const double* y, const double* no, int64_t no_len, const double* points, int m, int64_t num_symbols,
                                     const double* prior, int64_t prior_len, int hard_out, double* out, int32_t* out_idx, void* stream) {
  SAMD_REQUIRE(y && no && points, "null argument");
  SAMD_REQUIRE(hard_out ? out_idx != nullptr : out != nullptr, "output buffer missing");
  SAMD_REQUIRE(m >= 1 && m <= 10, "num_bits_per_symbol must be in 1..10");
  SAMD_REQUIRE(num_symbols >= 0 && (no_len == 1 || no_len == num_symbols), "no must be scalar or per symbol");
  const int P = 1 << m;
  SAMD_REQUIRE(!prior || prior_len == P || prior_len == num_symbols * P, "prior must be [2^m] or [num_symbols, 2^m]");
  if (num_symbols == 0) return SAMD_OK;
  hipLaunchKernelGGL(symbol_demap64_kernel, dim3(grid_for64(num_symbols, 256)), dim3(256), 0, (hipStream_t)stream, (const double2*)y, no,
                     no_len, (const double2*)points, P, num_symbols, prior, prior_len, hard_out, out, out_idx);
  return launch_status();
}

extern "C" int samd_symbol_logits2llrs_f64(const double* logits, int m, int64_t rows, const double* prior, int64_t prior_len, int method,
                                           int hard_out, double* out, void* stream) {
  SAMD_REQUIRE(logits && out, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= 8, "num_bits_per_symbol must be in 1..8");
  SAMD_REQUIRE(rows >= 0 && (method == 0 || method == 1), "bad argument");
  SAMD_REQUIRE(!prior || prior_len == m || prior_len == rows * m, "prior must be [m] or [rows, m]");
  if (rows == 0) return SAMD_OK;
  hipLaunchKernelGGL(logits2llrs64_kernel, dim3(grid_for64(rows, 256)), dim3(256), 0, (hipStream_t)stream, logits, prior, prior_len, m, rows,
                     method, hard_out, out);
  return launch_status();
}

extern "C" int samd_llrs2symbol_logits_f64(const double* llrs, int m, int64_t rows, int hard_out, double* out, int32_t* out_idx,
                                           void* stream) {
  SAMD_REQUIRE(llrs && (hard_out ? out_idx != nullptr : out != nullptr), "null argument");
  SAMD_REQUIRE(m >= 1 && m <= 8 && rows >= 0, "num_bits_per_symbol must be in 1..8");
  if (rows == 0) return SAMD_OK;
  hipLaunchKernelGGL(llrs2logits64_kernel, dim3(grid_for64(rows, 256)), dim3(256), 0, (hipStream_t)stream, llrs, m, rows, hard_out, out,
                     out_idx);
  return launch_status();
}

extern "C" int samd_symbol_logits2moments_c128(const double* logits, const double* points, int m, int64_t rows, double* mean, double* var,
                                               void* stream) {
  SAMD_REQUIRE(logits && points && mean && var, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= 10 && rows >= 0, "num_bits_per_symbol must be in 1..10");
  if (rows == 0) return SAMD_OK;
  hipLaunchKernelGGL(logits2moments64_kernel, dim3(grid_for64(rows, 256)), dim3(256), 0, (hipStream_t)stream, logits,
                     (const double2*)points, 1 << m, rows, (double2*)mean, var);
  return launch_status();
}

extern "C" int samd_pam2qam_logits_f64(const double* pam1, const double* pam2, int num_bits_per_symbol, int64_t rows, double* out,
                                       void* stream) {
  SAMD_REQUIRE(pam1 && pam2 && out, "null argument");
  SAMD_REQUIRE(num_bits_per_symbol >= 2 && num_bits_per_symbol <= 10 && num_bits_per_symbol % 2 == 0 && rows >= 0,
               "num_bits_per_symbol must be even, 2..10");
  if (rows == 0) return SAMD_OK;
  hipLaunchKernelGGL(pam2qam_logits64_kernel, dim3(grid_for64(rows << num_bits_per_symbol, 256)), dim3(256), 0, (hipStream_t)stream, pam1,
                     pam2, num_bits_per_symbol / 2, rows, out);
  return launch_status();
}
