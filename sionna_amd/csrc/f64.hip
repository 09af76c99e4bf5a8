// precision = "double" (reference src/sionna/phy/block.py:25-52): float64 variants of the belief-propagation
// decoders and the LLR demapper - the blocks whose results depend on the arithmetic precision.
//
//   LDPCBPDecoder.call / _bp_iter     fec/ldpc/decoding.py:544-637, 416-524 (flooding and array schedules)
//   vn_update_sum                     :681-732
//   cn_update_offset_minsum / minsum  :755-953
//   cn_update_tanh                    :955-1043
//   cn_update_phi                     :1045-1166 (float64 clip of phi: [1e-12, 28.324079], :1115-1116)
//   LDPC5GDecoder rate recovery / codeword extraction  :1438-1475, 1508-1531
//   Demapper.call + SymbolLogits2LLRs mapping.py:664-691, 927-967
//
// float32 is the hot path (north star); float64 is the reference's second numeric type and exists for analysis
// runs, so this engine is the plain HBM-resident formulation: messages batch-last [E][B] (every access of a wave is
// a contiguous row segment), one lane per (node, codeword), c2v resident + unclipped totals x_tot, v2c derived on
// the fly as clip(x_tot - c2v) - one code path for flooding (a schedule with a single sub-iteration of all check
// nodes) and array schedules.  Sums run sequentially in edge order like in the float32 engines and the oracle.
#include "ldpc5g.h"
#include "ldpc_graph.h"

namespace samd {
namespace {

constexpr double kLarge64 = 100000.0;
__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
__device__ __forceinline__ double sign_nz64(double x) { return x < 0.0 ? -1.0 : 1.0; }
__device__ __forceinline__ double sgn364(double x) { return x > 0.0 ? 1.0 : (x < 0.0 ? -1.0 : 0.0); }
__device__ __forceinline__ double phi64(double x) {
  x = clampd(x, 1e-12, 28.324079);
  const double e = exp(x);
  return log(e + 1.0) - log(e - 1.0);
}

// [B, cols] logits -> [cols][B] internal LLRs (-clip(logit))
__global__ void prep64_kernel(const double* __restrict__ in, double* __restrict__ llr_t, int batch, int cols, double llr_max) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  for (int c = blockIdx.y; c < cols; c += gridDim.y) llr_t[(size_t)c * batch + b] = -1.0 * clampd(in[(size_t)b * cols + c], -llr_max, llr_max);
}

__global__ void negate64_kernel(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = -1.0 * src[i];
}

// CN update of the listed check nodes.  from_state: the buffer holds v2c (first sub-iteration with msg_v2c given);
// otherwise it holds c2v and v2c_e = clip(x_tot[v] - c2v_e).
template <int MODE>
__global__ void cn64_kernel(double* __restrict__ msg, const double* __restrict__ xtot, const int32_t* __restrict__ cn_ptr,
                            const int32_t* __restrict__ cn_edge, const int32_t* __restrict__ cn_vn,
                            const int32_t* __restrict__ node_list, int n_nodes, int batch, int from_state,
                            double llr_max, double offset) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  for (int slot = blockIdx.y; slot < n_nodes; slot += gridDim.y) {
    const int cn = node_list ? node_list[slot] : slot;
    const int e0 = cn_ptr[cn], d = cn_ptr[cn + 1] - e0;
    auto load = [&](int i) -> double {
      const double m = msg[(size_t)cn_edge[e0 + i] * batch + b];
      if (from_state) return m;
      return clampd(-1.0 * m + xtot[(size_t)cn_vn[e0 + i] * batch + b], -llr_max, llr_max);
    };
    auto store = [&](int i, double v) { msg[(size_t)cn_edge[e0 + i] * batch + b] = v; };
    if constexpr (MODE == SAMD_CN_MINSUM || MODE == SAMD_CN_OFFSET_MINSUM) {
      double node_sign = 1.0, min1 = INFINITY;
      for (int i = 0; i < d; ++i) {
        const double x = clampd(load(i), -kLarge64, kLarge64);
        node_sign *= sign_nz64(x);
        min1 = fmin(min1, fabs(x));
      }
      double min2 = INFINITY, node_sum = 0.0;
      for (int i = 0; i < d; ++i) {
        const double t = fabs(clampd(load(i), -kLarge64, kLarge64)) - min1;
        const double r = (t == 0.0) ? kLarge64 : t;
        min2 = fmin(min2, r);
        node_sum += r;
      }
      min2 = min2 + min1;
      node_sum = node_sum - (2.0 * kLarge64 - 1.0);
      const double dm = 0.5 * (1.0 - sgn364(node_sum));
      const double min_e = (1.0 - dm) * min1 + dm * min2;
      // the new c2v of edge i must not disturb the v2c derivation of the edges still to come: results first
      // into registers is impossible for unbounded degrees, but store() only touches edge i and load(j != i)
      // reads edge j and x_tot - independent
      for (int i = 0; i < d; ++i) {
        const double x = clampd(load(i), -kLarge64, kLarge64);
        const double t = fabs(x) - min1;
        double m = (t == 0.0) ? min_e : min1;
        m = fmax(m - offset, 0.0);
        store(i, clampd((sign_nz64(x) * node_sign) * m, -llr_max, llr_max));
      }
    } else if constexpr (MODE == SAMD_CN_BOXPLUS_PHI) {
      double node_sign = 1.0, sum = 0.0;
      for (int i = 0; i < d; ++i) {
        const double x = load(i);
        node_sign *= sign_nz64(x);
        sum += phi64(fabs(x));
      }
      for (int i = 0; i < d; ++i) {
        const double x = load(i);
        const double e = -1.0 * phi64(fabs(x)) + sum;
        store(i, clampd((sign_nz64(x) * node_sign) * phi64(e), -llr_max, llr_max));
      }
    } else {
      double prod = 1.0;
      for (int i = 0; i < d; ++i) {
        const double t = tanh(load(i) / 2.0);
        prod *= (t == 0.0) ? 1e-12 : t;
      }
      const double ac = 1.0 - 1e-7;
      for (int i = 0; i < d; ++i) {
        double t = tanh(load(i) / 2.0);
        t = (t == 0.0) ? 1e-12 : t;
        double e = (1.0 / t) * prod;
        e = (fabs(e) < 1e-7) ? 0.0 : e;
        e = clampd(e, -ac, ac);
        store(i, clampd(2.0 * atanh(e), -llr_max, llr_max));
      }
    }
  }
}

__global__ void zero_inactive64_kernel(double* __restrict__ msg, const int32_t* __restrict__ cn_ptr,
                                       const int32_t* __restrict__ cn_edge, const int32_t* __restrict__ active, int num_cn,
                                       int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  for (int cn = blockIdx.y; cn < num_cn; cn += gridDim.y) {
    if (active[cn]) continue;
    for (int i = cn_ptr[cn]; i < cn_ptr[cn + 1]; ++i) msg[(size_t)cn_edge[i] * batch + b] = 0.0;
  }
}

// x_tot[v] = (sum_e c2v_e) + llr[v] for the listed VNs (all if node_list == nullptr)
__global__ void vn_total64_kernel(const double* __restrict__ msg, const double* __restrict__ llr_t, double* __restrict__ xtot,
                                  const int32_t* __restrict__ vn_ptr, const int32_t* __restrict__ node_list, int n_nodes,
                                  int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  for (int slot = blockIdx.y; slot < n_nodes; slot += gridDim.y) {
    const int vn = node_list ? node_list[slot] : slot;
    double x = 0.0;
    for (int e = vn_ptr[vn]; e < vn_ptr[vn + 1]; ++e) x += msg[(size_t)e * batch + b];
    xtot[(size_t)vn * batch + b] = x + llr_t[(size_t)vn * batch + b];
  }
}

__global__ void finish64_kernel(const double* __restrict__ xtot, double* __restrict__ out, int batch, int rows, int hard,
                                double llr_max) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    const double x = clampd(xtot[(size_t)r * batch + b], -llr_max, llr_max);
    out[(size_t)b * rows + r] = hard ? ((0.0 >= x) ? 1.0 : 0.0) : -1.0 * x;
  }
}

__global__ void v2c_state64_kernel(const double* __restrict__ msg, const double* __restrict__ xtot,
                                   const int32_t* __restrict__ vn_ptr, double* __restrict__ state, int num_vn, int batch,
                                   double llr_max) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  for (int vn = blockIdx.y; vn < num_vn; vn += gridDim.y) {
    const double x = xtot[(size_t)vn * batch + b];
    for (int e = vn_ptr[vn]; e < vn_ptr[vn + 1]; ++e)
      state[(size_t)e * batch + b] = -1.0 * clampd(-1.0 * msg[(size_t)e * batch + b] + x, -llr_max, llr_max);
  }
}

inline unsigned ygrid(int rows) { return (unsigned)std::max(1, std::min(rows, 4096)); }

void launch_cn64(int mode, dim3 grid, hipStream_t st, double* msg, const double* xtot, const samd_ldpc_graph* g,
                 const int32_t* nodes, int n_nodes, int batch, int from_state, double llr_max, double offset) {
#define SAMD_CN64(M) hipLaunchKernelGGL((cn64_kernel<M>), grid, dim3(64), 0, st, msg, xtot, g->cn_ptr, g->cn_edge, g->cn_vn, nodes, n_nodes, batch, from_state, llr_max, offset)
  switch (mode) {
    case SAMD_CN_BOXPLUS: SAMD_CN64(SAMD_CN_BOXPLUS); break;
    case SAMD_CN_BOXPLUS_PHI: SAMD_CN64(SAMD_CN_BOXPLUS_PHI); break;
    case SAMD_CN_MINSUM: offset = 0.0; SAMD_CN64(SAMD_CN_MINSUM); break;
    default: SAMD_CN64(SAMD_CN_OFFSET_MINSUM); break;
  }
#undef SAMD_CN64
}

// ---- 5G rate recovery / codeword extraction in float64 (index maps of ldpc5g.h)
__global__ void rate_recover64_kernel(const double* __restrict__ llr, double* __restrict__ out, RateMatch p, double llr_max,
                                      int batch) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= p.n_vn) return;
  for (int b = blockIdx.y; b < batch; b += gridDim.y) {
    const double* row = llr + (size_t)b * p.n;
    double r;
    int u = -1;
    if (v < p.k) u = v;
    else if (v >= p.k_ldpc) u = v - (p.k_ldpc - p.k);
    if (u < 0) r = -llr_max;                                       // filler bits
    else {
      const int t = u - 2 * p.z;
      if (t < 0 || t >= p.n) r = 0.0;                              // punctured
      else {
        int o = t;
        if (p.m_int > 0) { const int q = p.n / p.m_int; o = (t / q) + (t % q) * p.m_int; }
        r = row[o];
      }
    }
    out[(size_t)b * p.n_vn + v] = r;
  }
}

__global__ void extract64_kernel(const double* __restrict__ x_hat, double* __restrict__ out, RateMatch p, int batch) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= p.n) return;
  for (int b = blockIdx.y; b < batch; b += gridDim.y)
    out[(size_t)b * p.n + o] = x_hat[(size_t)b * p.n_vn + short_to_full(p, out_to_short(p, o))];
}

// ---- demapper (any constellation, app / maxlog, optional prior LLRs)
__device__ __forceinline__ double log_sigmoid64(double x) { return x < 0.0 ? x - log1p(exp(x)) : -log1p(exp(-x)); }

__global__ __launch_bounds__(256) void demap64_kernel(const double2* __restrict__ y, const double* __restrict__ no,
                                                      int64_t no_len, const double2* __restrict__ points, int m,
                                                      int64_t num_symbols, const double* __restrict__ prior,
                                                      int64_t prior_len, int maxlog, int hard_out,
                                                      double* __restrict__ out) {
  const int P = 1 << m;
  for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < num_symbols; s += (int64_t)gridDim.x * blockDim.x) {
    const double2 ys = y[s];
    const double n0 = fmax(no_len == 1 ? no[0] : no[s], 2.2250738585072014e-308);   // finfo(float64).tiny
    double ls0[10], ls1[10], m0[10], m1[10], s0[10], s1[10];
    for (int i = 0; i < m; ++i) {
      const double p = prior ? prior[prior_len == m ? i : s * m + i] : 0.0;
      ls1[i] = prior ? log_sigmoid64(p) : 0.0;
      ls0[i] = prior ? log_sigmoid64(-p) : 0.0;
      m0[i] = m1[i] = -INFINITY;
      s0[i] = s1[i] = 0.0;
    }
    for (int pass = 0; pass < (maxlog ? 1 : 2); ++pass)
      for (int c = 0; c < P; ++c) {
        const double dr = ys.x - points[c].x, di = ys.y - points[c].y;
        double e = -(dr * dr + di * di) / n0;
        if (prior) {
          double ps = 0.0;
          for (int i = 0; i < m; ++i) ps += ((c >> (m - 1 - i)) & 1) ? ls1[i] : ls0[i];
          e = ps + e;
        }
        for (int i = 0; i < m; ++i) {
          const bool one = (c >> (m - 1 - i)) & 1;
          if (pass == 0) { if (one) m1[i] = fmax(m1[i], e); else m0[i] = fmax(m0[i], e); }
          else { if (one) s1[i] += exp(e - m1[i]); else s0[i] += exp(e - m0[i]); }
        }
      }
    for (int i = 0; i < m; ++i) {
      const double r = maxlog ? (m1[i] - m0[i]) : ((log(s1[i]) + m1[i]) - (log(s0[i]) + m0[i]));
      out[s * m + i] = hard_out ? (r > 0.0 ? 1.0 : 0.0) : r;
    }
  }
}

}  // namespace
}  // namespace samd


// ---------------------------------------------------------------------------------------------------------------------
// precision = "double" of lmmse_equalizer / zf_equalizer / mf_equalizer (reference src/sionna/phy/mimo/equalization.py:
// 101-233, 235-298, 300-470) and, through them, of OFDMEqualizer.call (ofdm/equalization.py:107-230): one lane per
// item, complex128, run-time M <= 16 and K <= 8 (the analysis path: no unrolling, arrays in scratch).  Cholesky of the
// covariance, forward substitutions for L^-1 y and L^-1 H, Cholesky of the K x K Gramian + I - the formulas of the
// reference; results are held to the complex128 NumPy oracle at 1e-9 (tests/test_gpu_double.py).
namespace samd {
namespace {
struct c64 { double re, im; };
__device__ __forceinline__ c64 C64(double r, double i) { return c64{r, i}; }
__device__ __forceinline__ c64 operator+(c64 a, c64 b) { return C64(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ c64 operator-(c64 a, c64 b) { return C64(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ c64 operator*(c64 a, c64 b) { return C64(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ __forceinline__ c64 mulcj(c64 a, c64 b) { return C64(a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im); }   // a conj(b)
__device__ __forceinline__ c64 cjm(c64 a, c64 b) { return C64(a.re * b.re + a.im * b.im, a.re * b.im - a.im * b.re); }    // conj(a) b
__device__ __forceinline__ c64 sc(c64 a, double t) { return C64(a.re * t, a.im * t); }

constexpr int kEqM = 16, kEqK = 8;

// in-place Cholesky factor (lower) of the Hermitian n x n matrix a (row stride ld)
__device__ void chol64(c64* a, int n, int ld) {
  for (int j = 0; j < n; ++j) {
    double d = a[j * ld + j].re;
    for (int q = 0; q < j; ++q) d -= a[j * ld + q].re * a[j * ld + q].re + a[j * ld + q].im * a[j * ld + q].im;
    d = sqrt(d);
    a[j * ld + j] = C64(d, 0.0);
    for (int i = j + 1; i < n; ++i) {
      c64 v = a[i * ld + j];
      for (int q = 0; q < j; ++q) v = v - mulcj(a[i * ld + q], a[j * ld + q]);
      a[i * ld + j] = sc(v, 1.0 / d);
    }
    for (int i = 0; i < j; ++i) a[i * ld + j] = C64(0.0, 0.0);
  }
}
// x <- L^-1 x for `cols` right-hand sides stored as x[row * ldx + col]
__device__ void fwd64(const c64* l, int n, int ld, c64* x, int ldx, int cols) {
  for (int c = 0; c < cols; ++c)
    for (int i = 0; i < n; ++i) {
      c64 v = x[i * ldx + c];
      for (int q = 0; q < i; ++q) v = v - l[i * ld + q] * x[q * ldx + c];
      x[i * ldx + c] = sc(v, 1.0 / l[i * ld + i].re);
    }
}
// x <- L^-H x
__device__ void bwd64(const c64* l, int n, int ld, c64* x, int ldx, int cols) {
  for (int c = 0; c < cols; ++c)
    for (int i = n - 1; i >= 0; --i) {
      c64 v = x[i * ldx + c];
      for (int q = i + 1; q < n; ++q) v = v - cjm(l[q * ld + i], x[q * ldx + c]);
      x[i * ldx + c] = sc(v, 1.0 / l[i * ld + i].re);
    }
}

// mode: 0 LMMSE without whitening, 1 LMMSE with whitening, 2 ZF, 3 MF
__global__ __launch_bounds__(64) void equalize_items_f64_kernel(const double2* __restrict__ y, const double2* __restrict__ h,
                                                               const double2* __restrict__ s, int64_t n, int M, int K,
                                                               int mode, double2* __restrict__ x_hat, double* __restrict__ no_eff) {
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= n) return;
  c64 Y[kEqM], H[kEqM * kEqK], S[kEqM * kEqM], G[kEqK * kEqM], A[kEqM * kEqM];
  for (int i = 0; i < M; ++i) { const double2 v = y[it * M + i]; Y[i] = C64(v.x, v.y); }
  for (int i = 0; i < M * K; ++i) { const double2 v = h[it * M * K + i]; H[i] = C64(v.x, v.y); }        // H[i * K + k]
  for (int i = 0; i < M * M; ++i) { const double2 v = s[it * M * M + i]; S[i] = C64(v.x, v.y); }
  if (mode == 1) {
    // whitening: S = L L^H, y <- L^-1 y, H <- L^-1 H; G = (H^H H + I)^-1 H^H
    chol64(S, M, M);
    fwd64(S, M, M, Y, 1, 1);
    fwd64(S, M, M, H, K, K);
    for (int a = 0; a < K; ++a)
      for (int b = 0; b < K; ++b) {
        c64 v = C64(a == b ? 1.0 : 0.0, 0.0);
        for (int i = 0; i < M; ++i) v = v + cjm(H[i * K + a], H[i * K + b]);
        A[a * K + b] = v;
      }
    chol64(A, K, K);
    for (int a = 0; a < K; ++a)
      for (int i = 0; i < M; ++i) G[a * M + i] = C64(H[i * K + a].re, -H[i * K + a].im);    // H^H
    fwd64(A, K, K, G, M, M);
    bwd64(A, K, K, G, M, M);
  } else if (mode == 0) {
    // G = H^H (H H^H + S)^-1  ->  G^H = (H H^H + S)^-1 H (the matrix is Hermitian)
    for (int a = 0; a < M; ++a)
      for (int b = 0; b < M; ++b) {
        c64 v = S[a * M + b];
        for (int k = 0; k < K; ++k) v = v + mulcj(H[a * K + k], H[b * K + k]);
        A[a * M + b] = v;
      }
    chol64(A, M, M);
    c64 X[kEqM * kEqK];
    for (int i = 0; i < M * K; ++i) X[i] = H[i];
    fwd64(A, M, M, X, K, K);
    bwd64(A, M, M, X, K, K);                                    // X = (H H^H + S)^-1 H = G^H
    for (int a = 0; a < K; ++a)
      for (int i = 0; i < M; ++i) G[a * M + i] = C64(X[i * K + a].re, -X[i * K + a].im);
  } else {
    // ZF: G = (H^H H)^-1 H^H; MF: G = diag(H^H H)^-1 H^H
    for (int a = 0; a < K; ++a)
      for (int b = 0; b < K; ++b) {
        c64 v = C64(0.0, 0.0);
        for (int i = 0; i < M; ++i) v = v + cjm(H[i * K + a], H[i * K + b]);
        A[a * K + b] = v;
      }
    for (int a = 0; a < K; ++a)
      for (int i = 0; i < M; ++i) G[a * M + i] = C64(H[i * K + a].re, -H[i * K + a].im);
    if (mode == 2) {
      chol64(A, K, K);
      fwd64(A, K, K, G, M, M);
      bwd64(A, K, K, G, M, M);
    } else {
      for (int a = 0; a < K; ++a)
        for (int i = 0; i < M; ++i) G[a * M + i] = sc(G[a * M + i], 1.0 / A[a * K + a].re);
    }
  }
  for (int a = 0; a < K; ++a) {
    c64 gy = C64(0.0, 0.0), d = C64(0.0, 0.0);
    for (int i = 0; i < M; ++i) { gy = gy + G[a * M + i] * Y[i]; d = d + G[a * M + i] * H[i * K + a]; }
    if (mode <= 1) {
      // x_hat = (G y) / diag(G H), no_eff = real(1 / d - 1)   (equalization.py:213-231)
      const double dn = d.re * d.re + d.im * d.im;
      const c64 inv = C64(d.re / dn, -d.im / dn);
      const c64 xv = gy * inv;
      x_hat[it * K + a] = make_double2(xv.re, xv.im);
      no_eff[it * K + a] = inv.re - 1.0;
    } else {
      x_hat[it * K + a] = make_double2(gy.re, gy.im);
      // ZF: diag(G S G^H); MF: |diag((I - G H)(I - G H)^H + G S G^H)|
      c64 acc = C64(0.0, 0.0);
      for (int i = 0; i < M; ++i) {
        c64 t = C64(0.0, 0.0);
        for (int j = 0; j < M; ++j) t = t + G[a * M + j] * S[j * M + i];
        acc = acc + mulcj(t, G[a * M + i]);
      }
      if (mode == 3) {
        for (int b = 0; b < K; ++b) {
          c64 e = C64(a == b ? 1.0 : 0.0, 0.0);
          for (int i = 0; i < M; ++i) e = e - G[a * M + i] * H[i * K + b];
          acc = acc + mulcj(e, e);
        }
        no_eff[it * K + a] = sqrt(acc.re * acc.re + acc.im * acc.im);
      } else {
        no_eff[it * K + a] = acc.re;
      }
    }
  }
}
}  // namespace
}  // namespace samd

// precision = "double" of MaximumLikelihoodDetector.call (reference mimo/detection.py:463-537): the channel whitened with the
// Cholesky factor of s, the exponents -||y~ - H~ x||^2 (+ prior) of ALL P^K candidate vectors reduced per stream and point by
// an online logsumexp ("app") or a maximum ("maxlog").  One lane per problem, run-time M <= 16, K <= 8 (arrays in scratch,
// like the equaliser above); the K P (max, scaled sum) pairs and the prior of a lane live in LDS, lane-fastest, 32 lanes per
// workgroup so that K P = 200 points fit.  Held to oracle/ofdm.py::ml_detector (complex128) at 1e-9.
namespace samd {
namespace {
__global__ __launch_bounds__(32) void ml_items_f64_kernel(const double2* __restrict__ y, const double2* __restrict__ h,
                                                         const double2* __restrict__ s, const double* __restrict__ prior,
                                                         const double2* __restrict__ points, int64_t n, int M, int K, int nb,
                                                         int maxlog, double* __restrict__ out) {
  extern __shared__ double ml64_lds[];                       // acc [K P][2][32], prior [K P][32]
  const int P = 1 << nb, lane = threadIdx.x;
  double* acc = ml64_lds;
  double* pr = ml64_lds + (size_t)2 * K * P * 32;
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= n) return;
  c64 Y[kEqM], H[kEqM * kEqK], S[kEqM * kEqM];
  for (int i = 0; i < M; ++i) { const double2 v = y[it * M + i]; Y[i] = C64(v.x, v.y); }
  for (int i = 0; i < M * K; ++i) { const double2 v = h[it * M * K + i]; H[i] = C64(v.x, v.y); }
  for (int i = 0; i < M * M; ++i) { const double2 v = s[it * M * M + i]; S[i] = C64(v.x, v.y); }
  chol64(S, M, M);
  fwd64(S, M, M, Y, 1, 1);
  fwd64(S, M, M, H, K, K);
  for (int a = 0; a < K * P; ++a) {
    acc[(2 * a) * 32 + lane] = -INFINITY;
    acc[(2 * a + 1) * 32 + lane] = 0.0;
    if (prior) pr[a * 32 + lane] = prior[it * K * P + a];
  }
  int64_t nv = 1;
  for (int k = 0; k < K; ++k) nv *= P;
  for (int64_t v = 0; v < nv; ++v) {
    int idx[kEqK];
    c64 x[kEqK];
    int64_t r = v;
    for (int k = K - 1; k >= 0; --k) {                        // stream 0 is the slowest digit (detection.py:398-411)
      idx[k] = (int)(r & (P - 1));
      r >>= nb;
      x[k] = C64(points[idx[k]].x, points[idx[k]].y);
    }
    double e = 0.0;
    for (int m = 0; m < M; ++m) {
      c64 d = Y[m];
      for (int k = 0; k < K; ++k) d = d - H[m * K + k] * x[k];
      e -= d.re * d.re + d.im * d.im;
    }
    if (prior)
      for (int k = 0; k < K; ++k) e += pr[(k * P + idx[k]) * 32 + lane];
    for (int k = 0; k < K; ++k) {
      double* a = acc + (size_t)(2 * (k * P + idx[k])) * 32 + lane;
      const double mx = a[0];
      if (maxlog) {
        a[0] = fmax(mx, e);
      } else if (e > mx) {
        a[32] = a[32] * exp(mx - e) + 1.0;
        a[0] = e;
      } else {
        a[32] += exp(e - mx);
      }
    }
  }
  for (int a = 0; a < K * P; ++a) {
    const double mx = acc[(2 * a) * 32 + lane];
    out[it * K * P + a] = maxlog ? mx : mx + log(acc[(2 * a + 1) * 32 + lane]);
  }
}
}  // namespace
}  // namespace samd

extern "C" int samd_ml_detect_f64(const double* y, const double* h, const double* s, const double* prior, const double* points,
                                  int64_t n, int m, int k, int num_bits_per_symbol, int maxlog, double* out, void* stream) {
  SAMD_REQUIRE(y && h && s && points && out && n >= 0, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= samd::kEqM && k >= 1 && k <= samd::kEqK, "float64 ML detector: 1 <= K <= 8, 1 <= M <= 16");
  SAMD_REQUIRE(num_bits_per_symbol >= 1 && num_bits_per_symbol <= 8, "num_bits_per_symbol must be in 1..8");
  const int64_t P = (int64_t)1 << num_bits_per_symbol;
  SAMD_REQUIRE(k * num_bits_per_symbol <= 16 && k * P <= 200, "num_points ** num_streams <= 65536 and num_streams * num_points <= 200");
  if (n == 0) return SAMD_OK;
  const size_t lds = (size_t)3 * k * P * 32 * sizeof(double);
  SAMD_SET_MAX_LDS(samd::ml_items_f64_kernel, 160 * 1024);
  hipLaunchKernelGGL(samd::ml_items_f64_kernel, dim3((unsigned)((n + 31) / 32)), dim3(32), lds, (hipStream_t)stream, (const double2*)y,
                     (const double2*)h, (const double2*)s, prior, (const double2*)points, n, m, k, num_bits_per_symbol, maxlog, out);
  return samd::launch_status();
}

// precision = "double" of EPDetector.call (reference mimo/detection.py:1166-1312): whitening, the real-valued equivalent of the
// channel, l iterations of expectation propagation on the 2K real dimensions (equations (28)-(38) of [EP2014] as the reference
// writes them; the 2K x 2K inverse by Gauss-Jordan with partial pivoting) and the output stage of the float32 kernel (csrc/mimo.hip
// ep_emit): mode 0 max-log LLRs [nb] / 1 hard bits, 2 the logits of the two PAM constellations [2][P], 3 the QAM index of their
// argmax decisions.  One lane per problem, run-time M <= 16, K <= 8, arrays in scratch.  Held to oracle/ofdm.py::ep_detector.
namespace samd {
namespace {
constexpr int kEpN = 2 * kEqK;

__global__ __launch_bounds__(64) void ep_items_f64_kernel(const double2* __restrict__ y, const double2* __restrict__ h,
                                                         const double2* __restrict__ s, const double* __restrict__ pam, int64_t n, int M,
                                                         int K, int nbh, int iters, double beta, double es, double prec, int mode,
                                                         double* __restrict__ out) {
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= n) return;
  const int P = 1 << nbh, N2 = 2 * K;
  c64 Y[kEqM], H[kEqM * kEqK], S[kEqM * kEqM];
  for (int i = 0; i < M; ++i) { const double2 v = y[it * M + i]; Y[i] = C64(v.x, v.y); }
  for (int i = 0; i < M * K; ++i) { const double2 v = h[it * M * K + i]; H[i] = C64(v.x, v.y); }
  for (int i = 0; i < M * M; ++i) { const double2 v = s[it * M * M + i]; S[i] = C64(v.x, v.y); }
  chol64(S, M, M);
  fwd64(S, M, M, Y, 1, 1);
  fwd64(S, M, M, H, K, K);
  double hth[kEpN * kEpN], hty[kEpN];                         // H_r^T H_r, H_r^T y_r from the complex Gramian / matched filter
  for (int k = 0; k < K; ++k) {
    c64 mf = C64(0.0, 0.0);
    for (int m = 0; m < M; ++m) mf = mf + mulcj(Y[m], H[m * K + k]);
    hty[k] = mf.re; hty[K + k] = mf.im;
    for (int j = 0; j < K; ++j) {
      c64 g = C64(0.0, 0.0);
      for (int m = 0; m < M; ++m) g = g + mulcj(H[m * K + j], H[m * K + k]);   // conj(h[m][k]) h[m][j]
      hth[k * N2 + j] = g.re; hth[k * N2 + K + j] = -g.im; hth[(K + k) * N2 + j] = g.im; hth[(K + k) * N2 + K + j] = g.re;
    }
  }
  const double no = 0.5;
  double lam[kEpN], gam[kEpN], xo[kEpN], vo[kEpN];
  for (int r = 0; r < N2; ++r) { lam[r] = 1.0 / es; gam[r] = 0.0; xo[r] = 0.0; vo[r] = 1.0; }
  for (int itn = 0; itn < iters; ++itn) {
    double a[kEpN * kEpN], ai[kEpN * kEpN];
    for (int r = 0; r < N2; ++r)
      for (int c = 0; c < N2; ++c) { a[r * N2 + c] = hth[r * N2 + c] + (r == c ? no * lam[r] : 0.0); ai[r * N2 + c] = r == c ? 1.0 : 0.0; }
    for (int c = 0; c < N2; ++c) {
      int piv = c;
      double best = fabs(a[c * N2 + c]);
      for (int r = c + 1; r < N2; ++r)
        if (fabs(a[r * N2 + c]) > best) { best = fabs(a[r * N2 + c]); piv = r; }
      if (piv != c)
        for (int j = 0; j < N2; ++j) {
          double t = a[c * N2 + j]; a[c * N2 + j] = a[piv * N2 + j]; a[piv * N2 + j] = t;
          t = ai[c * N2 + j]; ai[c * N2 + j] = ai[piv * N2 + j]; ai[piv * N2 + j] = t;
        }
      const double inv = 1.0 / a[c * N2 + c];
      for (int j = 0; j < N2; ++j) { a[c * N2 + j] *= inv; ai[c * N2 + j] *= inv; }
      for (int r = 0; r < N2; ++r)
        if (r != c) {
          const double f = a[r * N2 + c];
          for (int j = 0; j < N2; ++j) { a[r * N2 + j] -= f * a[c * N2 + j]; ai[r * N2 + j] -= f * ai[c * N2 + j]; }
        }
    }
    double mu_all[kEpN];
    for (int r = 0; r < N2; ++r) {                             // (29) with the multipliers of the previous iteration
      double mu = 0.0;
      for (int c = 0; c < N2; ++c) mu += ai[r * N2 + c] * (hty[c] + no * gam[c]);
      mu_all[r] = mu;
    }
    for (int r = 0; r < N2; ++r) {
      const double mu = mu_all[r], sigma = no * ai[r * N2 + r];                          // (28)
      const double v_obs = fmax(1.0 / (1.0 / sigma - lam[r]), prec);                     // (31)
      const double x_obs = v_obs * (mu / sigma - gam[r]);                                // (32)
      double mx = -INFINITY;
      for (int p = 0; p < P; ++p) { const double d = x_obs - pam[p]; mx = fmax(mx, -(d * d) / (2.0 * v_obs)); }
      double den = 0.0, x = 0.0;
      for (int p = 0; p < P; ++p) {
        const double d = x_obs - pam[p], e = exp(-(d * d) / (2.0 * v_obs) - mx);
        den += e; x += e * pam[p];
      }
      x /= den;
      double v = 0.0;
      for (int p = 0; p < P; ++p) {
        const double d = x_obs - pam[p], dd = pam[p] - x;
        v += dd * dd * (exp(-(d * d) / (2.0 * v_obs) - mx) / den);
      }
      v = fmax(v, prec);                                                                 // (33)
      const double ln = 1.0 / v - 1.0 / v_obs, gn = x / v - x_obs / v_obs;               // (35), (36)
      const double l_new = ln < 0.0 ? lam[r] : ln, g_new = ln < 0.0 ? gam[r] : gn;
      lam[r] = (1.0 - beta) * l_new + beta * lam[r];                                     // (37), (38)
      gam[r] = (1.0 - beta) * g_new + beta * gam[r];
      xo[r] = x_obs; vo[r] = v_obs;
    }
  }
  const int w = mode == 2 ? 2 * P : mode == 3 ? 1 : 2 * nbh;
  for (int k = 0; k < K; ++k) {
    double* o = out + (it * K + k) * w;
    if (mode == 2) {
      for (int half = 0; half < 2; ++half)
        for (int p = 0; p < P; ++p) { const double d = xo[half * K + k] - pam[p]; o[half * P + p] = -(d * d) / (2.0 * vo[half * K + k]); }
    } else if (mode == 3) {
      int ind[2];
      for (int half = 0; half < 2; ++half) {
        double best = -INFINITY;
        int bi = 0;
        for (int p = 0; p < P; ++p) {
          const double d = xo[half * K + k] - pam[p], lg = -(d * d) / (2.0 * vo[half * K + k]);
          if (lg > best) { best = lg; bi = p; }
        }
        ind[half] = bi;
      }
      int qam = 0;
      for (int b = 0; b < nbh; ++b)
        qam |= (((ind[0] >> (nbh - 1 - b)) & 1) << (2 * nbh - 1 - 2 * b)) | (((ind[1] >> (nbh - 1 - b)) & 1) << (2 * nbh - 2 - 2 * b));
      o[0] = (double)qam;
    } else {
      for (int half = 0; half < 2; ++half)
        for (int b = 0; b < nbh; ++b) {
          double m1 = -INFINITY, m0 = -INFINITY;
          for (int p = 0; p < P; ++p) {
            const double d = xo[half * K + k] - pam[p], lg = -(d * d) / (2.0 * vo[half * K + k]);
            if ((p >> (nbh - 1 - b)) & 1) m1 = fmax(m1, lg); else m0 = fmax(m0, lg);
          }
          const double e = m1 - m0;
          o[2 * b + half] = mode == 1 ? (e > 0.0 ? 1.0 : 0.0) : e;
        }
    }
  }
}
}  // namespace
}  // namespace samd

extern "C" int samd_ep_f64(const double* y, const double* h, const double* s, const double* pam, int64_t n, int m, int k,
                           int num_bits_per_symbol, int num_iter, double beta, double es, double prec, int hard_out, double* out,
                           void* stream) {
  SAMD_REQUIRE(y && h && s && pam && out && n >= 0, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= samd::kEqM && k >= 1 && k <= samd::kEqK, "float64 EP detector: 1 <= K <= 8, 1 <= M <= 16");
  SAMD_REQUIRE(num_bits_per_symbol >= 2 && num_bits_per_symbol <= 10 && num_bits_per_symbol % 2 == 0, "QAM with 2..10 bits per symbol");
  SAMD_REQUIRE(num_iter >= 1 && hard_out >= 0 && hard_out <= 3, "bad detector parameters");
  if (n == 0) return SAMD_OK;
  hipLaunchKernelGGL(samd::ep_items_f64_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (const double2*)y,
                     (const double2*)h, (const double2*)s, pam, n, m, k, num_bits_per_symbol / 2, num_iter, beta, es, prec, hard_out, out);
  return samd::launch_status();
}

extern "C" int samd_lmmse_equalizer_c128(const double* y, const double* h, const double* s, int64_t n, int m, int k, int mode,
                                         double* x_hat, double* no_eff, void* stream) {
  SAMD_REQUIRE(y && h && s && x_hat && no_eff && n >= 0, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= samd::kEqM && k >= 1 && k <= samd::kEqK && k <= m, "float64 equaliser: 1 <= K <= 8, K <= M <= 16");
  SAMD_REQUIRE(mode >= 0 && mode <= 3, "mode: 0 LMMSE, 1 LMMSE with whitening, 2 ZF, 3 MF");
  if (n == 0) return SAMD_OK;
  hipLaunchKernelGGL(samd::equalize_items_f64_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     (const double2*)y, (const double2*)h, (const double2*)s, n, m, k, mode, (double2*)x_hat, no_eff);
  return samd::launch_status();
}

using namespace samd;

extern "C" size_t samd_ldpc_bp_workspace_bytes_f64(const samd_ldpc_graph_t* g, int batch) {
  if (!g || batch <= 0) return 0;
  return ((size_t)g->num_edges + 2 * (size_t)g->num_vn) * (size_t)batch * sizeof(double) + 256;
}

extern "C" int samd_ldpc_bp_decode_f64(const samd_ldpc_graph_t* g, const samd_ldpc_schedule_t* sched, const double* llr_in,
                                       double* out, int out_cols, double* state, int state_in, int state_out, int batch,
                                       int num_iter, int cn_mode, double llr_max, double offset, int hard_out,
                                       void* workspace, size_t workspace_bytes, void* stream) {
  SAMD_REQUIRE(g && llr_in && out && batch > 0, "bad argument");
  SAMD_REQUIRE(out_cols > 0 && out_cols <= g->num_vn && num_iter >= 0, "bad out_cols / num_iter");
  SAMD_REQUIRE(cn_mode >= 0 && cn_mode <= 3, "unknown cn_mode");
  SAMD_REQUIRE(!(state_in || state_out) || state, "state buffer missing");
  SAMD_REQUIRE(!sched || sched->num_cn == g->num_cn, "schedule belongs to another graph");
  if (!workspace || workspace_bytes < samd_ldpc_bp_workspace_bytes_f64(g, batch)) {
    set_error("workspace too small (samd_ldpc_bp_workspace_bytes_f64)");
    return SAMD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  double* msg = reinterpret_cast<double*>(align_up((size_t)workspace, 256));
  double* llr_t = msg + (size_t)g->num_edges * batch;
  double* xtot = llr_t + (size_t)g->num_vn * batch;
  const unsigned gx = (unsigned)((batch + 63) / 64);
  hipLaunchKernelGGL(prep64_kernel, dim3(gx, ygrid(g->num_vn)), dim3(64), 0, st, llr_in, llr_t, batch, g->num_vn, llr_max);
  SAMD_HIP_CHECK(hipMemcpyAsync(xtot, llr_t, (size_t)g->num_vn * batch * sizeof(double), hipMemcpyDeviceToDevice, st));
  const size_t nmsg = (size_t)g->num_edges * batch;
  if (state_in) hipLaunchKernelGGL(negate64_kernel, dim3((unsigned)((nmsg + 255) / 256)), dim3(256), 0, st, state, msg, nmsg);
  else SAMD_HIP_CHECK(hipMemsetAsync(msg, 0, nmsg * sizeof(double), st));
  bool from_state = state_in != 0;
  const int num_sub = sched ? sched->num_sub : 1;
  for (int it = 0; it < num_iter; ++it)
    for (int j = 0; j < num_sub; ++j) {
      const int32_t* cns = sched ? sched->cn_list + (size_t)j * sched->width : nullptr;
      const int n_cns = sched ? sched->width : g->num_cn;
      launch_cn64(cn_mode, dim3(gx, ygrid(n_cns)), st, msg, xtot, g, cns, n_cns, batch, from_state ? 1 : 0, llr_max, offset);
      if (from_state && sched)
        hipLaunchKernelGGL(zero_inactive64_kernel, dim3(gx, ygrid(g->num_cn)), dim3(64), 0, st, msg, g->cn_ptr, g->cn_edge,
                           sched->first_mask, g->num_cn, batch);
      from_state = false;
      const int32_t* vns = sched ? sched->vn_list + sched->vn_off[j] : nullptr;
      const int n_vns = sched ? sched->vn_off[j + 1] - sched->vn_off[j] : g->num_vn;
      hipLaunchKernelGGL(vn_total64_kernel, dim3(gx, ygrid(n_vns)), dim3(64), 0, st, msg, llr_t, xtot, g->vn_ptr, vns, n_vns, batch);
    }
  hipLaunchKernelGGL(finish64_kernel, dim3(gx, ygrid(out_cols)), dim3(64), 0, st, xtot, out, batch, out_cols, hard_out, llr_max);
  if (state_out && !(num_iter == 0 && state_in))
    hipLaunchKernelGGL(v2c_state64_kernel, dim3(gx, ygrid(g->num_vn)), dim3(64), 0, st, msg, xtot, g->vn_ptr, state, g->num_vn,
                       batch, llr_max);
  return launch_status();
}

extern "C" int samd_ldpc5g_rate_recover_f64(const samd_ldpc5g_t* h, const double* llr, double* out, int batch, double llr_max,
                                            void* stream) {
  SAMD_REQUIRE(h && llr && out && batch > 0, "bad argument");
  const RateMatch rm = make_rate_match(h);
  hipLaunchKernelGGL(rate_recover64_kernel, dim3((h->n_vn + 255) / 256, ygrid(batch)), dim3(256), 0, (hipStream_t)stream, llr, out,
                     rm, llr_max, batch);
  return launch_status();
}

extern "C" int samd_ldpc5g_extract_codeword_f64(const samd_ldpc5g_t* h, const double* x_hat, double* out, int batch, void* stream) {
  SAMD_REQUIRE(h && x_hat && out && batch > 0, "bad argument");
  const RateMatch rm = make_rate_match(h);
  hipLaunchKernelGGL(extract64_kernel, dim3((h->n + 255) / 256, ygrid(batch)), dim3(256), 0, (hipStream_t)stream, x_hat, out, rm,
                     batch);
  return launch_status();
}

extern "C" int samd_qam_demap_f64(const double* y, const double* no, int64_t no_len, const double* points, int m,
                                  int64_t num_symbols, const double* prior, int64_t prior_len, int method, int hard_out,
                                  double* out, void* stream) {
  SAMD_REQUIRE(y && no && points && out, "null argument");
  SAMD_REQUIRE(num_symbols >= 0 && (no_len == 1 || no_len == num_symbols), "no must be scalar or per symbol");
  SAMD_REQUIRE(m >= 1 && m <= 10, "num_bits_per_symbol must be in 1..10");
  SAMD_REQUIRE(method == 0 || method == 1, "method must be 0 (app) or 1 (maxlog)");
  SAMD_REQUIRE(!prior || prior_len == m || prior_len == num_symbols * m, "prior must be [m] or [num_symbols, m]");
  if (num_symbols == 0) return SAMD_OK;
  const int64_t gsz = std::min<int64_t>((num_symbols + 255) / 256, 256 * 32);
  hipLaunchKernelGGL(demap64_kernel, dim3((unsigned)gsz), dim3(256), 0, (hipStream_t)stream, (const double2*)y, no, no_len,
                     (const double2*)points, m, num_symbols, prior, prior_len, method, hard_out, out);
  return launch_status();
}
