// The small per-item linear algebra the reference's MIMO blocks are written in, as entry points of their own:
//   inv_cholesky    utils/linalg.py:8-32      A = L L^H            -> L^-1
//   matrix_pinv     utils/linalg.py:35-59     A [M,K]              -> (A^H A)^-1 A^H
//   whiten_channel  mimo/utils.py:292-356     y, H, S = L L^H      -> L^-1 y, L^-1 H
//   lmmse_matrix    mimo/equalization.py:11-99  H (and S)          -> H^H (H H^H + S)^-1, or (H^H H + I)^-1 H^H without S
// The receiver kernels (csrc/mimo.hip, csrc/f64.hip) carry this algebra fused and unrolled for their sizes; these entries
// are the stand-alone forms for code that calls the helpers directly (precoders, custom equalisers): one lane per item,
// run-time M, K <= 16, arrays in scratch, the formulas of the reference in its order of operations (Cholesky factor,
// triangular solves).  complex64 and complex128 (precision = "double", block.py:25-52).  Held to oracle/linalg.py - NumPy in
// complex128 - at 1e-4 relative in single (conditioning of the test matrices) and 1e-10 in double (tests/test_gpu_linalg.py).
#include "common.h"

namespace samd {
namespace {

constexpr int kD = 16;

template <typename R> struct cx { R re, im; };
template <typename R> __device__ __forceinline__ cx<R> C(R r, R i) { return cx<R>{r, i}; }
template <typename R> __device__ __forceinline__ cx<R> operator+(cx<R> a, cx<R> b) { return C<R>(a.re + b.re, a.im + b.im); }
template <typename R> __device__ __forceinline__ cx<R> operator-(cx<R> a, cx<R> b) { return C<R>(a.re - b.re, a.im - b.im); }
template <typename R> __device__ __forceinline__ cx<R> operator*(cx<R> a, cx<R> b) {
  return C<R>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
template <typename R> __device__ __forceinline__ cx<R> mulcj(cx<R> a, cx<R> b) {      // a conj(b)
  return C<R>(a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im);
}
template <typename R> __device__ __forceinline__ cx<R> cjm(cx<R> a, cx<R> b) {        // conj(a) b
  return C<R>(a.re * b.re + a.im * b.im, a.re * b.im - a.im * b.re);
}
template <typename R> __device__ __forceinline__ cx<R> sc(cx<R> a, R t) { return C<R>(a.re * t, a.im * t); }

// in-place lower Cholesky factor of the Hermitian n x n matrix a (row stride ld); the strict upper triangle is zeroed
template <typename R> __device__ void chol(cx<R>* a, int n, int ld) {
  for (int j = 0; j < n; ++j) {
    R d = a[j * ld + j].re;
    for (int q = 0; q < j; ++q) d -= a[j * ld + q].re * a[j * ld + q].re + a[j * ld + q].im * a[j * ld + q].im;
    d = sqrt(d);
    a[j * ld + j] = C<R>(d, R(0));
    const R inv = R(1) / d;
    for (int i = j + 1; i < n; ++i) {
      cx<R> v = a[i * ld + j];
      for (int q = 0; q < j; ++q) v = v - mulcj(a[i * ld + q], a[j * ld + q]);
      a[i * ld + j] = sc(v, inv);
    }
    for (int i = 0; i < j; ++i) a[i * ld + j] = C<R>(R(0), R(0));
  }
}
// x <- L^-1 x for `cols` right-hand sides x[row * ldx + col]
template <typename R> __device__ void fwd(const cx<R>* l, int n, int ld, cx<R>* x, int ldx, int cols) {
  for (int c = 0; c < cols; ++c)
    for (int i = 0; i < n; ++i) {
      cx<R> v = x[i * ldx + c];
      for (int q = 0; q < i; ++q) v = v - l[i * ld + q] * x[q * ldx + c];
      x[i * ldx + c] = sc(v, R(1) / l[i * ld + i].re);
    }
}
// x <- L^-H x
template <typename R> __device__ void bwd(const cx<R>* l, int n, int ld, cx<R>* x, int ldx, int cols) {
  for (int c = 0; c < cols; ++c)
    for (int i = n - 1; i >= 0; --i) {
      cx<R> v = x[i * ldx + c];
      for (int q = i + 1; q < n; ++q) v = v - cjm(l[q * ld + i], x[q * ldx + c]);
      x[i * ldx + c] = sc(v, R(1) / l[i * ld + i].re);
    }
}

template <typename R> __device__ __forceinline__ void load(const R* __restrict__ src, int64_t off, int count, cx<R>* dst) {
  for (int i = 0; i < count; ++i) dst[i] = C<R>(src[2 * (off + i)], src[2 * (off + i) + 1]);
}
template <typename R> __device__ __forceinline__ void store(const cx<R>* src, int count, R* __restrict__ dst, int64_t off) {
  for (int i = 0; i < count; ++i) { dst[2 * (off + i)] = src[i].re; dst[2 * (off + i) + 1] = src[i].im; }
}

template <typename R> __global__ __launch_bounds__(64) void inv_cholesky_kernel(const R* __restrict__ a, int64_t n, int M, R* __restrict__ out) {
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= n) return;
  cx<R> A[kD * kD], X[kD * kD];
  load(a, it * M * M, M * M, A);
  chol(A, M, M);
  for (int i = 0; i < M * M; ++i) X[i] = C<R>((i / M == i % M) ? R(1) : R(0), R(0));
  fwd(A, M, M, X, M, M);
  store(X, M * M, out, it * M * M);
}

template <typename R> __global__ __launch_bounds__(64) void whiten_kernel(const R* __restrict__ y, const R* __restrict__ h,
                                                                         const R* __restrict__ s, int64_t n, int M, int K,
                                                                         R* __restrict__ yw, R* __restrict__ hw) {
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= n) return;
  cx<R> S[kD * kD], H[kD * kD], Y[kD];
  load(s, it * M * M, M * M, S);
  load(h, it * M * K, M * K, H);
  load(y, it * M, M, Y);
  chol(S, M, M);
  fwd(S, M, M, Y, 1, 1);
  fwd(S, M, M, H, K, K);
  store(Y, M, yw, it * M);
  store(H, M * K, hw, it * M * K);
}

// mode 0: G = H^H (H H^H + S)^-1; 1: G = (H^H H + I)^-1 H^H (s == nullptr); 2: pseudo-inverse (H^H H)^-1 H^H
template <typename R> __global__ __launch_bounds__(64) void lmmse_matrix_kernel(const R* __restrict__ h, const R* __restrict__ s, int64_t n,
                                                                               int M, int K, int mode, R* __restrict__ g) {
  const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= n) return;
  cx<R> H[kD * kD], A[kD * kD], G[kD * kD];
  load(h, it * M * K, M * K, H);                                     // H[i * K + k]
  if (mode == 0) {
    load(s, it * M * M, M * M, A);
    for (int a = 0; a < M; ++a)
      for (int b = 0; b < M; ++b) {
        cx<R> v = A[a * M + b];
        for (int k = 0; k < K; ++k) v = v + mulcj(H[a * K + k], H[b * K + k]);
        A[a * M + b] = v;
      }
    chol(A, M, M);
    fwd(A, M, M, H, K, K);
    bwd(A, M, M, H, K, K);                                           // (H H^H + S)^-1 H = G^H
    for (int a = 0; a < K; ++a)
      for (int i = 0; i < M; ++i) G[a * M + i] = C<R>(H[i * K + a].re, -H[i * K + a].im);
  } else {
    for (int a = 0; a < K; ++a)
      for (int b = 0; b < K; ++b) {
        cx<R> v = C<R>((mode == 1 && a == b) ? R(1) : R(0), R(0));
        for (int i = 0; i < M; ++i) v = v + cjm(H[i * K + a], H[i * K + b]);
        A[a * K + b] = v;
      }
    chol(A, K, K);
    for (int a = 0; a < K; ++a)
      for (int i = 0; i < M; ++i) G[a * M + i] = C<R>(H[i * K + a].re, -H[i * K + a].im);
    fwd(A, K, K, G, M, M);
    bwd(A, K, K, G, M, M);
  }
  store(G, K * M, g, it * K * M);
}

inline unsigned blocks(int64_t n) { return (unsigned)((n + 63) / 64); }

template <typename R> int inv_cholesky(const R* a, int64_t n, int m, R* out, void* stream) {
  SAMD_REQUIRE(a && out && n >= 0, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= kD, "inv_cholesky: 1 <= M <= 16");
  if (n == 0) return SAMD_OK;
  hipLaunchKernelGGL(inv_cholesky_kernel<R>, dim3(blocks(n)), dim3(64), 0, (hipStream_t)stream, a, n, m, out);
  return launch_status();
}

template <typename R> int whiten(const R* y, const R* h, const R* s, int64_t n, int m, int k, R* yw, R* hw, void* stream) {
  SAMD_REQUIRE(y && h && s && yw && hw && n >= 0, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= kD && k >= 1 && k <= kD, "whiten_channel: 1 <= M, K <= 16");
  if (n == 0) return SAMD_OK;
  hipLaunchKernelGGL(whiten_kernel<R>, dim3(blocks(n)), dim3(64), 0, (hipStream_t)stream, y, h, s, n, m, k, yw, hw);
  return launch_status();
}

template <typename R> int lmmse_matrix(const R* h, const R* s, int64_t n, int m, int k, int mode, R* g, void* stream) {
  SAMD_REQUIRE(h && g && n >= 0, "null argument");
  SAMD_REQUIRE(m >= 1 && m <= kD && k >= 1 && k <= kD, "1 <= M, K <= 16");
  SAMD_REQUIRE(mode == 0 ? s != nullptr : s == nullptr, "covariance argument does not fit the mode");
  if (n == 0) return SAMD_OK;
  hipLaunchKernelGGL(lmmse_matrix_kernel<R>, dim3(blocks(n)), dim3(64), 0, (hipStream_t)stream, h, s, n, m, k, mode, g);
  return launch_status();
}

}  // namespace
}  // namespace samd

using namespace samd;

extern "C" int samd_inv_cholesky_c64(const float* a, int64_t n, int m, float* out, void* stream) { return inv_cholesky<float>(a, n, m, out, stream); }
extern "C" int samd_inv_cholesky_c128(const double* a, int64_t n, int m, double* out, void* stream) { return inv_cholesky<double>(a, n, m, out, stream); }

extern "C" int samd_whiten_channel_c64(const float* y, const float* h, const float* s, int64_t n, int m, int k, float* yw, float* hw,
                                       void* stream) {
  return whiten<float>(y, h, s, n, m, k, yw, hw, stream);
}
extern "C" int samd_whiten_channel_c128(const double* y, const double* h, const double* s, int64_t n, int m, int k, double* yw, double* hw,
                                        void* stream) {
  return whiten<double>(y, h, s, n, m, k, yw, hw, stream);
}

extern "C" int samd_lmmse_matrix_c64(const float* h, const float* s, int64_t n, int m, int k, float* g, void* stream) {
  return lmmse_matrix<float>(h, s, n, m, k, s ? 0 : 1, g, stream);
}
extern "C" int samd_lmmse_matrix_c128(const double* h, const double* s, int64_t n, int m, int k, double* g, void* stream) {
  return lmmse_matrix<double>(h, s, n, m, k, s ? 0 : 1, g, stream);
}

extern "C" int samd_matrix_pinv_c64(const float* a, int64_t n, int m, int k, float* out, void* stream) {
  SAMD_REQUIRE(k <= m, "matrix_pinv: the matrix must have full column rank (K <= M)");
  return lmmse_matrix<float>(a, nullptr, n, m, k, 2, out, stream);
}
extern "C" int samd_matrix_pinv_c128(const double* a, int64_t n, int m, int k, double* out, void* stream) {
  SAMD_REQUIRE(k <= m, "matrix_pinv: the matrix must have full column rank (K <= M)");
  return lmmse_matrix<double>(a, nullptr, n, m, k, 2, out, stream);
}
