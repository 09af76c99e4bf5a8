// Per-resource-element LMMSE MIMO equalisation.
//
// Replaces (reference src/sionna/phy/):
//   lmmse_equalizer            mimo/equalization.py:101-233
//   whiten_channel             mimo/utils.py:292-356      (L = chol(S), yw = L^-1 y, Hw = L^-1 H)
//   lmmse_matrix               mimo/equalization.py:11-99 (G = (Hw^H Hw + I)^-1 Hw^H via Cholesky)
//   inv_cholesky               utils/linalg.py:8-32
//   OFDMEqualizer.call         ofdm/equalization.py:109-275 (layout shuffles, covariance build
//                              S = H_u H_u^H + diag(no) + diag(sum err_var), stream re-ordering,
//                              data-symbol gather)
//
// MI355X design: one lane owns one resource element and solves its M x K problem entirely in
// registers (two tiny complex Cholesky factorisations + triangular solves, sizes are template
// parameters).  The reference materialises S ([...,M,M], 940 MB at config C4), ~10 transposed
// copies of y / h_hat / err_var and launches batched-cholesky TF kernels; here the fused OFDM
// kernel reads y, h_hat (120 B per RE for 4x2) once in their API layout - consecutive lanes are
// consecutive subcarriers, so every load is coalesced - and writes x_hat / no_eff directly in the
// [batch, tx, stream, data symbol] order.  The contraction sizes (4x2: 128 real flops for the
// Gramian) are far below one MFMA tile per RE, so the matrix cores are not used: the kernel is
// HBM-streaming (arithmetic intensity ~6 flop/B).
#include "common.h"

namespace samd {

struct c32 { float re, im; };
__device__ __forceinline__ c32 C(float r, float i) { return c32{r, i}; }
__device__ __forceinline__ c32 operator+(c32 a, c32 b) { return C(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ c32 operator-(c32 a, c32 b) { return C(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ c32 operator*(c32 a, c32 b) { return C(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
__device__ __forceinline__ c32 mulc(c32 a, c32 b) { return C(a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im); }  // a * conj(b)
__device__ __forceinline__ c32 scale(c32 a, float s) { return C(a.re * s, a.im * s); }
__device__ __forceinline__ c32 cj(c32 a) { return C(a.re, -a.im); }
__device__ __forceinline__ c32 cdiv(c32 a, c32 b) {
  const float d = b.re * b.re + b.im * b.im;
  return C((a.re * b.re + a.im * b.im) / d, (a.im * b.re - a.re * b.im) / d);
}

// In-place lower Cholesky of a Hermitian positive definite N x N matrix (lower part used).
// After the call a[i][j], j<i is L_ij and a[j][j].re is L_jj (real, positive).
template <int N>
__device__ __forceinline__ void cholesky(c32 (&a)[N][N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float d = a[j][j].re;
#pragma unroll
    for (int k = 0; k < j; ++k) d -= a[j][k].re * a[j][k].re + a[j][k].im * a[j][k].im;
    const float l = sqrtf(d);
    a[j][j] = C(l, 0.f);
    const float inv = 1.f / l;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      c32 v = a[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = v - mulc(a[i][k], a[j][k]);
      a[i][j] = scale(v, inv);
    }
  }
}

// x_hat = diag(G H)^-1 G y, no_eff = Re(1/diag(G H) - 1)   (mimo/equalization.py:195-231)
template <int M, int K>
__device__ __forceinline__ void lmmse_solve(c32 (&y)[M], c32 (&h)[M][K], c32 (&s)[M][M], bool whiten, c32 (&xh)[K],
                                            float (&ne)[K]) {
  c32 g[K][M];
  if (whiten) {
    // whitening: L = chol(S); yw = L^-1 y; Hw = L^-1 H (forward substitution)
    cholesky<M>(s);
#pragma unroll
    for (int i = 0; i < M; ++i) {
      c32 v = y[i];
#pragma unroll
      for (int k = 0; k < i; ++k) v = v - s[i][k] * y[k];
      y[i] = scale(v, 1.f / s[i][i].re);
#pragma unroll
      for (int c = 0; c < K; ++c) {
        c32 w = h[i][c];
#pragma unroll
        for (int k = 0; k < i; ++k) w = w - s[i][k] * h[k][c];
        h[i][c] = scale(w, 1.f / s[i][i].re);
      }
    }
    // A = Hw^H Hw + I, C = chol(A), G = A^-1 Hw^H (cholesky_solve: forward then backward)
    c32 a[K][K];
#pragma unroll
    for (int i = 0; i < K; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        c32 v = C(i == j ? 1.f : 0.f, 0.f);
#pragma unroll
        for (int m = 0; m < M; ++m) v = v + mulc(h[m][j], h[m][i]);     // conj(h[m][i]) * h[m][j]
        a[i][j] = v;
      }
    cholesky<K>(a);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      c32 z[K];
#pragma unroll
      for (int i = 0; i < K; ++i) {                                     // C z = Hw^H e_m
        c32 v = cj(h[m][i]);
#pragma unroll
        for (int k = 0; k < i; ++k) v = v - a[i][k] * z[k];
        z[i] = scale(v, 1.f / a[i][i].re);
      }
#pragma unroll
      for (int i = K - 1; i >= 0; --i) {                                // C^H g = z
        c32 v = z[i];
#pragma unroll
        for (int k = i + 1; k < K; ++k) v = v - cj(a[k][i]) * g[k][m];
        g[i][m] = scale(v, 1.f / a[i][i].re);
      }
    }
  } else {
    // G = H^H (H H^H + S)^-1 : solve (H H^H + S) G^H = H, column by column
    c32 q[M][M];
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) {
        c32 v = s[i][j];
#pragma unroll
        for (int c = 0; c < K; ++c) v = v + mulc(h[i][c], h[j][c]);
        q[i][j] = v;
      }
    cholesky<M>(q);
#pragma unroll
    for (int c = 0; c < K; ++c) {
      c32 z[M];
#pragma unroll
      for (int i = 0; i < M; ++i) {
        c32 v = h[i][c];
#pragma unroll
        for (int k = 0; k < i; ++k) v = v - q[i][k] * z[k];
        z[i] = scale(v, 1.f / q[i][i].re);
      }
      c32 gt[M];
#pragma unroll
      for (int i = M - 1; i >= 0; --i) {
        c32 v = z[i];
#pragma unroll
        for (int k = i + 1; k < M; ++k) v = v - cj(q[k][i]) * gt[k];
        gt[i] = scale(v, 1.f / q[i][i].re);
      }
#pragma unroll
      for (int i = 0; i < M; ++i) g[c][i] = cj(gt[i]);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    c32 gy = C(0.f, 0.f), d = C(0.f, 0.f);
#pragma unroll
    for (int m = 0; m < M; ++m) { gy = gy + g[k][m] * y[m]; d = d + g[k][m] * h[m][k]; }
    xh[k] = cdiv(gy, d);
    ne[k] = cdiv(C(1.f, 0.f), d).re - 1.f;
  }
}

// ---- standalone lmmse_equalizer on [N,M], [N,M,K], [N,M,M]
template <int M, int K>
__global__ __launch_bounds__(128) void lmmse_items_kernel(const float2* __restrict__ y, const float2* __restrict__ h,
                                                          const float2* __restrict__ s, int64_t n, int whiten,
                                                          float2* __restrict__ x_hat, float* __restrict__ no_eff) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  c32 yy[M], hh[M][K], ss[M][M], xh[K];
  float ne[K];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const float2 v = y[i * M + m];
    yy[m] = C(v.x, v.y);
#pragma unroll
    for (int k = 0; k < K; ++k) { const float2 w = h[(i * M + m) * K + k]; hh[m][k] = C(w.x, w.y); }
#pragma unroll
    for (int j = 0; j < M; ++j) { const float2 w = s[(i * M + m) * M + j]; ss[m][j] = C(w.x, w.y); }
  }
  lmmse_solve<M, K>(yy, hh, ss, whiten != 0, xh, ne);
#pragma unroll
  for (int k = 0; k < K; ++k) { x_hat[i * K + k] = make_float2(xh[k].re, xh[k].im); no_eff[i * K + k] = ne[k]; }
}

// ---- fused OFDM LMMSE equaliser: one lane per (b, rx, t, f_eff)
struct OfdmEqArgs {
  const float2* y;        // [B, RX, M, T, FFT]
  const float2* h_hat;    // [B, RX, M, S, T, F]   S = num_tx * num_streams_per_tx
  const float* err_var;   // nullptr | [S, T*F] (ev_mode 1) | [B, RX, M, S, T*F] (ev_mode 2)
  const float* no;        // [B, RX, M]
  const int32_t* sc_ind;  // [F]    effective subcarrier -> fft bin
  const int32_t* desired; // [RX, K] global stream ids of the streams detected by each receiver
  const int32_t* undesired;  // [RX, U]
  const int32_t* data_pos;   // [S, T*F] index of the data symbol carried by an RE, or -1
  float2* x_hat;          // [B, S, ND]
  float* no_eff;          // [B, S, ND]
  int B, RX, S, T, F, FFT, U, ND, ev_mode, whiten;
};

template <int M, int K>
__global__ __launch_bounds__(128) void ofdm_lmmse_kernel(OfdmEqArgs p) {
  const int64_t total = (int64_t)p.B * p.RX * p.T * p.F;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int TF = p.T * p.F;
  const int re = (int)(i % TF);
  const int f = re % p.F, t = re / p.F;
  const int rx = (int)((i / TF) % p.RX);
  const int64_t b = i / ((int64_t)TF * p.RX);
  int dpos[K];
  bool any = false;
#pragma unroll
  for (int k = 0; k < K; ++k) { dpos[k] = p.data_pos[(int64_t)p.desired[rx * K + k] * TF + re]; any |= dpos[k] >= 0; }
  if (!any) return;                                           // pilot-only resource element
  const int64_t brx = b * p.RX + rx;
  c32 y[M], h[M][K], s[M][M], xh[K];
  float ne[K];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const float2 v = p.y[((brx * M + m) * p.T + t) * p.FFT + p.sc_ind[f]];
    y[m] = C(v.x, v.y);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float2 w = p.h_hat[((brx * M + m) * p.S + p.desired[rx * K + k]) * TF + re];
      h[m][k] = C(w.x, w.y);
    }
#pragma unroll
    for (int j = 0; j < M; ++j) s[m][j] = C(0.f, 0.f);
    // thermal noise + channel-estimation error of ALL streams (ofdm/equalization.py:204-217)
    float dg = p.no[brx * M + m];
    if (p.ev_mode == 1) { for (int q = 0; q < p.S; ++q) dg += p.err_var[(int64_t)q * TF + re]; }
    else if (p.ev_mode == 2) { for (int q = 0; q < p.S; ++q) dg += p.err_var[((brx * M + m) * p.S + q) * TF + re]; }
    s[m][m] = C(dg, 0.f);
  }
  for (int u = 0; u < p.U; ++u) {                             // interference of the undesired streams
    c32 hu[M];
    const int q = p.undesired[rx * p.U + u];
#pragma unroll
    for (int m = 0; m < M; ++m) { const float2 w = p.h_hat[((brx * M + m) * p.S + q) * TF + re]; hu[m] = C(w.x, w.y); }
#pragma unroll
    for (int a = 0; a < M; ++a)
#pragma unroll
      for (int c = 0; c <= a; ++c) s[a][c] = s[a][c] + mulc(hu[a], hu[c]);
  }
  lmmse_solve<M, K>(y, h, s, p.whiten != 0, xh, ne);
#pragma unroll
  for (int k = 0; k < K; ++k)
    if (dpos[k] >= 0) {
      const int64_t o = (b * p.S + p.desired[rx * K + k]) * p.ND + dpos[k];
      p.x_hat[o] = make_float2(xh[k].re, xh[k].im);
      p.no_eff[o] = ne[k];
    }
}

}  // namespace samd

using namespace samd;

#define SAMD_MK_LIST(X) X(1, 1) X(2, 1) X(2, 2) X(4, 1) X(4, 2) X(4, 4) X(8, 1) X(8, 2) X(8, 4)

extern "C" int samd_lmmse_equalizer_c64(const float* y, const float* h, const float* s, int64_t n, int m, int k,
                                        int whiten, float* x_hat, float* no_eff, void* stream) {
  SAMD_REQUIRE(y && h && s && x_hat && no_eff && n >= 0, "bad argument");
  if (n == 0) return SAMD_OK;
  const dim3 grid((unsigned)((n + 127) / 128));
#define X(M, K)                                                                                                    \
  if (m == M && k == K) {                                                                                          \
    hipLaunchKernelGGL((lmmse_items_kernel<M, K>), grid, dim3(128), 0, (hipStream_t)stream, (const float2*)y,     \
                       (const float2*)h, (const float2*)s, n, whiten, (float2*)x_hat, no_eff);                     \
    return launch_status();                                                                                        \
  }
  SAMD_MK_LIST(X)
#undef X
  set_error("lmmse_equalizer: unsupported (num_rx_ant, num_streams) combination");
  return SAMD_ERR_UNSUPPORTED;
}

extern "C" int samd_ofdm_lmmse_c64(const float* y, const float* h_hat, const float* err_var, int ev_mode,
                                   const float* no, const int32_t* sc_ind, const int32_t* desired,
                                   const int32_t* undesired, const int32_t* data_pos, int batch, int num_rx,
                                   int num_rx_ant, int num_streams_total, int streams_per_rx, int num_undesired,
                                   int num_ofdm_symbols, int num_eff_subcarriers, int fft_size, int num_data,
                                   int whiten, float* x_hat, float* no_eff, void* stream) {
  SAMD_REQUIRE(y && h_hat && no && sc_ind && desired && data_pos && x_hat && no_eff, "null argument");
  SAMD_REQUIRE(ev_mode >= 0 && ev_mode <= 2 && (ev_mode == 0 || err_var), "bad err_var mode");
  SAMD_REQUIRE(num_undesired == 0 || undesired, "undesired stream table missing");
  OfdmEqArgs p{(const float2*)y, (const float2*)h_hat, err_var, no, sc_ind, desired, undesired, data_pos,
               (float2*)x_hat, no_eff, batch, num_rx, num_streams_total, num_ofdm_symbols, num_eff_subcarriers,
               fft_size, num_undesired, num_data, ev_mode, whiten};
  const int64_t total = (int64_t)batch * num_rx * num_ofdm_symbols * num_eff_subcarriers;
  if (total == 0) return SAMD_OK;
  const dim3 grid((unsigned)((total + 127) / 128));
#define X(M, K)                                                                                            \
  if (num_rx_ant == M && streams_per_rx == K) {                                                            \
    hipLaunchKernelGGL((ofdm_lmmse_kernel<M, K>), grid, dim3(128), 0, (hipStream_t)stream, p);             \
    return launch_status();                                                                                \
  }
  SAMD_MK_LIST(X)
#undef X
  set_error("ofdm_lmmse: unsupported (num_rx_ant, streams_per_rx) combination");
  return SAMD_ERR_UNSUPPORTED;
}
