// precision = "double" (reference src/sionna/phy/block.py:25-52) for the OFDM link blocks of config C4: float64 /
// complex128 variants of
//   complex_normal / AWGN.call       utils/misc.py:19-54, channel/awgn.py:63-78
//   ResourceGridMapper.call          ofdm/resource_grid.py:394-412
//   TDL.__call__                     channel/tr38901/tdl.py:372-470
//   cir_to_ofdm_channel              channel/utils.py:180-253
//   ApplyOFDMChannel.call            channel/apply_ofdm_channel.py:70-80
//   LSChannelEstimator (+ NN)        ofdm/channel_estimation.py:138-173, 257-285, 364-435
//
// Same data layouts and the same Philox stream positions as the complex64 kernels (channel.hip, ofdm.hip); the
// arithmetic is double throughout, evaluated the direct way (one output element per lane, ascending accumulation
// order): double precision is a correctness feature of the interface, the tuned kernels are the float32 ones.
// The random draws take the SAME 24-bit uniforms as the float32 stream (exact in both formats) and evaluate
// Box-Muller / the affine maps in double, so a double-precision run sees the float32 run's realisation to float32
// rounding.  Specification: oracle/f64_ofdm.py.
#include "common.h"

namespace samd {
namespace {

using c128 = double2;

__device__ __forceinline__ c128 cmul(c128 a, c128 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__device__ __forceinline__ double u01d(uint32_t x) {
  return (double)(x >> 8) * 5.9604644775390625e-08 + 2.98023223876953125e-08;      // 2^-24, 2^-25: the float32 uniforms
}

__device__ __forceinline__ double unid(uint64_t seed, uint64_t call, uint64_t i, double lo, double hi) {
  const uint4 r = philox_block(seed, call, i >> 2);
  const uint32_t w = (i & 3) == 0 ? r.x : (i & 3) == 1 ? r.y : (i & 3) == 2 ? r.z : r.w;
  return lo + (hi - lo) * u01d(w);
}

constexpr double kPi = 3.14159265358979323846;

__device__ __forceinline__ c128 box_muller_d(uint32_t a, uint32_t b) {
  const double r = sqrt(-2.0 * log(u01d(a)));
  double s, c;
  sincos(6.283185307179586 * u01d(b), &s, &c);
  return make_double2(r * c, r * s);
}

// y[i] = x[i] (0 if x == nullptr) + sqrt(no / 2) * (w_re + j w_im); Philox block i/2 carries elements 2*blk, 2*blk + 1
__global__ __launch_bounds__(256) void awgn128_kernel(const c128* __restrict__ x, const double* __restrict__ no, int64_t no_len,
                                                      uint64_t seed, uint64_t call, int64_t n, c128* __restrict__ y) {
  const int64_t nblk = (n + 1) / 2;
  const double sh = sqrt(0.5);
  for (int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < nblk; blk += (int64_t)gridDim.x * blockDim.x) {
    const uint4 r = philox_block(seed, call, (uint64_t)blk);
    const c128 w[2] = {box_muller_d(r.x, r.y), box_muller_d(r.z, r.w)};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t i = blk * 2 + e;
      if (i < n) {
        const double s = sqrt(no_len == 1 ? no[0] : no[i]);
        const c128 xi = x ? x[i] : make_double2(0.0, 0.0);
        y[i] = make_double2(xi.x + (w[e].x * sh) * s, xi.y + (w[e].y * sh) * s);
      }
    }
  }
}

__global__ __launch_bounds__(256) void rg_map128_kernel(const c128* __restrict__ x, const c128* __restrict__ pilots,
                                                        const int32_t* __restrict__ data_pos, const int32_t* __restrict__ pilot_pos,
                                                        int64_t total, int S, int TF, int ND, int NP, c128* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int re = (int)(i % TF);
    const int64_t bs = i / TF;
    const int s = (int)(bs % S);
    const int d = data_pos[s * TF + re];
    c128 v = make_double2(0.0, 0.0);
    if (d >= 0) v = x[bs * ND + d];
    else {
      const int p = pilot_pos[s * TF + re];
      if (p >= 0) v = pilots[s * NP + p];
    }
    out[i] = v;
  }
}

// a[b, ra, ta, p, t]: one lane per output; the draws are those of tdl_cir_kernel (ofdm.hip): call+0 doppler[b],
// call+1 theta[b,p,n], call+2 phi[b,ra,ta,p,n], call+3 phi_0[b]
__global__ __launch_bounds__(256) void tdl_cir128_kernel(uint64_t seed, uint64_t call, int B, int RA, int TA, int P, int T, int N,
                                                         double sampling_frequency, const double* __restrict__ mean_powers,
                                                         double min_doppler, double max_doppler, int los, double los_power,
                                                         double los_aoa, c128* __restrict__ a) {
  const int64_t total = (int64_t)B * RA * TA * P * T;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(o % T);
    const int64_t i = o / T;                // ((b*RA + ra)*TA + ta)*P + p
    const int p = (int)(i % P);
    const int64_t b = i / ((int64_t)P * RA * TA);
    const double doppler = unid(seed, call, (uint64_t)b, min_doppler, max_doppler);
    const double ts = (double)t / sampling_frequency;
    double accx = 0.0, accy = 0.0;
    for (int n = 0; n < N; ++n) {
      const double theta = unid(seed, call + 1, (uint64_t)((b * P + p) * N + n), -kPi / (double)N, kPi / (double)N);
      const double phi = unid(seed, call + 2, (uint64_t)(i * N + n), -kPi, kPi);
      const double alpha = (2.0 * kPi / (double)N) * (double)(n + 1) + theta;
      double sn, cs;
      sincos(doppler * ts * cos(alpha) + phi, &sn, &cs);
      accx += cs; accy += sn;
    }
    const double k = sqrt(mean_powers[p]) / sqrt((double)N);
    c128 h = make_double2(k * accx, k * accy);
    if (los && p == 0) {
      const double phi0 = unid(seed, call + 3, (uint64_t)b, -kPi, kPi);
      double sn, cs;
      sincos(doppler * ts * cos(los_aoa) + phi0, &sn, &cs);
      const double kf = sqrt(los_power);
      h.x += cs * kf; h.y += sn * kf;
    }
    a[o] = h;
  }
}

// h_freq[b,rx,ra,tx,ta,t,f] = sum_p a[b,rx,ra,tx,ta,p,t] exp(-j 2 pi f_f tau[b,rx,tx,p])
__global__ __launch_bounds__(256) void cir_to_ofdm128_kernel(const c128* __restrict__ a, const double* __restrict__ tau,
                                                             const double* __restrict__ freq, int64_t total, int RX, int RA, int TX,
                                                             int TA, int P, int T, int F, c128* __restrict__ h) {
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(o % F);
    const int t = (int)((o / F) % T);
    const int64_t link = o / ((int64_t)F * T);          // (((b*RX + rx)*RA + ra)*TX + tx)*TA + ta
    const int tx = (int)((link / TA) % TX);
    const int64_t brx = link / ((int64_t)TA * TX * RA);  // b*RX + rx
    const double* tg = tau + (brx * TX + tx) * P;
    const c128* ap = a + link * P * T + t;
    const double w = -2.0 * kPi * freq[f];
    c128 acc = make_double2(0.0, 0.0);
    for (int p = 0; p < P; ++p) {
      double sn, cs;
      sincos(w * tg[p], &sn, &cs);
      const c128 v = cmul(ap[(int64_t)p * T], make_double2(cs, sn));
      acc.x += v.x; acc.y += v.y;
    }
    h[o] = acc;
  }
}

// normalisation (channel/utils.py:245-251): one workgroup per (b, rx, tx); c = mean over (ra, ta, t, f) of |h|^2 by a fixed-shape
// tree (deterministic), h *= 1/sqrt(c) (0 where c == 0: divide_no_nan)
__global__ __launch_bounds__(256) void c2o_normalize128_kernel(c128* __restrict__ h, int RX, int RA, int TX, int TA, int64_t TF) {
  __shared__ double red[256];
  const int grp = blockIdx.x;
  const int tx = grp % TX, rx = (grp / TX) % RX;
  const int64_t b = grp / (TX * RX);
  const int64_t cnt = (int64_t)RA * TA * TF;
  auto at = [&](int64_t i) -> c128* {
    const int64_t re = i % TF;
    const int ta = (int)((i / TF) % TA), ra = (int)(i / (TF * TA));
    return h + (((((b * RX + rx) * RA + ra) * TX + tx) * TA + ta) * TF + re);
  };
  double e = 0.0;
  for (int64_t i = threadIdx.x; i < cnt; i += blockDim.x) { const c128 v = *at(i); e += v.x * v.x + v.y * v.y; }
  red[threadIdx.x] = e;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double c = red[0] / (double)cnt;
  const double inv = c > 0.0 ? 1.0 / sqrt(c) : 0.0;
  for (int64_t i = threadIdx.x; i < cnt; i += blockDim.x) { c128* q = at(i); *q = make_double2(q->x * inv, q->y * inv); }
}

__global__ __launch_bounds__(256) void apply_ofdm_channel128_kernel(const c128* __restrict__ x, const c128* __restrict__ h,
                                                                    int64_t total, int RXA, int TXA, int TF, c128* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int re = (int)(i % TF);
    const int64_t br = i / TF;
    const int64_t b = br / RXA;
    c128 acc = make_double2(0.0, 0.0);
    for (int k = 0; k < TXA; ++k) {
      const c128 v = cmul(h[(br * TXA + k) * TF + re], x[(b * TXA + k) * TF + re]);
      acc.x += v.x; acc.y += v.y;
    }
    y[i] = acc;
  }
}

__global__ __launch_bounds__(256) void ls_gather_scale128_kernel(const c128* __restrict__ y, const int32_t* __restrict__ src,
                                                                 const c128* __restrict__ coef, int64_t total, int S, int N_out,
                                                                 int N_in, c128* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % N_out);
    const int64_t r = i / N_out;
    const int s = (int)(r % S);
    const int64_t bra = r / S;
    out[i] = cmul(y[bra * N_in + src[(int64_t)s * N_out + j]], coef[(int64_t)s * N_out + j]);
  }
}

inline int grid_for(int64_t n, int block) {
  const int64_t g = (n + block - 1) / block;
  return (int)std::min<int64_t>(std::max<int64_t>(g, 1), 256 * 32);
}

}  // namespace
}  // namespace samd

using namespace samd;

extern "C" int samd_awgn_c128(const double* x, const double* no, int64_t no_len, uint64_t seed, uint64_t call, int64_t n, double* y,
                              void* stream) {
  SAMD_REQUIRE(no && y && n >= 0, "bad argument");
  SAMD_REQUIRE(no_len == 1 || no_len == n, "no must be scalar or per element");
  if (n == 0) return SAMD_OK;
  hipLaunchKernelGGL(awgn128_kernel, dim3(grid_for((n + 1) / 2, 256)), dim3(256), 0, (hipStream_t)stream, (const c128*)x, no, no_len,
                     seed, call, n, (c128*)y);
  return launch_status();
}

extern "C" int samd_rg_map_c128(const double* x, const double* pilots, const int32_t* data_pos, const int32_t* pilot_pos, int batch,
                                int num_streams, int num_re, int num_data, int num_pilots, double* out, void* stream) {
  SAMD_REQUIRE(x && data_pos && pilot_pos && out && (pilots || num_pilots == 0), "null argument");
  const int64_t total = (int64_t)batch * num_streams * num_re;
  if (total == 0) return SAMD_OK;
  hipLaunchKernelGGL(rg_map128_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const c128*)x,
                     (const c128*)pilots, data_pos, pilot_pos, total, num_streams, num_re, num_data, num_pilots, (c128*)out);
  return launch_status();
}

extern "C" int samd_tdl_cir_c128(uint64_t seed, uint64_t call, int batch, int num_rx_ant, int num_tx_ant, int num_paths,
                                 int num_time_steps, int num_sinusoids, double sampling_frequency, const double* mean_powers,
                                 double min_doppler, double max_doppler, int los, double los_power, double los_aoa, double* a,
                                 void* stream) {
  SAMD_REQUIRE(mean_powers && a && batch > 0 && num_paths > 0 && num_time_steps > 0 && num_sinusoids > 0, "bad argument");
  const int64_t total = (int64_t)batch * num_rx_ant * num_tx_ant * num_paths * num_time_steps;
  hipLaunchKernelGGL(tdl_cir128_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, seed, call, batch, num_rx_ant,
                     num_tx_ant, num_paths, num_time_steps, num_sinusoids, sampling_frequency, mean_powers, min_doppler, max_doppler,
                     los, los_power, los_aoa, (c128*)a);
  return launch_status();
}

extern "C" int samd_cir_to_ofdm_c128(const double* a, const double* tau, const double* frequencies, int batch, int num_rx,
                                     int num_rx_ant, int num_tx, int num_tx_ant, int num_paths, int num_time_steps, int num_freqs,
                                     int normalize, double* h_freq, void* stream) {
  SAMD_REQUIRE(a && tau && frequencies && h_freq, "null argument");
  SAMD_REQUIRE(batch > 0 && num_rx > 0 && num_rx_ant > 0 && num_tx > 0 && num_tx_ant > 0 && num_paths > 0 && num_time_steps > 0 &&
                   num_freqs > 0, "bad size");
  const int64_t total = (int64_t)batch * num_rx * num_rx_ant * num_tx * num_tx_ant * num_time_steps * num_freqs;
  hipLaunchKernelGGL(cir_to_ofdm128_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const c128*)a, tau,
                     frequencies, total, num_rx, num_rx_ant, num_tx, num_tx_ant, num_paths, num_time_steps, num_freqs, (c128*)h_freq);
  if (normalize)
    hipLaunchKernelGGL(c2o_normalize128_kernel, dim3((unsigned)(batch * num_rx * num_tx)), dim3(256), 0, (hipStream_t)stream,
                       (c128*)h_freq, num_rx, num_rx_ant, num_tx, num_tx_ant, (int64_t)num_time_steps * num_freqs);
  return launch_status();
}

extern "C" int samd_apply_ofdm_channel_c128(const double* x, const double* h_freq, int batch, int num_rx_x_ant, int num_tx_x_ant,
                                            int num_re, double* y, void* stream) {
  SAMD_REQUIRE(x && h_freq && y, "null argument");
  const int64_t total = (int64_t)batch * num_rx_x_ant * num_re;
  if (total == 0) return SAMD_OK;
  hipLaunchKernelGGL(apply_ofdm_channel128_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const c128*)x,
                     (const c128*)h_freq, total, num_rx_x_ant, num_tx_x_ant, num_re, (c128*)y);
  return launch_status();
}

extern "C" int samd_ls_gather_scale_c128(const double* y, const int32_t* src, const double* coef, int rows, int num_streams, int n_out,
                                         int n_in, double* out, void* stream) {
  SAMD_REQUIRE(y && src && coef && out, "null argument");
  const int64_t total = (int64_t)rows * num_streams * n_out;
  if (total == 0) return SAMD_OK;
  hipLaunchKernelGGL(ls_gather_scale128_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const c128*)y, src,
                     (const c128*)coef, total, num_streams, n_out, n_in, (c128*)out);
  return launch_status();
}
