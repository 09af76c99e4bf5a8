// precision = "double" (reference src/sionna/phy/block.py:25-52) for the time-domain channel: float64 variants of
//   cir_to_time_channel    channel/utils.py:256-349
//   ApplyTimeChannel.call  channel/apply_time_channel.py:85-137
// Layouts of samd_cir_to_time_c64 / samd_apply_time_channel_c64 (csrc/ofdm_time.hip).  One output per lane, ascending path / tap
// order; the normalisation is a second pass (the deferred-scale form of the float32 kernels is a throughput feature).  Held to
// oracle/f64_ofdm.py at 1e-9; the tuned kernels are the float32 ones.
#include "common.h"

namespace samd {
namespace {

constexpr double kPi64 = 3.14159265358979323846;

__device__ __forceinline__ double sinc64(double x) {       // tf.experimental.numpy.sinc: sin(pi x) / (pi x), 1 at 0
  if (x == 0.0) return 1.0;
  const double y = kPi64 * x;
  return sin(y) / y;
}

// h[b,rx,ra,tx,ta,t,l] = sum_p a[b,rx,ra,tx,ta,p,t] sinc(l_min + l - tau[b,rx,tx,p] W)
__global__ __launch_bounds__(256) void cir_to_time128_kernel(const double2* __restrict__ a, const double* __restrict__ tau, double bandwidth,
                                                             int l_min, int L, int64_t total, int RX, int RA, int TX, int TA, int P, int T,
                                                             double2* __restrict__ h) {
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int l = (int)(o % L);
    const int t = (int)((o / L) % T);
    const int64_t link = o / ((int64_t)L * T);             // (((b*RX + rx)*RA + ra)*TX + tx)*TA + ta
    const int tx = (int)((link / TA) % TX);
    const int64_t brx = link / ((int64_t)TA * TX * RA);
    const double* tg = tau + (brx * TX + tx) * P;
    const double2* ap = a + link * P * T + t;
    double re = 0.0, im = 0.0;
    for (int p = 0; p < P; ++p) {
      const double w = sinc64((double)(l_min + l) - tg[p] * bandwidth);
      const double2 v = ap[(int64_t)p * T];
      re += v.x * w; im += v.y * w;
    }
    h[o] = make_double2(re, im);
  }
}

// utils.py:337-347: c = mean over (ra, ta, t) of sum_l |h|^2 per (b, rx, tx); h *= 1 / sqrt(c) (0 where c == 0)
__global__ __launch_bounds__(256) void time_normalize128_kernel(double2* __restrict__ h, int RX, int RA, int TX, int TA, int64_t TL, int T) {
  __shared__ double red[256];
  const int grp = blockIdx.x;
  const int tx = grp % TX, rx = (grp / TX) % RX;
  const int64_t b = grp / (TX * RX);
  const int64_t cnt = (int64_t)RA * TA * TL;
  auto at = [&](int64_t i) -> double2* {
    const int64_t e = i % TL;
    const int ta = (int)((i / TL) % TA), ra = (int)(i / (TL * TA));
    return h + (((((b * RX + rx) * RA + ra) * TX + tx) * TA + ta) * TL + e);
  };
  double en = 0.0;
  for (int64_t i = threadIdx.x; i < cnt; i += blockDim.x) { const double2 v = *at(i); en += v.x * v.x + v.y * v.y; }
  red[threadIdx.x] = en;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const double c = red[0] / (double)((int64_t)RA * TA * T);
  const double inv = c > 0.0 ? 1.0 / sqrt(c) : 0.0;
  for (int64_t i = threadIdx.x; i < cnt; i += blockDim.x) { double2* q = at(i); *q = make_double2(q->x * inv, q->y * inv); }
}

// y[b, rxa, t] = sum_{txa} sum_l h[b, rxa, txa, t, l] x[b, txa, t - l], 0 <= t - l < Tn
__global__ __launch_bounds__(256) void apply_time128_kernel(const double2* __restrict__ x, const double2* __restrict__ h, int64_t total,
                                                            int RXA, int TXA, int Tn, int L, double2* __restrict__ y) {
  const int Tout = Tn + L - 1;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(o % Tout);
    const int64_t brxa = o / Tout;
    const int64_t b = brxa / RXA;
    const int lo = t - (Tn - 1) > 0 ? t - (Tn - 1) : 0, hi = t < L - 1 ? t : L - 1;
    double re = 0.0, im = 0.0;
    for (int k = 0; k < TXA; ++k) {
      const double2* hp = h + ((brxa * TXA + k) * Tout + t) * L;
      const double2* xp = x + (b * TXA + k) * Tn;
      for (int l = lo; l <= hi; ++l) {
        const double2 hv = hp[l], xv = xp[t - l];
        re += hv.x * xv.x - hv.y * xv.y;
        im += hv.x * xv.y + hv.y * xv.x;
      }
    }
    y[o] = make_double2(re, im);
  }
}

inline int grid_for_t(int64_t n, int block) {
  const int64_t g = (n + block - 1) / block;
  return (int)std::min<int64_t>(std::max<int64_t>(g, 1), 256 * 32);
}

}  // namespace
}  // namespace samd

using namespace samd;

extern "C" int samd_cir_to_time_c128(double bandwidth, const double* a, const double* tau, int l_min, int l_max, int batch, int num_rx,
                                     int num_rx_ant, int num_tx, int num_tx_ant, int num_paths, int num_time_steps, int normalize,
                                     double* h_time, void* stream) {
  SAMD_REQUIRE(a && tau && h_time && batch > 0 && l_max >= l_min && num_paths > 0 && num_time_steps > 0, "bad argument");
  const int L = l_max - l_min + 1;
  const int64_t total = (int64_t)batch * num_rx * num_rx_ant * num_tx * num_tx_ant * num_time_steps * L;
  hipLaunchKernelGGL(cir_to_time128_kernel, dim3(grid_for_t(total, 256)), dim3(256), 0, (hipStream_t)stream, (const double2*)a, tau,
                     bandwidth, l_min, L, total, num_rx, num_rx_ant, num_tx, num_tx_ant, num_paths, num_time_steps, (double2*)h_time);
  if (normalize)
    hipLaunchKernelGGL(time_normalize128_kernel, dim3((unsigned)(batch * num_rx * num_tx)), dim3(256), 0, (hipStream_t)stream,
                       (double2*)h_time, num_rx, num_rx_ant, num_tx, num_tx_ant, (int64_t)num_time_steps * L, num_time_steps);
  return launch_status();
}

extern "C" int samd_apply_time_channel_c128(const double* x, const double* h_time, int batch, int num_rx, int num_rx_ant, int num_tx,
                                            int num_tx_ant, int num_time_samples, int l_tot, double* y, void* stream) {
  SAMD_REQUIRE(x && h_time && y && batch >= 0 && num_time_samples > 0 && l_tot > 0, "bad argument");
  const int64_t total = (int64_t)batch * num_rx * num_rx_ant * (num_time_samples + l_tot - 1);
  if (total == 0) return SAMD_OK;
  // h [B, rx, ra, tx, ta, Tout, L]: for a fixed (b, rx, ra) the (tx, ta) links are contiguous -> one index k over num_tx * num_tx_ant
  hipLaunchKernelGGL(apply_time128_kernel, dim3(grid_for_t(total, 256)), dim3(256), 0, (hipStream_t)stream, (const double2*)x,
                     (const double2*)h_time, total, num_rx * num_rx_ant, num_tx * num_tx_ant, num_time_samples, l_tot, (double2*)y);
  return launch_status();
}
