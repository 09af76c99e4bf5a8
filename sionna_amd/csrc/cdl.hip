// Clustered-delay-line channel model (3GPP TR 38.901 Sec. 7.7.1, Sec. 7.5 steps 10-11 without
// sub-clustering).
//   CDL.__call__                        /root/reference/src/sionna/phy/channel/tr38901/cdl.py:258-333
//   ChannelCoefficientsGenerator        /root/reference/src/sionna/phy/channel/tr38901/channel_coefficients.py:173-194,
//                                       459-1031 (phase matrix, field matrix, array offsets, Doppler, NLoS sum, LoS,
//                                       K-factor combination, ordering by delay)
//
// MI355X design.  The reference materialises [batch, clusters, rays, rx_ant, tx_ant, time] tensors for
// every factor of step 11.  Here everything that depends only on the model, the arrays and their
// orientations is tabulated once on the host for all 20 x 20 (zenith ray, azimuth ray) pairs of a
// cluster - the random coupling only permutes ray indices - and one thread owns one (batch, cluster,
// rx antenna, tx antenna) coefficient: it walks the 20 rays, draws their initial phases from the
// Philox stream, looks the field / array terms up (tables stay in L2) and accumulates the Doppler
// rotation over time in registers; every output is written once.
// RNG layout = oracle/cdl.py: call+0..2 speed / azimuth / zenith of the velocity [b]; call+3..6 sort
// keys of the AoA, AoD, ZoA, ZoD shuffles [b, n, m]; call+7 initial phases [b, n, m, 4].
#include "common.h"

namespace samd {
namespace {

constexpr int kRays = 20;

__device__ __forceinline__ uint32_t key32(uint64_t seed, uint64_t call, uint64_t i) {
  const uint4 r = philox_block(seed, call, i >> 2);
  return (i & 3) == 0 ? r.x : (i & 3) == 1 ? r.y : (i & 3) == 2 ? r.z : r.w;
}

// perm[b][n][type][r] = index of the ray with the r-th smallest key (stable): cdl.py:629-663
__global__ void cdl_coupling_kernel(uint64_t seed, uint64_t call, int64_t total /*B*N*4*/, int N,
                                    unsigned char* __restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int type = (int)(i & 3);
  const int64_t bn = i >> 2;                                  // b * N + n
  uint32_t key[kRays];
#pragma unroll
  for (int m = 0; m < kRays; ++m) key[m] = key32(seed, call + 3 + type, (uint64_t)bn * kRays + m);
#pragma unroll
  for (int m = 0; m < kRays; ++m) {
    int rank = 0;
#pragma unroll
    for (int j = 0; j < kRays; ++j) rank += (key[j] < key[m] || (key[j] == key[m] && j < m)) ? 1 : 0;
    perm[i * kRays + rank] = (unsigned char)m;
  }
}

__device__ __forceinline__ void sincos_r(float x, float* s, float* c) { sincosf(x, s, c); }
__device__ __forceinline__ void sincos_r(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ float sin_r(float x) { return sinf(x); }
__device__ __forceinline__ double sin_r(double x) { return sin(x); }
__device__ __forceinline__ float cos_r(float x) { return cosf(x); }
__device__ __forceinline__ double cos_r(double x) { return cos(x); }
// uniform draw in the block's precision from the float32 stream's 24-bit uniforms (exact in double)
template <typename R>
__device__ __forceinline__ R unir(uint64_t seed, uint64_t call, uint64_t i, R lo, R hi) {
  const uint32_t w = key32(seed, call, i);
  const R u = (R)(w >> 8) * (R)5.9604644775390625e-08 + (R)2.98023223876953125e-08;
  return lo + (hi - lo) * u;
}

template <typename R, typename R2>
struct CdlArgsT {
  uint64_t seed, call;
  int B, N, U, S, T;
  // tables, index ((n * 20 + zenith ray) * 20 + azimuth ray): fields per polarisation (theta, phi),
  // array responses per antenna, arrival unit vectors
  const R* f_rx;      // [N][20][20][2][2]
  const R* f_tx;      // [N][20][20][2][2]
  const R2* a_rx;     // [N][20][20][U]
  const R2* a_tx;     // [N][20][20][S]
  const R* r_rx;      // [N][20][20][3]
  const int32_t* pol_rx;  // [U]
  const int32_t* pol_tx;  // [S]
  const int32_t* order;   // [N] cluster of output position n (ascending delay)
  const R* amp;       // [N] sqrt(P_n / 20) (times sqrt(1/(K+1)) with a LoS path)
  const R* los;       // nullable: [2 pol][2] f_rx, [2][2] f_tx, [U] a_rx (c64), [S] a_tx (c64), [3] r_rx, [1] sqrt(K/(K+1))
  const unsigned char* perm;  // [B][N][4][20]
  R xpr_scale;        // sqrt(1 / kappa)
  R two_pi_over_lambda, sampling_frequency, min_speed, max_speed;
  R2* a;              // [B][U][S][N][T]
};

template <typename R, typename R2>
__global__ __launch_bounds__(128) void cdl_cir_kernel(CdlArgsT<R, R2> p) {
  const R pi = (R)3.14159265358979323846;
  const int64_t total = (int64_t)p.B * p.U * p.S * p.N;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int no = (int)(i % p.N);                              // output position
  const int s = (int)((i / p.N) % p.S);
  const int u = (int)((i / ((int64_t)p.N * p.S)) % p.U);
  const int64_t b = i / ((int64_t)p.N * p.S * p.U);
  const int n = p.order[no];
  // velocity of this batch example (cdl.py:262-280)
  const R v_r = unir<R>(p.seed, p.call, (uint64_t)b, p.min_speed, p.max_speed);
  const R v_phi = unir<R>(p.seed, p.call + 1, (uint64_t)b, (R)0, (R)2 * pi);
  const R v_th = unir<R>(p.seed, p.call + 2, (uint64_t)b, (R)0, pi);
  const R vx = v_r * cos_r(v_phi) * sin_r(v_th), vy = v_r * sin_r(v_phi) * sin_r(v_th), vz = v_r * cos_r(v_th);
  const unsigned char* pm = p.perm + ((b * p.N + n) * 4) * kRays;            // aoa, aod, zoa, zod
  const int pu = p.pol_rx[u], ps = p.pol_tx[s];
  R2* out = p.a + i * p.T;
  constexpr int kChunk = 16;
  for (int t0 = 0; t0 < p.T; t0 += kChunk) {
    R ax[kChunk], ay[kChunk];
#pragma unroll
    for (int k = 0; k < kChunk; ++k) ax[k] = ay[k] = (R)0;
    for (int m = 0; m < kRays; ++m) {
      const int ia = pm[m], ja = pm[kRays + m], iz = pm[2 * kRays + m], jz = pm[3 * kRays + m];
      const int64_t er = ((int64_t)n * kRays + iz) * kRays + ia, et = ((int64_t)n * kRays + jz) * kRays + ja;
      const R frt = p.f_rx[(er * 2 + pu) * 2], frp = p.f_rx[(er * 2 + pu) * 2 + 1];
      const R ftt = p.f_tx[(et * 2 + ps) * 2], ftp = p.f_tx[(et * 2 + ps) * 2 + 1];
      // phase matrix with the cross-polarisation power ratio (channel_coefficients.py:482-515)
      const uint64_t e4 = (((uint64_t)b * p.N + n) * kRays + m) * 4;
      R s0, c0, s1, c1, s2, c2, s3, c3;
      sincos_r(unir<R>(p.seed, p.call + 7, e4, -pi, pi), &s0, &c0);
      sincos_r(unir<R>(p.seed, p.call + 7, e4 + 1, -pi, pi), &s1, &c1);
      sincos_r(unir<R>(p.seed, p.call + 7, e4 + 2, -pi, pi), &s2, &c2);
      sincos_r(unir<R>(p.seed, p.call + 7, e4 + 3, -pi, pi), &s3, &c3);
      const R k = p.xpr_scale;
      // F_rx^T PM F_tx
      R cr = frt * (c0 * ftt + k * c1 * ftp) + frp * (k * c2 * ftt + c3 * ftp);
      R ci = frt * (s0 * ftt + k * s1 * ftp) + frp * (k * s2 * ftt + s3 * ftp);
      const R2 ar = p.a_rx[er * p.U + u], at = p.a_tx[et * p.S + s];
      const R gr = ar.x * at.x - ar.y * at.y, gi = ar.x * at.y + ar.y * at.x;
      const R hr = (cr * gr - ci * gi) * p.amp[n], hi = (cr * gi + ci * gr) * p.amp[n];
      const R w = p.two_pi_over_lambda * (p.r_rx[er * 3] * vx + p.r_rx[er * 3 + 1] * vy + p.r_rx[er * 3 + 2] * vz);
#pragma unroll
      for (int kk = 0; kk < kChunk; ++kk) {
        R sn, cs;
        sincos_r(w * ((R)(t0 + kk) / p.sampling_frequency), &sn, &cs);
        ax[kk] += hr * cs - hi * sn;
        ay[kk] += hr * sn + hi * cs;
      }
    }
    if (p.los && no == 0) {                                    // specular path into the first tap (:919-1031)
      const R* l = p.los;
      const R frt = l[pu * 2], frp = l[pu * 2 + 1], ftt = l[4 + ps * 2], ftp = l[4 + ps * 2 + 1];
      const R c = frt * ftt - frp * ftp;                   // PM = diag(1, -1)
      const R2 ar = reinterpret_cast<const R2*>(l + 8)[u];
      const R2 at = reinterpret_cast<const R2*>(l + 8 + 2 * p.U)[s];
      const R* rr = l + 8 + 2 * p.U + 2 * p.S;
      const R kf = rr[3];
      const R gr = (ar.x * at.x - ar.y * at.y) * c * kf, gi = (ar.x * at.y + ar.y * at.x) * c * kf;
      const R w = p.two_pi_over_lambda * (rr[0] * vx + rr[1] * vy + rr[2] * vz);
#pragma unroll
      for (int kk = 0; kk < kChunk; ++kk) {
        R sn, cs;
        sincos_r(w * ((R)(t0 + kk) / p.sampling_frequency), &sn, &cs);
        ax[kk] += gr * cs - gi * sn;
        ay[kk] += gr * sn + gi * cs;
      }
    }
#pragma unroll
    for (int kk = 0; kk < kChunk; ++kk)
      if (t0 + kk < p.T) out[t0 + kk] = R2{ax[kk], ay[kk]};
  }
}

}  // namespace
}  // namespace samd

using namespace samd;

extern "C" size_t samd_cdl_workspace_bytes(int batch, int num_clusters) {
  if (batch <= 0 || num_clusters <= 0) return 0;
  return (size_t)batch * num_clusters * 4 * kRays + 256;
}

extern "C" int samd_cdl_cir_c64(uint64_t seed, uint64_t call, int batch, int num_clusters, int num_rx_ant,
                                int num_tx_ant, int num_time_steps, float sampling_frequency, const float* f_rx,
                                const float* f_tx, const float* a_rx, const float* a_tx, const float* r_rx,
                                const int32_t* pol_rx, const int32_t* pol_tx, const int32_t* order, const float* amp,
                                const float* los, float xpr_scale, float two_pi_over_lambda, float min_speed,
                                float max_speed, void* workspace, size_t workspace_bytes, float* a, void* stream) {
  SAMD_REQUIRE(f_rx && f_tx && a_rx && a_tx && r_rx && pol_rx && pol_tx && order && amp && a, "null argument");
  SAMD_REQUIRE(batch > 0 && num_clusters > 0 && num_rx_ant > 0 && num_tx_ant > 0 && num_time_steps > 0, "bad shape");
  if (!workspace || workspace_bytes < samd_cdl_workspace_bytes(batch, num_clusters)) {
    set_error("workspace too small");
    return SAMD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  unsigned char* perm = reinterpret_cast<unsigned char*>(align_up((size_t)workspace, 256));
  const int64_t nperm = (int64_t)batch * num_clusters * 4;
  cdl_coupling_kernel<<<(unsigned)((nperm + 127) / 128), 128, 0, st>>>(seed, call, nperm, num_clusters, perm);
  if (int rc = launch_status()) return rc;
  CdlArgsT<float, float2> p{seed, call, batch, num_clusters, num_rx_ant, num_tx_ant, num_time_steps, f_rx, f_tx, (const float2*)a_rx,
            (const float2*)a_tx, r_rx, pol_rx, pol_tx, order, amp, los, perm, xpr_scale, two_pi_over_lambda,
            sampling_frequency, min_speed, max_speed, (float2*)a};
  const int64_t total = (int64_t)batch * num_rx_ant * num_tx_ant * num_clusters;
  SAMD_REQUIRE((total + 127) / 128 < (1ll << 31), "grid too large");
  cdl_cir_kernel<float, float2><<<(unsigned)((total + 127) / 128), 128, 0, st>>>(p);
  return launch_status();
}

// precision = "double" (reference block.py:25-52): the same kernel on float64 tables; the draws are the float32 stream's uniforms
extern "C" int samd_cdl_cir_c128(uint64_t seed, uint64_t call, int batch, int num_clusters, int num_rx_ant, int num_tx_ant,
                                 int num_time_steps, double sampling_frequency, const double* f_rx, const double* f_tx,
                                 const double* a_rx, const double* a_tx, const double* r_rx, const int32_t* pol_rx,
                                 const int32_t* pol_tx, const int32_t* order, const double* amp, const double* los, double xpr_scale,
                                 double two_pi_over_lambda, double min_speed, double max_speed, void* workspace,
                                 size_t workspace_bytes, double* a, void* stream) {
  SAMD_REQUIRE(f_rx && f_tx && a_rx && a_tx && r_rx && pol_rx && pol_tx && order && amp && a, "null argument");
  SAMD_REQUIRE(batch > 0 && num_clusters > 0 && num_rx_ant > 0 && num_tx_ant > 0 && num_time_steps > 0, "bad shape");
  if (!workspace || workspace_bytes < samd_cdl_workspace_bytes(batch, num_clusters)) {
    set_error("workspace too small");
    return SAMD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  unsigned char* perm = reinterpret_cast<unsigned char*>(align_up((size_t)workspace, 256));
  const int64_t nperm = (int64_t)batch * num_clusters * 4;
  cdl_coupling_kernel<<<(unsigned)((nperm + 127) / 128), 128, 0, st>>>(seed, call, nperm, num_clusters, perm);
  if (int rc = launch_status()) return rc;
  CdlArgsT<double, double2> p{seed, call, batch, num_clusters, num_rx_ant, num_tx_ant, num_time_steps, f_rx, f_tx, (const double2*)a_rx,
                              (const double2*)a_tx, r_rx, pol_rx, pol_tx, order, amp, los, perm, xpr_scale, two_pi_over_lambda,
                              sampling_frequency, min_speed, max_speed, (double2*)a};
  const int64_t total = (int64_t)batch * num_rx_ant * num_tx_ant * num_clusters;
  SAMD_REQUIRE((total + 127) / 128 < (1ll << 31), "grid too large");
  cdl_cir_kernel<double, double2><<<(unsigned)((total + 127) / 128), 128, 0, st>>>(p);
  return launch_status();
}
