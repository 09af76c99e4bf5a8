// OFDM resource-grid plumbing, frequency-domain channel generation and application.
//
// Replaces (reference src/sionna/phy/):
//   ResourceGridMapper.call          ofdm/resource_grid.py:394-412     (tf.scatter_nd x2 + transposes)
//   ResourceGridDemapper.call        ofdm/resource_grid.py:466-520     (gathers + transposes)
//   RemoveNulledSubcarriers.call     ofdm/resource_grid.py:551-552
//   TDL.__call__                     channel/tr38901/tdl.py:372-470    (sum-of-sinusoids taps)
//   cir_to_ofdm_channel              channel/utils.py:180-253          (taps -> frequency response)
//   ApplyOFDMChannel.call            channel/apply_ofdm_channel.py:70-80
//
// All of these are HBM-streaming or trig-bound element-wise kernels; what the reference
// materialises as rank-8 broadcast temporaries (3.4 GB for the TDL sinusoids, 12.8 GB for
// the per-path frequency responses at config C4) stays in registers / LDS here.
#include "common.h"
#include "options.h"

namespace samd {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// out[b, s, re] = x[b, s, data_pos[s,re]] | pilots[s, pilot_pos[s,re]] | 0
__global__ __launch_bounds__(256) void rg_map_kernel(const float2* __restrict__ x, const float2* __restrict__ pilots,
                                                     const int32_t* __restrict__ data_pos,
                                                     const int32_t* __restrict__ pilot_pos, int64_t total, int S,
                                                     int TF, int ND, int NP, float2* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int re = (int)(i % TF);
    const int64_t bs = i / TF;
    const int s = (int)(bs % S);
    const int d = data_pos[s * TF + re];
    float2 v = make_float2(0.f, 0.f);
    if (d >= 0) v = x[bs * ND + d];
    else {
      const int p = pilot_pos[s * TF + re];
      if (p >= 0) v = pilots[s * NP + p];
    }
    out[i] = v;
  }
}

// out[b, g, j] = in[b, src_group[g], idx[g, j]]   (element = EL floats: 2 for complex64, 1 for float32)
template <int EL>
__global__ __launch_bounds__(256) void gather3_kernel(const float* __restrict__ in, const int32_t* __restrict__ src_group,
                                                      const int32_t* __restrict__ idx, int64_t total, int G_in,
                                                      int N_in, int G_out, int N_out, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % N_out);
    const int64_t bg = i / N_out;
    const int g = (int)(bg % G_out);
    const int64_t b = bg / G_out;
    const int64_t src = ((b * G_in + src_group[g]) * N_in + idx[(int64_t)g * N_out + j]) * EL;
#pragma unroll
    for (int e = 0; e < EL; ++e) out[i * EL + e] = in[src + e];
  }
}

// ---- TDL taps.  One thread per (b, ra, ta, p); loops over time steps and sinusoids.
// RNG layout = oracle/ofdm.py::tdl_cir: call+0 doppler[b], call+1 theta[b,p,n], call+2
// phi[b,ra,ta,p,n], call+3 phi_0[b]; element i of a call = word i%4 of Philox block i/4.
__device__ __forceinline__ float uni(uint64_t seed, uint64_t call, uint64_t i, float lo, float hi) {
  const uint4 r = philox_block(seed, call, i >> 2);
  const uint32_t w = (i & 3) == 0 ? r.x : (i & 3) == 1 ? r.y : (i & 3) == 2 ? r.z : r.w;
  return lo + (hi - lo) * u01(w);
}

// consecutive elements of one stream: the Philox block of four words is computed once per four elements, not once per element
struct UniStream {
  uint64_t seed, call, blk = ~0ull;
  uint4 r;
  __device__ __forceinline__ UniStream(uint64_t s, uint64_t c) : seed(s), call(c) { r = make_uint4(0u, 0u, 0u, 0u); }
  __device__ __forceinline__ float operator()(uint64_t i, float lo, float hi) {
    if ((i >> 2) != blk) { blk = i >> 2; r = philox_block(seed, call, blk); }
    const uint32_t w = (i & 3) == 0 ? r.x : (i & 3) == 1 ? r.y : (i & 3) == 2 ? r.z : r.w;
    return lo + (hi - lo) * u01(w);
  }
};

typedef float tdl_f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void tdl_cir_kernel(uint64_t seed, uint64_t call, int B, int RA, int TA, int P,
                                                      int T, int N, float sampling_frequency,
                                                      const float* __restrict__ mean_powers, float min_doppler,
                                                      float max_doppler, int los, float los_power, float los_aoa,
                                                      float2* __restrict__ a) {
  const int64_t total = (int64_t)B * RA * TA * P;
  const float pi = 3.14159265358979323846f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    const int64_t r1 = i / P;            // (b*RA + ra)*TA + ta
    const int64_t b = r1 / ((int64_t)RA * TA);
    const float doppler = uni(seed, call, (uint64_t)b, min_doppler, max_doppler);
    const float amp = sqrtf(mean_powers[p]);
    const float norm = 1.f / sqrtf((float)N);
    float2* o = a + i * T;
    // time steps in register-resident chunks: every tap is accumulated over the sinusoids in
    // registers (in sinusoid order, as the oracle sums) and written exactly once.
    // Per sinusoid the phase advances by a constant w = doppler cos(alpha) / fs per step: the chunk's first
    // step is evaluated directly (the reference's float32 argument), the other 15 by rotating with
    // (cos w, sin w) - |error| < 2e-6, below the float32 rounding of the reference's own arguments (~1e-5 at
    // arguments of ~1e2) - 2 instead of 16 sincos per sinusoid and chunk.  The per-sinusoid random draws
    // (Philox) are made once per thread, not once per chunk.
    constexpr int kChunk = 16;
    constexpr int kMaxSin = 24;                                    // register-resident draws up to this many sinusoids
    float ca_r[kMaxSin], ph_r[kMaxSin];
    const bool cached = N <= kMaxSin;
    if (cached) {
      UniStream u_theta(seed, call + 1), u_phi(seed, call + 2);
#pragma unroll
      for (int n = 0; n < kMaxSin; ++n)
        if (n < N) {
          const float theta = u_theta((uint64_t)((b * P + p) * N + n), -pi / (float)N, pi / (float)N);
          ph_r[n] = u_phi((uint64_t)(i * N + n), -pi, pi);
          ca_r[n] = cosf((2.f * pi / (float)N) * (float)(n + 1) + theta);
        }
    }
    for (int t0 = 0; t0 < T; t0 += kChunk) {
      // (cos, sin) pairs in packed registers (round 6): accumulate = one v_pk_add_f32, rotate = two v_pk_mul_f32 + one
      // v_pk_add_f32 - (c, s) cw + (-s, c) sw, the same products and the same sum as c cw - s sw / s cw + c sw, no contraction -
      // instead of eight scalar operations per step; the steps beyond T of the last chunk are not computed
      tdl_f32x2 acc[kChunk];
#pragma unroll
      for (int k = 0; k < kChunk; ++k) acc[k] = tdl_f32x2{0.f, 0.f};
      const int kmax = T - t0 < kChunk ? T - t0 : kChunk;
      auto one = [&](float ca, float phi) {
        float sn, cs, sw, cw;
        sincosf(doppler * ((float)t0 / sampling_frequency) * ca + phi, &sn, &cs);
        sincosf(doppler * (1.f / sampling_frequency) * ca, &sw, &cw);
        tdl_f32x2 v = {cs, sn};
        const tdl_f32x2 cw2 = {cw, cw}, sw2 = {sw, sw};
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
          if (k < kmax) {
            acc[k] += v;
            const tdl_f32x2 rot = {-v.y, v.x};
            v = v * cw2 + rot * sw2;
          }
        }
      };
      if (cached) {
#pragma unroll
        for (int n = 0; n < kMaxSin; ++n)
          if (n < N) one(ca_r[n], ph_r[n]);
      } else {
        for (int n = 0; n < N; ++n) {
          const float theta = uni(seed, call + 1, (uint64_t)((b * P + p) * N + n), -pi / (float)N, pi / (float)N);
          const float phi = uni(seed, call + 2, (uint64_t)(i * N + n), -pi, pi);
          one(cosf((2.f * pi / (float)N) * (float)(n + 1) + theta), phi);
        }
      }
      float phi0 = 0.f;
      if (los && p == 0) phi0 = uni(seed, call + 3, (uint64_t)b, -pi, pi);
#pragma unroll
      for (int k = 0; k < kChunk; ++k) {
        const int t = t0 + k;
        if (t < T) {
          float2 h = make_float2(amp * (acc[k].x * norm), amp * (acc[k].y * norm));
          if (los && p == 0) {
            const float arg = doppler * ((float)t / sampling_frequency) * cosf(los_aoa) + phi0;
            float sn, cs;
            sincosf(arg, &sn, &cs);
            const float kf = sqrtf(los_power);
            h.x += cs * kf; h.y += sn * kf;
          }
          o[t] = h;
        }
      }
    }
  }
}

// ---- taps -> frequency response (cir_to_ofdm_channel, channel/utils.py:180-253).  One workgroup per (b, rx, tx).
// The P x F phase table e^{-j 2 pi f tau_p} is built once per workgroup in LDS; then a thread owns ONE subcarrier f
// (threads = G groups x F subcarriers) and keeps its column of the table in registers (MAXP complex values), the G
// groups split the (ra, ta, t) rows, the taps a[.., p, t] of a row are the same address for all lanes of a group
// (broadcast loads) and every store is a contiguous run of subcarriers.  The first version looped over outputs with
// the table read from LDS per product, the taps gathered per lane from L1 and unfused multiply / add: 1.06 ms per 8192
// batch items of config C4; this one 0.84 ms (phase table with sincosf, 46 dependent packed FMAs per output).  The accumulation order over the paths (ascending) is unchanged.
// The optional normalisation (unit mean energy over ra, ta, t, f) is a deterministic in-block reduction as before.
typedef float c2o_f32x2 __attribute__((ext_vector_type(2)));

// TAPS_LDS = false: the taps of a link do not fit beside the phase table (many antenna pairs / clusters / symbols) and
// are read from global memory (the lanes of a group read the same address: one L1 broadcast).  MAXP = 0: more than 64
// paths - the phase column stays in LDS and the path loop has a run-time trip count.  Same products, same ascending
// order: the variants return identical values (tests/test_gpu_ofdm.py::test_cir_to_ofdm_large_links).
template <int MAXP, bool TAPS_LDS>
__global__ __launch_bounds__(256) void cir_to_ofdm_kernel(const float2* __restrict__ a, const float* __restrict__ tau,
                                                          const float* __restrict__ freqs, int RX, int RA, int TX,
                                                          int TA, int P, int T, int F, int normalize,
                                                          float2* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float2 tab[];   // [P][F] phases, taps [RA*TA][P][T], red[256]
  float2* taps = tab + (size_t)P * F;
  float* red = reinterpret_cast<float*>(taps + (TAPS_LDS ? (size_t)RA * TA * P * T : 0));
  const int tx = blockIdx.x % TX;
  const int rx = (blockIdx.x / TX) % RX;
  const int b = blockIdx.x / (TX * RX);
  const float* tb = tau + ((size_t)(b * RX + rx) * TX + tx) * P;
  const int nt = blockDim.x;
  for (int i = threadIdx.x; i < P * F; i += nt) {
    const int p = i / F, f = i % F;
    float sn, cs;
    sincosf(-2.f * 3.14159265358979323846f * freqs[f] * tb[p], &sn, &cs);
    tab[i] = make_float2(cs, sn);
  }
  // the taps of this (b, rx, tx): RA*TA contiguous runs of P*T values -> LDS (every tap is used by all F subcarriers)
  const int pt = P * T;
  if (TAPS_LDS)
    for (int i = threadIdx.x; i < RA * TA * pt; i += nt) {
      const int q = i % pt, ta = (i / pt) % TA, ra = i / (pt * TA);
      taps[i] = a[((((size_t)(b * RX + rx) * RA + ra) * TX + tx) * TA + ta) * (size_t)pt + q];
    }
  __syncthreads();
  const int rows = RA * TA * T;                                   // (ra, ta, t) rows of F outputs each
  const int G = F <= nt ? nt / F : 1;                             // row groups working side by side
  float energy = 0.f;
  for (int f0 = 0; f0 < F; f0 += nt) {                            // one trip unless F > blockDim
    const int g = F <= nt ? (int)threadIdx.x / F : 0;
    const int f = F <= nt ? (int)threadIdx.x % F : f0 + (int)threadIdx.x;
    const bool act = g < G && f < F;
    float2 ph[MAXP > 0 ? MAXP : 1];
#pragma unroll
    for (int p = 0; p < MAXP; ++p) ph[p] = (act && p < P) ? tab[p * F + f] : make_float2(0.f, 0.f);
    for (int row = g; act && row < rows; row += G) {
      const int t = row % T, ta = (row / T) % TA, ra = row / (T * TA);
      // same address for the lanes of a group: LDS (or L1) broadcast
      const float2* ap = TAPS_LDS ? taps + (size_t)(ra * TA + ta) * pt + t
                                  : a + ((((size_t)(b * RX + rx) * RA + ra) * TX + tx) * TA + ta) * (size_t)pt + t;
      // h += a * ph as two packed fused multiply-adds per path: (h.x, h.y) += a.x * (ph.x, ph.y); += a.y * (-ph.y, ph.x)
      // (the library is built with -ffp-contract=off for the bit-exact decoders; this kernel is held to 1e-4 against
      // the float64 oracle, and the fused form is the more accurate one)
      c2o_f32x2 hv = {0.f, 0.f};
      if constexpr (MAXP > 0) {
#pragma unroll
        for (int p = 0; p < MAXP; ++p)
          if (p < P) {
            const float2 av = ap[p * T];
            hv = __builtin_elementwise_fma(c2o_f32x2{av.x, av.x}, c2o_f32x2{ph[p].x, ph[p].y}, hv);
            hv = __builtin_elementwise_fma(c2o_f32x2{av.y, av.y}, c2o_f32x2{-ph[p].y, ph[p].x}, hv);
          }
      } else {
        for (int p = 0; p < P; ++p) {
          const float2 av = ap[p * T], pv = tab[p * F + f];
          hv = __builtin_elementwise_fma(c2o_f32x2{av.x, av.x}, c2o_f32x2{pv.x, pv.y}, hv);
          hv = __builtin_elementwise_fma(c2o_f32x2{av.y, av.y}, c2o_f32x2{-pv.y, pv.x}, hv);
        }
      }
      out[(((((size_t)(b * RX + rx) * RA + ra) * TX + tx) * TA + ta) * T + t) * F + f] = make_float2(hv.x, hv.y);
      energy += hv.x * hv.x + hv.y * hv.y;
    }
    if (F <= nt) break;
  }
  if (!normalize) return;
  // (recomputing the sums in a second pass instead of re-reading the result was measured slower: 1.09 vs 0.84 ms -
  // the workgroup's 68 KB of output are still in L2 when they are scaled)
  const int per = rows * F;
  red[threadIdx.x] = energy;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o && (int)threadIdx.x + o < nt) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float c = sqrtf(red[0] / (float)per);
  const float inv = c > 0.f ? 1.f / c : 0.f;                  // divide_no_nan
  float2* ob = out + ((size_t)(b * RX + rx) * RA) * TX * TA * (size_t)T * F;
  for (int i = threadIdx.x; i < per; i += nt) {
    const int f = i % F;
    const int t = (i / F) % T;
    const int ta = (i / (F * T)) % TA;
    const int ra = i / (F * T * TA);
    float2* o = ob + ((((size_t)ra * TX + tx) * TA + ta) * T + t) * F + f;
    *o = make_float2(o->x * inv, o->y * inv);
  }
}

// Round 5: the same transform with the workgroup's results staged in REGISTERS (config C4: 8512 complex values = 68 KB per
// (batch, rx, tx) link).  The kernel above writes H, re-reads it from L2 and writes it again scaled (WRITE_SIZE 1.99 x the
// output, profiles/r04zz_pmc), spends ~100 vector instructions of index arithmetic (three integer divisions) per output
// next to the 46 packed FMAs of a 23-path link, and walks the paths through 24 scalar branches per row (`p < P`).  Here:
// taps and phases are padded with zeros to MAXP paths (no branch); a thread owns subcarrier f and every G-th row, its RPT
// rows are an unrolled loop over a register array (independent FMA chains, (ra, ta, t) advanced by carries - no division
// in the loops); the energy is reduced once and every value is stored ONCE, scaled.  Same products per output, summed in
// ascending path order in two accumulators (real-tap and imaginary-tap contributions); held to 1e-4 of the float64 oracle
// like the kernel above.
template <int MAXP, int RPT>
__global__ __launch_bounds__(512) void cir_to_ofdm_reg_kernel(const float2* __restrict__ a, const float* __restrict__ tau,
                                                              const float* __restrict__ freqs, int RX, int RA, int TX,
                                                              int TA, int P, int T, int F, int normalize,
                                                              float2* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float2 tab[];   // [MAXP][F] phases, taps [RA*TA][T][MAXP] (paths adjacent), red[blockDim]
  float2* taps = tab + (size_t)MAXP * F;
  float* red = reinterpret_cast<float*>(taps + (size_t)RA * TA * MAXP * T);
  const int tx = blockIdx.x % TX;
  const int rx = (blockIdx.x / TX) % RX;
  const int b = blockIdx.x / (TX * RX);
  const float* tb = tau + ((size_t)(b * RX + rx) * TX + tx) * P;
  const int nt = blockDim.x, tid = threadIdx.x;
  for (int i = tid; i < MAXP * F; i += nt) {
    const int p = i / F, f = i - p * F;
    float sn = 0.f, cs = 0.f;
    if (p < P) sincosf(-2.f * 3.14159265358979323846f * freqs[f] * tb[p], &sn, &cs);
    tab[i] = make_float2(cs, sn);
  }
  const int mt = MAXP * T;
  for (int i = tid; i < RA * TA * mt; i += nt) {
    const int lk = i / mt, q = i - lk * mt;                      // link (ra, ta), q = t * MAXP + p: a row's paths are adjacent
    const int t = q / MAXP, p = q - t * MAXP;
    const int ta = lk % TA, ra = lk / TA;
    taps[i] = p < P ? a[((((size_t)(b * RX + rx) * RA + ra) * TX + tx) * TA + ta) * (size_t)(P * T) + (size_t)p * T + t] : make_float2(0.f, 0.f);
  }
  __syncthreads();
  const int rows = RA * TA * T;
  const int G = nt / F;                                           // row groups side by side (host: F <= blockDim)
  const int g = tid / F, f = tid - g * F;
  const bool act = g < G;
  float2 ph[MAXP];
#pragma unroll
  for (int p = 0; p < MAXP; ++p) ph[p] = act ? tab[p * F + f] : make_float2(0.f, 0.f);
  c2o_f32x2 acc[RPT];
  float energy = 0.f;
  {
    int t = g % T, lk = g / T;                                    // row g = (link lk, symbol t); advanced by G with carries
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      c2o_f32x2 hv = {0.f, 0.f};
      if (act && lk < RA * TA) {
        const float2* ap = taps + (size_t)(lk * T + t) * MAXP;
        // a e^{j phi} = (a.x c - a.y s, a.x s + a.y c) as two accumulators: u += a.x (c, s), v += a.y (s, c) - the swapped
        // pair is an operand selection of the packed FMA, no negation per path - and h = (u.x - v.x, u.y + v.y) at the end
        c2o_f32x2 u = {0.f, 0.f}, v = {0.f, 0.f};
#pragma unroll
        for (int p = 0; p < MAXP; ++p) {
          const float2 av = ap[p];                                // the lanes of a group read one address: LDS broadcast
          u = __builtin_elementwise_fma(c2o_f32x2{av.x, av.x}, c2o_f32x2{ph[p].x, ph[p].y}, u);
          v = __builtin_elementwise_fma(c2o_f32x2{av.y, av.y}, c2o_f32x2{ph[p].y, ph[p].x}, v);
        }
        hv = c2o_f32x2{u.x - v.x, u.y + v.y};
        energy += hv.x * hv.x + hv.y * hv.y;
      }
      acc[r] = hv;
      t += G;
      while (t >= T) { t -= T; ++lk; }
    }
  }
  float inv = 1.f;
  if (normalize) {
    red[tid] = energy;
    __syncthreads();
    int o = 1;
    while (o < nt) o <<= 1;
    for (o >>= 1; o > 0; o >>= 1) {
      if (tid < o && tid + o < nt) red[tid] += red[tid + o];
      __syncthreads();
    }
    const float c = sqrtf(red[0] / (float)(rows * F));
    inv = c > 0.f ? 1.f / c : 0.f;                               // divide_no_nan
  }
  float2* ob = out + ((size_t)(b * RX + rx) * RA) * TX * TA * (size_t)T * F;
  {
    int t = g % T, lk = g / T;
    int ta = lk % TA, ra = lk / TA;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      if (act && ra < RA)
        ob[(unsigned)((((ra * TX + tx) * TA + ta) * T + t) * F + f)] = normalize ? make_float2(acc[r].x * inv, acc[r].y * inv)
                                                                              : make_float2(acc[r].x, acc[r].y);
      t += G;
      while (t >= T) { t -= T; if (++ta == TA) { ta = 0; ++ra; } }
    }
  }
}

// Box-Muller on two uniforms in (0,1): channel.hip's box_muller, operation for operation (the fused channel kernel adds the noise
// awgn_kernel would add)
__device__ __forceinline__ float2 c2o_box_muller(uint32_t a, uint32_t b) {
  const float r = sqrtf(-2.0f * logf(u01(a)));
  const float t = 6.283185307179586f * u01(b);
  float sn, cs;
  sincosf(t, &sn, &cs);
  return make_float2(r * cs, r * sn);
}

// The same kernel with the paths walked in passes of PW: only PW phases (and the PW taps of a row, and with PF those of the
// next row, requested a row ahead) are live beside the RPT staged results, so the kernel needs 80-94 instead of 164 vector
// registers and three to four workgroups instead of two (at 384 threads) share a CU.  History on the C4 shapes (8192 links
// of 4 x 2 antennas, 23 paths, 14 x 76 grid; tools/c2o_ab.py, profiles/r05o*): all phases in registers 688 us -> passes of 8
// 450 us -> every batch of staging loads issued back to back before its first use, passes of 4 with look-ahead 330-357 us
// (a plain fill of the 558 MB output: 88 us).  Where the rest goes (ablation build, profiles/r05o8_c2o_ab.txt; kernel-only
// times through the C-ABI): the FMA loop 165 us (92 us at the packed-FMA issue rate: the LDS broadcast reads and the waits
// behind them), the stores 47, the taps gather 37, sincosf 15 - and ~80 us that no part owns: a workgroup is one link, its
// life is a chain of dependent round trips (tau -> sincos -> LDS -> barrier -> FMA -> store), and a CU holds three of them.
// The vector pipe was 60 % busy with 3.4 k instructions per wave of which 1.15 k are the packed FMAs in the first pass
// version (profiles/r05p): predication of the rows beyond the link, 64-bit address arithmetic, integer divisions in the
// staging loops, two accumulators per row and pass.  Here: the tap array is padded with zero rows to RPT * G rows (no
// predicate in the FMA loop), a result is ONE accumulator updated by two packed FMAs per path (h += a.x (c, s) + a.y (-s, c)),
// the staging loops decompose indices with multiply-high, the energy is reduced with wave shuffles.  Sum over the paths in
// ascending order, pass by pass (the same chain for every PW: the variants are bit-identical to each other); held to 1e-4 of
// the float64 oracle like the kernels above.
//
// FUSED (round 6; OFDMChannel.call channel/ofdm_channel.py:109-115 when nobody reads h_freq): the link's frequency response
// never leaves the workgroup - ApplyOFDMChannel (apply_ofdm_channel.py:70-80) and the AWGN of channel/awgn.py:63-78 happen on
// the staged registers: y[b, rx, ra, t, f] = sum_ta h x[b, 0, ta, t, f] + sqrt(no) w, one transmitter (TX = 1; the sum over
// transmit antennas runs inside the workgroup through the LDS of the dead phase / tap tables, ascending ta like
// apply_ofdm_channel_kernel), the noise of element i from Philox block i / 2 like awgn_kernel: the same bits as the three
// separate kernels, without the 558 MB of h_freq written and read back at config C4.
// 0: the fused kernel is built without the in-kernel noise (measured slower than awgn_kernel in place on y, profiles/r06g): the
// host adds it with samd_awgn_c64; the entry refuses a noise variance then.  1 keeps the bit-identical in-kernel form.
#ifndef C2O_FUSED_NOISE
#define C2O_FUSED_NOISE 0
#endif
struct C2oFuse {
  const float2* x;       // [B, 1, TA, T, F] transmitted grid
  const float* no;       // one noise variance, or null: no noise
  uint64_t seed, call;   // the AWGN block's Philox stream
  float2* y;             // [B, RX, RA, T, F]
};
template <int MAXP, int RPT, int PW, bool PF, bool FUSED>
__device__ __forceinline__ void cir_to_ofdm_pass_body(const float2* __restrict__ a, const float* __restrict__ tau,
                                                      const float* __restrict__ freqs, int RX, int RA, int TX,
                                                      int TA, int P, int T, int F, int normalize,
                                                      float2* __restrict__ out, C2oFuse fu) {
  static_assert(MAXP % PW == 0, "pass width must divide the padded path count");
  extern __shared__ __attribute__((aligned(16))) float2 tab[];   // [MAXP][F] phases, taps [RPT * G][MAXP] (zero rows behind the link), red[8]
  const int nt = blockDim.x, tid = threadIdx.x;
  const int G = nt / F;                                           // row groups side by side (host: F <= blockDim)
  const int rows = RA * TA * T, rows_pad = RPT * G;
  float2* taps = tab + (size_t)MAXP * F;
  float* red = reinterpret_cast<float*>(taps + (size_t)rows_pad * MAXP);
  const int tx = blockIdx.x % TX;
  const int rx = (blockIdx.x / TX) % RX;
  const int b = blockIdx.x / (TX * RX);
  const float* tb = tau + ((size_t)(b * RX + rx) * TX + tx) * P;
  const int g0 = tid / F, f = tid - g0 * F;
  const bool act = g0 < G;
  const int g = act ? g0 : 0;                                     // the spare lanes of the block shadow group 0 and store nothing
  // Staging.  A workgroup of this kernel lives ~100 k cycles of which the FMA loop is ~5 k per wave: what it waits for are the
  // dependent global round trips of the staging loops.  So every batch of loads is issued back to back into registers
  // (KB independent requests per thread) before the first of them is used: one round trip per batch instead of one per element.
  constexpr int KB = 8;
  auto magic = [](unsigned d) { return d > 1 ? 0xFFFFFFFFu / d + 1u : 0u; };        // n / d = umulhi(n, magic(d)) for n, d < 2^13
  auto divu = [](unsigned n, unsigned d, unsigned m) { return d > 1 ? __umulhi(n, m) : n; };
  {
    const float wf = -2.f * 3.14159265358979323846f * freqs[f];
    for (int pb = g0; pb < MAXP; pb += KB * G) {                  // phases: thread (g, f) fills the rows p = g, g + G, ...
      float tv[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const int p = pb + k * G;
        tv[k] = (act && p < P) ? tb[p] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const int p = pb + k * G;
        if (act && p < MAXP) {
          float sn = 0.f, cs = 0.f;
          if (p < P) sincosf(wf * tv[k], &sn, &cs);
          tab[p * F + f] = make_float2(cs, sn);
        }
      }
    }
  }
  // Which rows a thread owns (round 6).  grouped: group g owns the UNITS u = (ra, t) with u mod G = g and of each all TA transmit
  // antennas, local row r = (u / G) TA + ta - the sum over the transmit antennas of ApplyOFDMChannel then stays inside a thread
  // (the fused kernel needs no exchange of products).  Needs RPT a multiple of TA; else rows g, g + G, ... of (ra, ta, t) order.
  const bool grouped = (RPT % TA) == 0;
  if (grouped) {
    for (int i = tid; i < rows_pad * MAXP; i += nt) taps[i] = make_float2(0.f, 0.f);
    __syncthreads();
  }
  {
    // taps: source order (ra | ta, p, t) with t fastest - consecutive lanes on consecutive addresses - transposed into [row][p]
    const unsigned pt = (unsigned)(P * T), lpt = (unsigned)TA * pt, total = (unsigned)RA * lpt;
    const unsigned m_lpt = magic(lpt), m_pt = magic(pt), m_t = magic((unsigned)T);
    const unsigned m_ta = magic((unsigned)TA), m_g = magic((unsigned)G);
    const float2* src = a + (((size_t)(b * RX + rx) * RA) * TX + tx) * (size_t)lpt;
    const size_t ra_stride = (size_t)TX * lpt;
    for (unsigned i0 = (unsigned)tid; i0 < total; i0 += (unsigned)(KB * nt)) {
      float2 v[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const unsigned i = i0 + (unsigned)(k * nt);
        if (i < total) {
          const unsigned ra = divu(i, lpt, m_lpt);
          v[k] = src[ra * ra_stride + (i - ra * lpt)];
        }
      }
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const unsigned i = i0 + (unsigned)(k * nt);
        if (i < total) {
          const unsigned lk = divu(i, pt, m_pt), q = i - lk * pt;
          const unsigned pp = divu(q, (unsigned)T, m_t), t = q - pp * (unsigned)T;
          unsigned row = lk * (unsigned)T + t;
          if (grouped) {                                          // unit u = (ra, t) -> group u mod G, local rows (u / G) TA + ta
            const unsigned ra = divu(lk, (unsigned)TA, m_ta), ta = lk - ra * (unsigned)TA;
            const unsigned u = ra * (unsigned)T + t, j = divu(u, (unsigned)G, m_g);
            row = (u - j * (unsigned)G) + (j * (unsigned)TA + ta) * (unsigned)G;
          }
          taps[row * MAXP + pp] = v[k];
        }
      }
    }
    if (!grouped) {
      const int zp = MAXP - P;                                    // zero columns of the real rows, zero rows behind them
      for (int i = tid; i < rows * zp; i += nt) {
        const int row = i / zp;
        taps[(unsigned)(row * MAXP + P + (i - row * zp))] = make_float2(0.f, 0.f);
      }
      for (int i = rows * MAXP + tid; i < rows_pad * MAXP; i += nt) taps[i] = make_float2(0.f, 0.f);
    }
  }
  __syncthreads();
  c2o_f32x2 acc[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) acc[r] = c2o_f32x2{0.f, 0.f};
  const unsigned rstride = (unsigned)(G * MAXP);
#pragma unroll 1
  for (int p0 = 0; p0 < MAXP; p0 += PW) {
    float2 ph[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) ph[p] = tab[(p0 + p) * F + f];
    unsigned ai = (unsigned)(g * MAXP + p0);
    asm volatile("" : "+v"(ai));          // (the RPT tap addresses of a thread are re-derived per pass, not kept in RPT registers)
    if constexpr (PF) {
      float2 cur[PW], nxt[PW];                                    // the taps of row r + 1 are requested before row r is summed
#pragma unroll
      for (int p = 0; p < PW; ++p) cur[p] = taps[ai + p];         // the lanes of a group read one address: LDS broadcast
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        ai += rstride;
        if (r + 1 < RPT) {
#pragma unroll
          for (int p = 0; p < PW; ++p) nxt[p] = taps[ai + p];
        }
        c2o_f32x2 h = acc[r];
#pragma unroll
        for (int p = 0; p < PW; ++p) {
          h = __builtin_elementwise_fma(c2o_f32x2{cur[p].x, cur[p].x}, c2o_f32x2{ph[p].x, ph[p].y}, h);
          h = __builtin_elementwise_fma(c2o_f32x2{cur[p].y, cur[p].y}, c2o_f32x2{-ph[p].y, ph[p].x}, h);
        }
        acc[r] = h;
#pragma unroll
        for (int p = 0; p < PW; ++p) cur[p] = nxt[p];
      }
    } else {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {                             // taps read where they are used: fewest registers, most waves
        const float2* ap = taps + ai;
        c2o_f32x2 h = acc[r];
#pragma unroll
        for (int p = 0; p < PW; ++p) {
          const float2 av = ap[p];
          h = __builtin_elementwise_fma(c2o_f32x2{av.x, av.x}, c2o_f32x2{ph[p].x, ph[p].y}, h);
          h = __builtin_elementwise_fma(c2o_f32x2{av.y, av.y}, c2o_f32x2{-ph[p].y, ph[p].x}, h);
        }
        acc[r] = h;
        ai += rstride;
      }
    }
  }
  float inv = 1.f;
  if (normalize) {
    float energy = 0.f;
    if (act) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) energy += acc[r].x * acc[r].x + acc[r].y * acc[r].y;  // rows behind the link are zero
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) energy += __shfl_xor(energy, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = energy;
    __syncthreads();
    float e = 0.f;
    for (int w = 0; w < (nt + 63) / 64; ++w) e += red[w];
    const float c = sqrtf(e / (float)(rows * F));
    inv = c > 0.f ? 1.f / c : 0.f;                               // divide_no_nan
  }
  // local row r of group g -> (ra, ta, t): a running state (r is a compile-time index of the result registers, TA is not)
  struct RowWalk {
    int u, ta, t, ra;          // grouped: unit, antenna;  else: (t, ta, ra) of row g + r G
    bool grouped;
    int G_, T_, TA_;
    unsigned mT;
    __device__ __forceinline__ void init(int g_, bool gr, int Gq, int Tq, int TAq, unsigned m) {
      grouped = gr; G_ = Gq; T_ = Tq; TA_ = TAq; mT = m;
      if (gr) { u = g_; ta = 0; ra = (int)(Tq > 1 ? __umulhi((unsigned)g_, m) : (unsigned)g_); t = g_ - ra * Tq; }
      else { u = 0; t = g_ % Tq; ta = (g_ / Tq) % TAq; ra = (g_ / Tq) / TAq; }
    }
    __device__ __forceinline__ void step() {
      if (grouped) {
        const bool w = ++ta == TA_;                               // wave-uniform
        if (w) {
          ta = 0; u += G_;
          ra = (int)(T_ > 1 ? __umulhi((unsigned)u, mT) : (unsigned)u); t = u - ra * T_;
        }
      } else {
        t += G_;
        while (t >= T_) { t -= T_; if (++ta == TA_) { ta = 0; ++ra; } }
      }
    }
  };
  const unsigned m_T = magic((unsigned)T);
  if constexpr (FUSED) {
    // y = sum over the transmit antennas of h x (+ noise): the group's units keep their TA rows in this thread's registers, x of
    // this batch item (TA T F values, the same for every receive antenna) comes through the LDS of the dead phase / tap tables -
    // one coalesced load per workgroup instead of 24 scattered ones per thread (host: grouped, x fits the tables' LDS)
    float2* xs = tab;
    const float2* xb = fu.x + (size_t)b * TA * T * F;
    __syncthreads();                                              // every wave is through its FMA loop
    {
      const int nx = TA * T * F;
      for (int i0x = tid; i0x < nx; i0x += KB * nt) {
        float2 xv[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) xv[k] = xb[min(i0x + k * nt, nx - 1)];
#pragma unroll
        for (int k = 0; k < KB; ++k)
          if (i0x + k * nt < nx) xs[i0x + k * nt] = xv[k];
      }
    }
    __syncthreads();
    const float sh = sqrtf(1.0f / 2.0f);
    const float sn = (C2O_FUSED_NOISE && fu.no) ? sqrtf(fu.no[0]) : 0.f;
    (void)sh; (void)sn;
    float2* yb = fu.y + ((size_t)(b * RX + rx) * RA) * (size_t)T * F;
    const uint64_t i0 = ((uint64_t)(b * RX + rx) * RA) * (uint64_t)T * F;
    (void)i0;
    RowWalk w;
    w.init(g, true, G, T, TA, m_T);
    float2 sy = make_float2(0.f, 0.f);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const float2 hv = make_float2(acc[r].x * inv, acc[r].y * inv);            // the value cir_to_ofdm stores
      const float2 v = cmul(hv, xs[(unsigned)((w.ta * T + w.t) * F + f)]);
      // apply_ofdm_channel_kernel: acc = 0; acc += v_k, k ascending.  A BRANCH on the wave-uniform antenna index (the empty asm
      // keeps it one): as two selections it read a vcc written by s_cselect_b64 - ~24 cycles of the vector pipe each on gfx950
      // (profiles/r06w_valu_rate2.txt), 48 per staged result
      if (w.ta == 0) { asm volatile(""); sy = make_float2(0.f, 0.f); }
      sy.x += v.x; sy.y += v.y;
      if (w.ta == TA - 1 && act && w.ra < RA) {
        const unsigned rel = (unsigned)((w.ra * T + w.t) * F + f);
        float2 so = sy;
        if (C2O_FUSED_NOISE && fu.no) {                           // awgn_kernel: element i takes half (i & 1) of Philox block i / 2
          const uint64_t i = i0 + rel;
          const uint4 rr = philox_block(fu.seed, fu.call, i >> 1);
          const float2 wn = (i & 1) ? c2o_box_muller(rr.z, rr.w) : c2o_box_muller(rr.x, rr.y);
          so = make_float2(so.x + (wn.x * sh) * sn, so.y + (wn.y * sh) * sn);
        }
        yb[rel] = so;
      }
      w.step();
    }
    return;
  }
  float2* ob = out + ((size_t)(b * RX + rx) * RA) * TX * TA * (size_t)T * F;
  {
    RowWalk w;
    w.init(g, grouped, G, T, TA, m_T);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      if (act && w.ra < RA)
        ob[(unsigned)((((w.ra * TX + tx) * TA + w.ta) * T + w.t) * F + f)] = make_float2(acc[r].x * inv, acc[r].y * inv);
      w.step();
    }
  }
}

template <int MAXP, int RPT, int PW, bool PF>
__global__ __launch_bounds__(512) void cir_to_ofdm_pass_kernel(const float2* __restrict__ a, const float* __restrict__ tau,
                                                               const float* __restrict__ freqs, int RX, int RA, int TX,
                                                               int TA, int P, int T, int F, int normalize,
                                                               float2* __restrict__ out, C2oFuse fu) {
  cir_to_ofdm_pass_body<MAXP, RPT, PW, PF, false>(a, tau, freqs, RX, RA, TX, TA, P, T, F, normalize, out, fu);
}
// the fused form keeps the occupancy of the plain one (5 waves per SIMD = three workgroups of 384 threads per CU - two were
// measured 30 % slower, profiles/r05w): the Philox / Box-Muller epilogue may spill around the 48 staged result registers
template <int MAXP, int RPT, int PW, bool PF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(5, 5)))
void cir_to_ofdm_fused_kernel(const float2* __restrict__ a, const float* __restrict__ tau, const float* __restrict__ freqs, int RX,
                              int RA, int TX, int TA, int P, int T, int F, int normalize, float2* __restrict__ out, C2oFuse fu) {
  cir_to_ofdm_pass_body<MAXP, RPT, PW, PF, true>(a, tau, freqs, RX, RA, TX, TA, P, T, F, normalize, out, fu);
}

// y[b,rx,ra,t,f] = sum_{tx,ta} h[b,rx,ra,tx,ta,t,f] * x[b,tx,ta,t,f]
__global__ __launch_bounds__(256) void apply_ofdm_channel_kernel(const float2* __restrict__ x,
                                                                 const float2* __restrict__ h, int64_t total,
                                                                 int RXA, int TXA, int TF,
                                                                 float2* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int re = (int)(i % TF);
    const int64_t br = i / TF;             // b*RXA + rxa
    const int64_t b = br / RXA;
    float2 acc = make_float2(0.f, 0.f);
    for (int k = 0; k < TXA; ++k) {
      const float2 v = cmul(h[(br * TXA + k) * TF + re], x[(b * TXA + k) * TF + re]);
      acc.x += v.x; acc.y += v.y;
    }
    y[i] = acc;
  }
}

// LS estimate + (nearest-neighbour) spreading: out[b, rxa, s, j] = y[b, rxa, src[s,j]] * coef[s,j]
__global__ __launch_bounds__(256) void ls_gather_scale_kernel(const float2* __restrict__ y,
                                                              const int32_t* __restrict__ src,
                                                              const float2* __restrict__ coef, int64_t total, int S,
                                                              int N_out, int N_in, float2* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % N_out);
    const int64_t r = i / N_out;
    const int s = (int)(r % S);
    const int64_t bra = r / S;
    out[i] = cmul(y[bra * N_in + src[(int64_t)s * N_out + j]], coef[(int64_t)s * N_out + j]);
  }
}

// TDL spatial correlation (channel/tr38901/tdl.py:474-492): for every (batch, path, time step) the vector of the
// num_rx_ant x num_tx_ant antenna-pair coefficients (rx major) is multiplied by the square root of the spatial
// correlation matrix, out[i] = sum_j mat[i][j] a[j] (Kronecker case: mat = sqrt(R_rx) (x) conj(sqrt(R_tx)), built on the
// host).  a / out [B, 1, RA, 1, TA, inner] with inner = num_paths * num_time_steps; one lane per output element,
// consecutive lanes are consecutive inner positions (coalesced); ascending-j accumulation.
template <typename R2, typename R>
__global__ __launch_bounds__(256) void spatial_corr_kernel(const R2* __restrict__ a, const R2* __restrict__ mat, int batch,
                                                           int ra, int ta, int64_t inner, R2* __restrict__ out) {
  const int n = ra * ta;
  const int64_t total = (int64_t)batch * n * inner;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = idx % inner;
    const int i = (int)((idx / inner) % n);
    const int64_t b = idx / (inner * n);
    const R2* src = a + b * n * inner + q;
    R re = (R)0, im = (R)0;
    for (int j = 0; j < n; ++j) {
      const R2 m = mat[i * n + j], v = src[(int64_t)j * inner];
      re += m.x * v.x - m.y * v.y;
      im += m.x * v.y + m.y * v.x;
    }
    out[idx] = R2{re, im};
  }
}

}  // namespace samd

using namespace samd;

static inline int grid_for(int64_t n, int block) {
  const int64_t g = (n + block - 1) / block;
  return (int)std::min<int64_t>(std::max<int64_t>(g, 1), 256 * 32);
}

extern "C" int samd_rg_map_c64(const float* x, const float* pilots, const int32_t* data_pos, const int32_t* pilot_pos,
                               int batch, int num_streams, int num_re, int num_data, int num_pilots, float* out,
                               void* stream) {
  SAMD_REQUIRE(x && data_pos && pilot_pos && out && (pilots || num_pilots == 0), "null argument");
  const int64_t total = (int64_t)batch * num_streams * num_re;
  if (total == 0) return SAMD_OK;
  hipLaunchKernelGGL(rg_map_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)x,
                     (const float2*)pilots, data_pos, pilot_pos, total, num_streams, num_re, num_data, num_pilots,
                     (float2*)out);
  return launch_status();
}

extern "C" int samd_gather3(const float* in, const int32_t* src_group, const int32_t* idx, int batch, int groups_in,
                            int n_in, int groups_out, int n_out, int floats_per_elem, float* out, void* stream) {
  SAMD_REQUIRE(in && src_group && idx && out, "null argument");
  SAMD_REQUIRE(floats_per_elem == 1 || floats_per_elem == 2 || floats_per_elem == 4,
               "element must be float32 (1), complex64 / float64 (2) or complex128 (4 floats)");
  const int64_t total = (int64_t)batch * groups_out * n_out;
  if (total == 0) return SAMD_OK;
  if (floats_per_elem == 4)
    hipLaunchKernelGGL(gather3_kernel<4>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in, src_group,
                       idx, total, groups_in, n_in, groups_out, n_out, out);
  else if (floats_per_elem == 2)
    hipLaunchKernelGGL(gather3_kernel<2>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in, src_group,
                       idx, total, groups_in, n_in, groups_out, n_out, out);
  else
    hipLaunchKernelGGL(gather3_kernel<1>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, in, src_group,
                       idx, total, groups_in, n_in, groups_out, n_out, out);
  return launch_status();
}

extern "C" int samd_tdl_cir_c64(uint64_t seed, uint64_t call, int batch, int num_rx_ant, int num_tx_ant, int num_paths,
                                int num_time_steps, int num_sinusoids, float sampling_frequency,
                                const float* mean_powers, float min_doppler, float max_doppler, int los,
                                float los_power, float los_aoa, float* a, void* stream) {
  SAMD_REQUIRE(mean_powers && a && batch > 0 && num_paths > 0 && num_time_steps > 0 && num_sinusoids > 0, "bad argument");
  const int64_t total = (int64_t)batch * num_rx_ant * num_tx_ant * num_paths;
  hipLaunchKernelGGL(tdl_cir_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, seed, call, batch,
                     num_rx_ant, num_tx_ant, num_paths, num_time_steps, num_sinusoids, sampling_frequency, mean_powers,
                     min_doppler, max_doppler, los, los_power, los_aoa, (float2*)a);
  return launch_status();
}

extern "C" int samd_spatial_corr_c64(const float* a, const float* mat, int batch, int num_rx_ant, int num_tx_ant, int64_t inner,
                                     float* out, void* stream) {
  SAMD_REQUIRE(a && mat && out && a != out, "bad argument (out must not alias a)");
  SAMD_REQUIRE(batch > 0 && num_rx_ant > 0 && num_tx_ant > 0 && inner > 0, "bad size");
  const int64_t total = (int64_t)batch * num_rx_ant * num_tx_ant * inner;
  hipLaunchKernelGGL((spatial_corr_kernel<float2, float>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)a,
                     (const float2*)mat, batch, num_rx_ant, num_tx_ant, inner, (float2*)out);
  return launch_status();
}

// precision = "double" (reference block.py:25-52)
extern "C" int samd_spatial_corr_c128(const double* a, const double* mat, int batch, int num_rx_ant, int num_tx_ant, int64_t inner,
                                      double* out, void* stream) {
  SAMD_REQUIRE(a && mat && out && a != out, "bad argument (out must not alias a)");
  SAMD_REQUIRE(batch > 0 && num_rx_ant > 0 && num_tx_ant > 0 && inner > 0, "bad size");
  const int64_t total = (int64_t)batch * num_rx_ant * num_tx_ant * inner;
  hipLaunchKernelGGL((spatial_corr_kernel<double2, double>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const double2*)a, (const double2*)mat, batch, num_rx_ant, num_tx_ant, inner, (double2*)out);
  return launch_status();
}

// OFDMChannel.call (channel/ofdm_channel.py:109-115) = cir_to_ofdm_channel + ApplyOFDMChannel (+ AWGN) in ONE launch for the case
// that h_freq itself is not needed: a [B,rx,ra,1,ta,P,T], tau [B,rx,1,P], x [B,1,ta,T,F] -> y [B,rx,ra,T,F].  no: DEVICE float[1]
// or null (no noise); (seed, call): the Philox stream samd_awgn_c64 would be given.  Bit-identical to the three separate entries.
// SAMD_ERR_UNSUPPORTED (several transmitters, shapes outside the staged-register kernel): the caller runs the separate entries.
extern "C" int samd_ofdm_channel_fused_c64(const float* a, const float* tau, const float* frequencies, const float* x, const float* no,
                                           uint64_t seed, uint64_t call, int batch, int num_rx, int num_rx_ant, int num_tx,
                                           int num_tx_ant, int num_paths, int num_time_steps, int num_freqs, int normalize, float* y,
                                           void* stream) {
  SAMD_REQUIRE(a && tau && frequencies && x && y && batch > 0, "bad argument");
  SAMD_REQUIRE(num_paths >= 1 && num_freqs >= 1 && num_time_steps >= 1, "bad size");
  static samd::CachedOpt opt_off("SAMD_NO_FUSED_CHANNEL");
  if (num_tx != 1 || opt_off.is_set()) { set_error("fused OFDM channel: one transmitter only"); return SAMD_ERR_UNSUPPORTED; }
  if (no && !C2O_FUSED_NOISE) { set_error("fused OFDM channel: built without in-kernel noise (add it with samd_awgn_c64)"); return SAMD_ERR_UNSUPPORTED; }
  const int rows = num_rx_ant * num_tx_ant * num_time_steps;
  int best_nt = 0, best_rpt = 0;
  double best_u = 0.0;
  for (int nt = 256; nt <= 512 && num_freqs <= 512; nt += 64) {
    if (num_freqs > nt) continue;
    const int gq = nt / num_freqs, rpt = (rows + gq - 1) / gq;
    const double u = (double)(gq * num_freqs) / nt;
    if (rpt <= 40 && u > best_u + 1e-9) { best_u = u; best_nt = nt; best_rpt = rpt; }
  }
  const int mp = num_paths <= 8 ? 8 : num_paths <= 16 ? 16 : num_paths <= 24 ? 24 : num_paths <= 32 ? 32 : 0;
  if (!best_nt || !mp) { set_error("fused OFDM channel: shape outside the staged-register kernel"); return SAMD_ERR_UNSUPPORTED; }
  const size_t lds_p = ((size_t)mp * num_freqs + (size_t)(((best_rpt + 7) / 8) * 8) * (best_nt / num_freqs) * mp) * sizeof(float2) + 64;
  const size_t stage_b = (size_t)num_tx_ant * num_time_steps * num_freqs * sizeof(float2);   // x of one batch item, staged in LDS
  const size_t lds_f = std::max(lds_p, stage_b + 64);          // (few paths: x needs more than the tables)
  const int rpt_pad = ((best_rpt + 7) / 8) * 8;
  if (lds_f > 64 * 1024 || (size_t)num_paths * num_time_steps * num_rx_ant * num_tx_ant >= 8192 || rpt_pad % num_tx_ant != 0 ||
      (size_t)(rpt_pad / num_tx_ant) * (best_nt / num_freqs) < (size_t)num_rx_ant * num_time_steps) {
    set_error("fused OFDM channel: shape outside the staged-register kernel");
    return SAMD_ERR_UNSUPPORTED;
  }
  typedef void (*kern_t)(const float2*, const float*, const float*, int, int, int, int, int, int, int, int, float2*, C2oFuse);
#define SAMD_C2F_K(MP) {cir_to_ofdm_fused_kernel<MP, 8, 4, true>, cir_to_ofdm_fused_kernel<MP, 16, 4, true>, \
                        cir_to_ofdm_fused_kernel<MP, 24, 4, true>, cir_to_ofdm_fused_kernel<MP, 32, 4, true>, \
                        cir_to_ofdm_fused_kernel<MP, 40, 4, true>}
  static const kern_t fk[4][5] = {SAMD_C2F_K(8), SAMD_C2F_K(16), SAMD_C2F_K(24), SAMD_C2F_K(32)};
#undef SAMD_C2F_K
  const C2oFuse fu{(const float2*)x, no, seed, call, (float2*)y};
  hipLaunchKernelGGL(fk[mp / 8 - 1][(best_rpt + 7) / 8 - 1], dim3(batch * num_rx), dim3(best_nt), lds_f, (hipStream_t)stream,
                     (const float2*)a, tau, frequencies, num_rx, num_rx_ant, num_tx, num_tx_ant, num_paths, num_time_steps, num_freqs,
                     normalize, (float2*)nullptr, fu);
  return launch_status();
}

extern "C" int samd_cir_to_ofdm_c64(const float* a, const float* tau, const float* frequencies, int batch, int num_rx,
                                    int num_rx_ant, int num_tx, int num_tx_ant, int num_paths, int num_time_steps,
                                    int num_freqs, int normalize, float* h_freq, void* stream) {
  SAMD_REQUIRE(a && tau && frequencies && h_freq && batch > 0, "bad argument");
  SAMD_REQUIRE(num_paths >= 1 && num_freqs >= 1 && num_time_steps >= 1, "bad size");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(batch * num_rx * num_tx);
  typedef void (*kern_t)(const float2*, const float*, const float*, int, int, int, int, int, int, int, int, float2*);
  {
    // results staged in registers (cir_to_ofdm_reg_kernel): block size = the multiple of 64 that fills its lanes best with
    // G = blockDim / F row groups; every thread's ceil(rows / G) rows must fit the unrolled register array
    static samd::CachedOpt opt_old("SAMD_C2O_TWO_PASS");       // development: the two-pass kernel of rounds 1-4
    const int rows = num_rx_ant * num_tx_ant * num_time_steps;
    int best_nt = 0, best_rpt = 0;
    double best_u = 0.0;
    for (int nt = 256; nt <= 512 && num_freqs <= 512; nt += 64) {
      if (num_freqs > nt) continue;
      const int gq = nt / num_freqs, rpt = (rows + gq - 1) / gq;
      const double u = (double)(gq * num_freqs) / nt;
      if (rpt <= 40 && u > best_u + 1e-9) { best_u = u; best_nt = nt; best_rpt = rpt; }
    }
    const int mp = num_paths <= 8 ? 8 : num_paths <= 16 ? 16 : num_paths <= 24 ? 24 : num_paths <= 32 ? 32 : 0;
    const size_t lds_r = ((size_t)mp * num_freqs + (size_t)num_rx_ant * num_tx_ant * mp * num_time_steps) * sizeof(float2) + 512 * sizeof(float);
    if (best_nt && mp && lds_r <= 64 * 1024 && !opt_old.is_set()) {
#define SAMD_C2R_K(MP) {cir_to_ofdm_reg_kernel<MP, 8>, cir_to_ofdm_reg_kernel<MP, 16>, cir_to_ofdm_reg_kernel<MP, 24>, \
                        cir_to_ofdm_reg_kernel<MP, 32>, cir_to_ofdm_reg_kernel<MP, 40>}
      static const kern_t rk[4][5] = {SAMD_C2R_K(8), SAMD_C2R_K(16), SAMD_C2R_K(24), SAMD_C2R_K(32)};
#undef SAMD_C2R_K
      // paths in passes of 4 with the next row's taps requested a row ahead (cir_to_ofdm_pass_kernel: 94 registers, 5 waves per
      // SIMD); development: SAMD_C2O_PASS = 0 keeps all phases in registers (the kernel of profiles/r05k, 164 registers), 2 =
      // passes of 2 (80 registers), 8 = passes of 8 without the look-ahead (94 registers) - profiles/r05o
      static samd::CachedOpt opt_pass("SAMD_C2O_PASS");
      const long pw = opt_pass.get(4);
#define SAMD_C2P_K(MP, PW, PF) {cir_to_ofdm_pass_kernel<MP, 8, PW, PF>, cir_to_ofdm_pass_kernel<MP, 16, PW, PF>, cir_to_ofdm_pass_kernel<MP, 24, PW, PF>, \
                                cir_to_ofdm_pass_kernel<MP, 32, PW, PF>, cir_to_ofdm_pass_kernel<MP, 40, PW, PF>}
      typedef void (*pkern_t)(const float2*, const float*, const float*, int, int, int, int, int, int, int, int, float2*, C2oFuse);
      static const pkern_t pk4[4][5] = {SAMD_C2P_K(8, 4, true), SAMD_C2P_K(16, 4, true), SAMD_C2P_K(24, 4, true), SAMD_C2P_K(32, 4, true)};
      static const pkern_t pk2[4][5] = {SAMD_C2P_K(8, 2, true), SAMD_C2P_K(16, 2, true), SAMD_C2P_K(24, 2, true), SAMD_C2P_K(32, 2, true)};
      static const pkern_t pk8n[4][5] = {SAMD_C2P_K(8, 8, false), SAMD_C2P_K(16, 8, false), SAMD_C2P_K(24, 8, false), SAMD_C2P_K(32, 8, false)};
#undef SAMD_C2P_K
      const size_t lds_p = ((size_t)mp * num_freqs + (size_t)(((best_rpt + 7) / 8) * 8) * (best_nt / num_freqs) * mp) * sizeof(float2) + 64;
      const bool pass = (pw == 8 || pw == 4 || pw == 2) && lds_p <= 64 * 1024 && (size_t)num_paths * num_time_steps * num_rx_ant * num_tx_ant < 8192;
      if (pass)
        hipLaunchKernelGGL((pw == 8 ? pk8n : pw == 4 ? pk4 : pk2)[mp / 8 - 1][(best_rpt + 7) / 8 - 1], grid, dim3(best_nt), lds_p, st,
                           (const float2*)a, tau, frequencies, num_rx, num_rx_ant, num_tx, num_tx_ant, num_paths, num_time_steps, num_freqs,
                           normalize, (float2*)h_freq, C2oFuse{});
      else
        hipLaunchKernelGGL(rk[mp / 8 - 1][(best_rpt + 7) / 8 - 1], grid, dim3(best_nt), lds_r, st, (const float2*)a, tau, frequencies, num_rx,
                           num_rx_ant, num_tx, num_tx_ant, num_paths, num_time_steps, num_freqs, normalize, (float2*)h_freq);
      return launch_status();
    }
  }
  const size_t tab_b = (size_t)num_paths * num_freqs * sizeof(float2) + 256 * sizeof(float);
  const size_t taps_b = (size_t)num_rx_ant * num_tx_ant * num_paths * num_time_steps * sizeof(float2);
  SAMD_REQUIRE(tab_b <= 160 * 1024, "phase table (num_paths x num_freqs) exceeds the LDS");
  const bool taps_lds = tab_b + taps_b <= 160 * 1024;      // else the taps are read from global memory (L1 broadcast)
  const size_t lds = tab_b + (taps_lds ? taps_b : 0);
  const dim3 blk(256);
#define SAMD_C2O_K(MAXP) {cir_to_ofdm_kernel<MAXP, false>, cir_to_ofdm_kernel<MAXP, true>}
  static const kern_t kerns[6][2] = {SAMD_C2O_K(8), SAMD_C2O_K(16), SAMD_C2O_K(24), SAMD_C2O_K(32), SAMD_C2O_K(64), SAMD_C2O_K(0)};
#undef SAMD_C2O_K
  const int ki = num_paths <= 8 ? 0 : num_paths <= 16 ? 1 : num_paths <= 24 ? 2 : num_paths <= 32 ? 3 : num_paths <= 64 ? 4 : 5;
  const kern_t kern = kerns[ki][taps_lds ? 1 : 0];
  if (lds > 64 * 1024) SAMD_SET_MAX_LDS(kern, 160 * 1024);     // once per (kernel, device): common.h
  hipLaunchKernelGGL(kern, grid, blk, lds, st, (const float2*)a, tau, frequencies, num_rx, num_rx_ant, num_tx, num_tx_ant,
                     num_paths, num_time_steps, num_freqs, normalize, (float2*)h_freq);
  return launch_status();
}

extern "C" int samd_apply_ofdm_channel_c64(const float* x, const float* h_freq, int batch, int num_rx_x_ant,
                                           int num_tx_x_ant, int num_re, float* y, void* stream) {
  SAMD_REQUIRE(x && h_freq && y, "null argument");
  const int64_t total = (int64_t)batch * num_rx_x_ant * num_re;
  if (total == 0) return SAMD_OK;
  hipLaunchKernelGGL(apply_ofdm_channel_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)x, (const float2*)h_freq, total, num_rx_x_ant, num_tx_x_ant, num_re, (float2*)y);
  return launch_status();
}

// ---- LinearInterpolator (channel_estimation.py:437-733): frequency then time interpolation of the
// estimates at the pilots.  hp [rows, S, P]; fi0/fi1 [S,T,F] = 1 + pilot number of the left/right
// support (0: the zero pad of symbols without pilots), fx0/fx1 their subcarrier positions;
// t0/t1 [S,T] the supporting OFDM symbols.  divide_no_nan: a zero span gives slope 0.
template <typename R2, typename R>
__device__ __forceinline__ R2 lin_freq(const R2* __restrict__ hp_rs, const int32_t* __restrict__ fi0,
                                           const int32_t* __restrict__ fi1, const float* __restrict__ fx0,
                                           const float* __restrict__ fx1, int idx, int f) {
  const int i0 = fi0[idx], i1 = fi1[idx];
  const R2 y0 = i0 > 0 ? hp_rs[i0 - 1] : R2{(R)0, (R)0};
  const R2 y1 = i1 > 0 ? hp_rs[i1 - 1] : R2{(R)0, (R)0};
  const R x0 = (R)fx0[idx], dx = (R)fx1[idx] - x0;
  R2 slope = R2{(R)0, (R)0};
  if (dx != (R)0) slope = R2{(y1.x - y0.x) / dx, (y1.y - y0.y) / dx};
  const R w = (R)f - x0;
  return R2{w * slope.x + y0.x, w * slope.y + y0.y};
}

template <typename R2, typename R>
__global__ void lin_interp_kernel(const R2* __restrict__ hp, const int32_t* __restrict__ fi0,
                                  const int32_t* __restrict__ fi1, const float* __restrict__ fx0,
                                  const float* __restrict__ fx1, const int32_t* __restrict__ t0,
                                  const int32_t* __restrict__ t1, const float* __restrict__ npil, long long total, int S,
                                  int P, int T, int F, int time_avg, R2* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
  const int f = (int)(i % F);
  const int t = (int)((i / F) % T);
  const int s = (int)((i / ((long long)F * T)) % S);
  const long long r = i / ((long long)F * T * S);
  const R2* hp_rs = hp + (r * S + s) * P;
  const int base = s * T * F;
  R2 res;
  if (time_avg) {
    R2 acc = R2{(R)0, (R)0};
    for (int tt = 0; tt < T; ++tt) {
      const R2 v = lin_freq<R2, R>(hp_rs, fi0, fi1, fx0, fx1, base + tt * F + f, f);
      acc.x += v.x; acc.y += v.y;
    }
    res = R2{acc.x / (R)npil[s], acc.y / (R)npil[s]};
  } else {
    const int a = t0[s * T + t], b = t1[s * T + t];
    const R2 y0 = lin_freq<R2, R>(hp_rs, fi0, fi1, fx0, fx1, base + a * F + f, f);
    const R2 y1 = lin_freq<R2, R>(hp_rs, fi0, fi1, fx0, fx1, base + b * F + f, f);
    const R dx = (R)(b - a);
    R2 slope = R2{(R)0, (R)0};
    if (dx != (R)0) slope = R2{(y1.x - y0.x) / dx, (y1.y - y0.y) / dx};
    const R w = (R)(t - a);
    res = R2{w * slope.x + y0.x, w * slope.y + y0.y};
  }
  out[i] = res;
  }
}

extern "C" int samd_lin_interp_c64(const float* hp, const int32_t* fi0, const int32_t* fi1, const float* fx0,
                                   const float* fx1, const int32_t* t0, const int32_t* t1, const float* npil, int rows,
                                   int num_streams, int num_pilots, int num_ofdm_symbols, int num_subcarriers,
                                   int time_avg, float* out, void* stream) {
  SAMD_REQUIRE(hp && fi0 && fi1 && fx0 && fx1 && t0 && t1 && npil && out, "null argument");
  const long long total = (long long)rows * num_streams * num_ofdm_symbols * num_subcarriers;
  if (total == 0) return SAMD_OK;
  hipLaunchKernelGGL((lin_interp_kernel<float2, float>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, (const float2*)hp,
                     fi0, fi1, fx0, fx1, t0, t1, npil, total, num_streams, num_pilots, num_ofdm_symbols, num_subcarriers,
                     time_avg, (float2*)out);
  return launch_status();
}

// precision = "double" (reference block.py:25-52): complex128 estimates, the same index / position tables (positions are small
// integers, exact in float32)
extern "C" int samd_lin_interp_c128(const double* hp, const int32_t* fi0, const int32_t* fi1, const float* fx0, const float* fx1,
                                    const int32_t* t0, const int32_t* t1, const float* npil, int rows, int num_streams, int num_pilots,
                                    int num_ofdm_symbols, int num_subcarriers, int time_avg, double* out, void* stream) {
  SAMD_REQUIRE(hp && fi0 && fi1 && fx0 && fx1 && t0 && t1 && npil && out, "null argument");
  const long long total = (long long)rows * num_streams * num_ofdm_symbols * num_subcarriers;
  if (total == 0) return SAMD_OK;
  hipLaunchKernelGGL((lin_interp_kernel<double2, double>), dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const double2*)hp, fi0, fi1, fx0, fx1, t0, t1, npil, total, num_streams, num_pilots, num_ofdm_symbols,
                     num_subcarriers, time_avg, (double2*)out);
  return launch_status();
}

extern "C" int samd_ls_gather_scale_c64(const float* y, const int32_t* src, const float* coef, int rows, int num_streams,
                                        int n_out, int n_in, float* out, void* stream) {
  SAMD_REQUIRE(y && src && coef && out, "null argument");
  const int64_t total = (int64_t)rows * num_streams * n_out;
  if (total == 0) return SAMD_OK;
  hipLaunchKernelGGL(ls_gather_scale_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)y, src, (const float2*)coef, total, num_streams, n_out, n_in, (float2*)out);
  return launch_status();
}
