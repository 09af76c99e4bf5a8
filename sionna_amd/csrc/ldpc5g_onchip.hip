// On-chip flooding min-sum decoder for 5G-NR LDPC codes, statically scheduled and fully
// unrolled ("v2" of the engine in ldpc5g.hip; same algorithm, same arithmetic, same results).
//
// Replaces LDPC5GDecoder.call = rate recovery + LDPCBPDecoder._bp_iter x num_iter +
// cn_update_(offset_)minsum + vn_update_sum + output mapping
// (reference src/sionna/phy/fec/ldpc/decoding.py:1427-1536, 416-524, 681-953).
//
// Why a second engine: the first one walks a base-graph row with a rolled loop - one scalar
// table load, one LDS read and one dependent min-update per trip - so every wave sits in
// LDS/scalar latency (measured 97 ms per 65536 C2 decodes).  Here
//   * every base row / column is processed by a function instantiated for its exact degree
//     (rows: 3..10 and 19 - all degrees of BG1/BG2; columns: padded to a few classes with a
//     zero "dummy check node"), so the table entries arrive as wide scalar loads, all LDS
//     reads of a node are issued back to back and the arithmetic is straight-line;
//   * the (row, 64-lane chunk) and (column, chunk) work items are assigned to the 16 waves
//     of the workgroup on the host (longest-processing-time first) - no atomics, no
//     divergence: all lanes of a wave share (c, shift) in SGPRs and differ only in z;
//   * the two smallest magnitudes are tracked with v_min/v_med3, the sign bits are spliced
//     in with v_bfi: ~15 VALU per edge in the CN phase, ~11 in the VN phase.
// LDS per codeword: xt[nbu*Z] + llr[nbu*Z] floats, check-node state float2 m12[(ncu+1)*Z]
// + uint pk[(ncu+1)*Z] (the extra block is the dummy CN): 141.8 KB for C2 -> 1 workgroup
// (16 waves) per CU.
#include "ldpc5g.h"

#include <algorithm>
#include <numeric>

namespace samd {

__device__ __forceinline__ unsigned f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(unsigned x) { return __uint_as_float(x); }
__device__ __forceinline__ unsigned bfi(unsigned mask, unsigned a, unsigned b) { return (a & mask) | (b & ~mask); }
__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// Instruction selection follows the measured gfx950 issue rates (tools/ubench/valu_rate.hip):
// v_add/v_sub/v_and/v_fma and v_cmp+v_cndmask run at the full rate, v_min/v_max/v_med3, shifts,
// v_bfi and v_alignbit at half rate.  All LDS addressing is done in BYTES (no scaling shifts),
// the modulo of the cyclic shift is an AND when Z is a power of two, the per-edge sign bits are
// kept MSB-first so that "next edge" is one add (w += w) and collecting them is one
// v_alignbit per edge.
//
// Check-node state: m12[cn] = (M1, M2) magnitudes after offset and clip; pk[cn] = position of
// the unique minimum (bits 0-4) | sign of the c2v of edge i at bit 31-i.

// (zz4 + s4) mod 4Z  /  (zz4 - s4) mod 4Z on byte offsets; zw = 4Z-1 (POW2) or 4Z
template <bool POW2>
__device__ __forceinline__ unsigned wrap_add(unsigned zz4, unsigned s4, unsigned zw) {
  const unsigned t = zz4 + s4;
  return POW2 ? (t & zw) : min(t, t - zw);
}
template <bool POW2>
__device__ __forceinline__ unsigned wrap_sub(unsigned zz4, unsigned s4, unsigned zw) {
  const unsigned t = zz4 - s4;
  return POW2 ? (t & zw) : min(t, t + zw);
}

// ---- one check node per lane: row of exact degree D.  ent[i] = (c*Z*4) | (shift*4 << 18)
// FUSE1: the row's last edge goes to a degree-1 variable node of the same lane (the identity
// block of the base graph's extension part, shift 0).  Its total x_tot = c2v + llr is formed here
// from the row's own state instead of by a VN-phase item - same arithmetic ((0 + c2v) + llr), no
// LDS round trip: 42 of BG1's 68 columns (13 % of the edges, 62 % of the VN work items) vanish
// from the VN phase.
// NCH: number of consecutive 64-lane chunks of lifted copies handled by the wave in one pass
// (lane z and lane z+64 share every scalar): two independent dependency chains per wave hide
// LDS latency at 4 waves / SIMD and halve the per-item scalar work.
template <int D, int NCH, bool POW2, bool FUSE1>
__device__ __forceinline__ void cn_row(const int32_t* __restrict__ ent, unsigned zz4, unsigned zw, unsigned cn4,
                                       const char* __restrict__ xt_b, const char* __restrict__ llr_b,
                                       char* __restrict__ m12_b, char* __restrict__ pk_b, float llr_max,
                                       float offset) {
  int e[D];
#pragma unroll
  for (int i = 0; i < D; ++i) e[i] = ent[i];
  float x[NCH][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      if (FUSE1 && i == D - 1)                              // channel LLR of the fused degree-1 VN
        x[h][i] = *reinterpret_cast<const float*>(llr_b + (unsigned)(e[i] & 0x3FFFF) + zz4 + 256u * h);
      else
        x[h][i] = *reinterpret_cast<const float*>(xt_b + (unsigned)(e[i] & 0x3FFFF) +
                                                  wrap_add<POW2>(zz4 + 256u * h, (unsigned)e[i] >> 18, zw));
    }
  float2 om[NCH];
  unsigned w[NCH], oidx[NCH], idx[NCH], neg[NCH];
  float m1s[NCH], min2[NCH];                                // m1s: the minimum WITH its sign (|.| is free)
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    om[h] = *reinterpret_cast<const float2*>(m12_b + 2 * (cn4 + 256u * h));
    w[h] = *reinterpret_cast<const unsigned*>(pk_b + cn4 + 256u * h);
    oidx[h] = w[h] & 31u;
    m1s[h] = INFINITY; min2[h] = INFINITY; idx[h] = 0; neg[h] = 0;
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      // previous c2v of this edge from the compressed state, then v2c = clip(x_tot - c2v)
      const float mag = (oidx[h] == (unsigned)i) ? om[h].y : om[h].x;
      const float c2v = u2f((w[h] & 0x80000000u) | f2u(mag));     // mag >= 0: one v_and_or
      asm("v_add_u32 %0, %1, %1" : "=v"(w[h]) : "v"(w[h]));       // w += w at the full VALU rate (not a shift)
      const float xi = (FUSE1 && i == D - 1) ? c2v + x[h][i] : x[h][i];   // x_tot of the fused degree-1 VN
      const float v2c = med3(xi - c2v, -llr_max, llr_max);
      neg[h] = __builtin_amdgcn_alignbit(neg[h], f2u(v2c), 31);   // (neg << 1) | sign(v2c); v2c is never -0
      const bool lt = fabsf(v2c) < fabsf(m1s[h]);
      idx[h] = lt ? (unsigned)i : idx[h];
      min2[h] = med3(fabsf(m1s[h]), min2[h], fabsf(v2c));         // second smallest, with multiplicity
      m1s[h] = lt ? v2c : m1s[h];
    }
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    const float min1 = fabsf(m1s[h]);
    // unique minimum <=> min2 > min1; (min2 - min1) + min1 is the reference's arithmetic (:863)
    const float min_e = (min2[h] > min1) ? ((min2[h] - min1) + min1) : min1;
    float a1 = min1, a2 = min_e;
    a1 -= offset; a2 -= offset;                                   // plain min-sum: offset = 0 (exact)
    a1 = med3(a1, 0.f, llr_max);
    a2 = med3(a2, 0.f, llr_max);
    // neg holds sign(v2c_i) at bit D-1-i; own sign x node sign, then MSB-first
    const unsigned all = (1u << D) - 1u;
    const unsigned sg = (__popc(neg[h]) & 1) ? (neg[h] ^ all) : neg[h];
    *reinterpret_cast<float2*>(m12_b + 2 * (cn4 + 256u * h)) = make_float2(a1, a2);
    *reinterpret_cast<unsigned*>(pk_b + cn4 + 256u * h) = idx[h] | (sg << (32 - D));
  }
}

// ---- partial sums over D edge slots of NCH variable nodes per lane (real edges, then dummies).
// ent[2i] = (r*Z*4) | (shift*4 << 18), ent[2i+1] = position of the edge inside its row
template <int D, int NCH, bool POW2>
__device__ __forceinline__ void vn_part(const int32_t* __restrict__ ent, unsigned zz4, unsigned zw,
                                        const char* __restrict__ m12_b, const char* __restrict__ pk_b,
                                        float (&x)[NCH]) {
  int e0[D], e1[D];
#pragma unroll
  for (int i = 0; i < D; ++i) { e0[i] = ent[2 * i]; e1[i] = ent[2 * i + 1]; }
  float2 m[NCH][D];
  unsigned q[NCH][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const unsigned a4 = (unsigned)(e0[i] & 0x3FFFF) + wrap_sub<POW2>(zz4 + 256u * h, (unsigned)e0[i] >> 18, zw);
      m[h][i] = *reinterpret_cast<const float2*>(m12_b + 2 * a4);
      q[h][i] = *reinterpret_cast<const unsigned*>(pk_b + a4);
    }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const unsigned pos = (unsigned)e1[i];
      const float mag = ((q[h][i] & 31u) == pos) ? m[h][i].y : m[h][i].x;
      x[h] += u2f(((q[h][i] << pos) & 0x80000000u) | f2u(mag));    // ascending CN = edge order
    }
}

// one VN work item: column c, NCH chunks starting at lane offset zz
template <int NCH, bool POW2>
__device__ __forceinline__ void vn_item(const int32_t* __restrict__ ent, int nfull, int rem, unsigned zz4,
                                        unsigned zw, unsigned vn4, float* __restrict__ xt,
                                        const float* __restrict__ llr, const char* __restrict__ m12_b,
                                        const char* __restrict__ pk_b) {
  float x[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) x[h] = 0.f;
  for (int f = 0; f < nfull; ++f) {
    vn_part<16, NCH, POW2>(ent, zz4, zw, m12_b, pk_b, x);
    ent += 32;
  }
#define SAMD_VN(D) case D: vn_part<D, NCH, POW2>(ent, zz4, zw, m12_b, pk_b, x); break
  switch (rem) {
    SAMD_VN(1); SAMD_VN(2); SAMD_VN(3); SAMD_VN(4); SAMD_VN(5); SAMD_VN(6); SAMD_VN(7); SAMD_VN(8);
    SAMD_VN(10); SAMD_VN(12); SAMD_VN(14);
    default: break;
  }
#undef SAMD_VN
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    const unsigned v = (vn4 >> 2) + 64u * h;
    xt[v] = x[h] + llr[v];                                  // unclipped x_tot (decoding.py:716)
  }
}

template <int NCH, bool POW2>
__device__ __forceinline__ void cn_item(int desc, const int32_t* __restrict__ ent, unsigned zz4, unsigned zw,
                                        unsigned cn4, const char* xt_b, const char* llr_b, char* m12_b, char* pk_b,
                                        float llr_max, float offset) {
#define SAMD_CN(D, F) case D: cn_row<D, NCH, POW2, F>(ent, zz4, zw, cn4, xt_b, llr_b, m12_b, pk_b, llr_max, offset); break
  if ((desc >> 24) & 1) {                                   // last edge fused with its degree-1 VN
    switch ((desc >> 16) & 0xFF) {
      SAMD_CN(3, true); SAMD_CN(4, true); SAMD_CN(5, true); SAMD_CN(6, true); SAMD_CN(7, true);
      SAMD_CN(8, true); SAMD_CN(9, true); SAMD_CN(10, true);
      default: break;
    }
  } else {
    switch ((desc >> 16) & 0xFF) {
      SAMD_CN(3, false); SAMD_CN(4, false); SAMD_CN(5, false); SAMD_CN(6, false); SAMD_CN(7, false);
      SAMD_CN(8, false); SAMD_CN(9, false); SAMD_CN(10, false); SAMD_CN(19, false);
      default: break;
    }
  }
#undef SAMD_CN
}

// NW = waves per workgroup.  One workgroup owns one codeword; codes whose state needs <= 80 / <= 40 KB of
// LDS run as 2 x 8 / 4 x 4 waves per CU: the per-wave item lists get longer (better balance) and the two
// barriers per iteration only synchronise the waves of one codeword.
// LLRG: the clipped channel LLRs live in the caller's workspace (one nbu*Z row per workgroup, L2 resident:
// read once per VN and iteration) instead of LDS - codes up to (nbu + 3 (ncu+1)) Z 4 <= 160 KB fit.
// XTG (with LLRG): the variable-node totals x_tot live in the workspace too; LDS then only holds the
// compressed check-node state, 12 (ncu+1) Z bytes - every 5G code with rate >= ~0.45 at Z = 384 fits.
// PKG (with XTG): the packed sign / min-position words move out as well; LDS = (M1, M2) only, 8 (ncu+1) Z
// bytes <= 144 KB for every 5G code.
template <bool POW2, int NW, bool LLRG = false, bool XTG = false, bool PKG = false>
__global__ __launch_bounds__(NW * 64) void ldpc5g_decode_v2_kernel(
    const float* __restrict__ llr_in, float* __restrict__ out, float* __restrict__ llr_ws, RateMatch p, int n_cn, int ncu, int nbu, int batch,
    int num_iter, float llr_max, float offset, int hard_out, int return_infobits,
    const int32_t* __restrict__ row_pad, const int32_t* __restrict__ row_deg, const int32_t* __restrict__ col_pad,
    const int32_t* __restrict__ col_cls, const int32_t* __restrict__ cn_sched_ptr,
    const int32_t* __restrict__ cn_sched, const int32_t* __restrict__ vn_sched_ptr,
    const int32_t* __restrict__ vn_sched) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = NW * 64;
  const unsigned z = (unsigned)p.z;
  const unsigned zw = POW2 ? 4u * z - 1u : 4u * z;
  const int n_vn = p.n_vn;
  const int nx = nbu * (int)z, ns = (ncu + 1) * (int)z;
  // workspace row of a workgroup: [llr nx | xt nx | pk ns] (only the parts that are global)
  float* wsrow = llr_ws + (size_t)blockIdx.x * ((size_t)(XTG ? 2 : 1) * nx + (PKG ? ns : 0));
  float* xt = XTG ? wsrow + nx : smem;
  float* llr = LLRG ? wsrow : xt + nx;
  float2* m12 = reinterpret_cast<float2*>(XTG ? smem : (LLRG ? smem + nx : smem + 2 * nx));
  unsigned* pk = PKG ? reinterpret_cast<unsigned*>(wsrow + 2 * nx) : reinterpret_cast<unsigned*>(m12 + ns);
  const char* xt_b = reinterpret_cast<const char*>(xt);
  const char* llr_b = reinterpret_cast<const char*>(llr);
  char* m12_b = reinterpret_cast<char*>(m12);
  char* pk_b = reinterpret_cast<char*>(pk);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = cn_sched_ptr[w], c1 = cn_sched_ptr[w + 1];
  const int v0 = vn_sched_ptr[w], v1 = vn_sched_ptr[w + 1];

  for (int b = blockIdx.x; b < batch; b += gridDim.x) {
    const float* row = llr_in + (size_t)b * p.n;
    for (int v = tid; v < nx; v += NT) {
      // decoding.py:552-565: clip, then logits -> LLR; "+ 0.f" turns -0 into +0 (numerically the
      // same LLR) so that the sign bit of every later v2c equals (v2c < 0)
      const float l = (v < n_vn) ? (-1.f * clampf(recover_llr(p, row, v, llr_max), -llr_max, llr_max)) + 0.f : 0.f;
      llr[v] = l;
      xt[v] = l;
    }
    for (int c = tid; c < ns; c += NT) { m12[c] = make_float2(0.f, 0.f); pk[c] = 0u; }
    __syncthreads();

    for (int it = 0; it < num_iter; ++it) {
      for (int t = c0; t < c1; ++t) {
        // desc: r | chunk<<8 | degree<<16 | fused<<24 | pair<<25  (pair: chunks q and q+1, all lanes valid)
        const int desc = __builtin_amdgcn_readfirstlane(cn_sched[t]);
        const int r = desc & 0xFF;
        onchip_setprio(desc >> 28);                          // longest remaining work first (ldpc5g.h)
        const unsigned zz = (unsigned)(((desc >> 8) & 0xFF) * 64 + lane);
        const unsigned cn = (unsigned)r * z + zz;
        const int32_t* ent = row_pad + r * kRowStride;
        if ((desc >> 25) & 1) {
          cn_item<2, POW2>(desc, ent, 4u * zz, zw, 4u * cn, xt_b, llr_b, m12_b, pk_b, llr_max, offset);
        } else if (zz < z && cn < (unsigned)n_cn) {
          cn_item<1, POW2>(desc, ent, 4u * zz, zw, 4u * cn, xt_b, llr_b, m12_b, pk_b, llr_max, offset);
        }
      }
      __syncthreads();
      for (int t = v0; t < v1; ++t) {                        // columns of degree >= 2 (and unfused ones)
        // desc: c | chunk<<8 | nfull<<16 | rem<<20 | pair<<25
        const int desc = __builtin_amdgcn_readfirstlane(vn_sched[t]);
        const int c = desc & 0xFF;
        onchip_setprio(desc >> 28);
        const unsigned zz = (unsigned)(((desc >> 8) & 0xFF) * 64 + lane);
        const unsigned vn = (unsigned)c * z + zz;
        const int32_t* ent = col_pad + c * (2 * kColStride);
        const int nfull = (desc >> 16) & 0xF, rem = (desc >> 20) & 0x1F;
        if ((desc >> 25) & 1) {
          vn_item<2, POW2>(ent, nfull, rem, 4u * zz, zw, 4u * vn, xt, llr, m12_b, pk_b);
        } else if (zz < z && vn < (unsigned)n_vn) {
          vn_item<1, POW2>(ent, nfull, rem, 4u * zz, zw, 4u * vn, xt, llr, m12_b, pk_b);
        }
      }
      __syncthreads();
    }
    if (!return_infobits) {
      // totals of the fused degree-1 VNs are only needed for the codeword output: one class-1 pass
      for (int t = vn_sched_ptr[NW + 1 + w]; t < vn_sched_ptr[NW + 2 + w]; ++t) {
        const int desc = __builtin_amdgcn_readfirstlane(vn_sched[t]);
        const int c = desc & 0xFF;
        const unsigned zz = (unsigned)(((desc >> 8) & 0xFF) * 64 + lane);
        const int vn = c * (int)z + (int)zz;
        if (zz < z && vn < n_vn) {
          float x1[1] = {0.f};
          vn_part<1, 1, POW2>(col_pad + c * (2 * kColStride), 4u * zz, zw, m12_b, pk_b, x1);
          xt[vn] = x1[0] + llr[vn];
        }
      }
      __syncthreads();
    }
    // ---------------- output (decoding.py:620-626, 1486-1531)
    if (return_infobits) {
      float* o = out + (size_t)b * p.k;
      for (int v = tid; v < p.k; v += NT) {
        const float x = clampf(xt[v], -llr_max, llr_max);
        o[v] = hard_out ? ((0.f >= x) ? 1.f : 0.f) : -1.f * x;
      }
    } else {
      float* o = out + (size_t)b * p.n;
      for (int i = tid; i < p.n; i += NT) {
        const float x = clampf(xt[short_to_full(p, out_to_short(p, i))], -llr_max, llr_max);
        o[i] = hard_out ? ((0.f >= x) ? 1.f : 0.f) : -1.f * x;
      }
    }
    __syncthreads();
  }
}

static const int kCnDegrees[] = {3, 4, 5, 6, 7, 8, 9, 10, 19};
static const int kVnRemClasses[] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14};   // remainder after full chunks of 16

int build_onchip_tables(samd_ldpc5g* h, const std::vector<std::vector<std::pair<int, int>>>& by_row) {
  const int z = h->z;
  h->ncu = (h->n_cn + z - 1) / z;
  h->nbu = (h->n_vn + z - 1) / z;
  // packed fields: byte offsets < 2^18, byte shifts < 2^11, row/col index < 256
  bool ok = h->mb <= 255 && h->nb <= 255 && (h->ncu + 1) * z * 4 < (1 << 18) && h->nbu * z * 4 < (1 << 18);
  std::vector<int32_t> row_pad((size_t)h->mb * kRowStride, 0), row_deg(h->mb, 0);
  std::vector<std::vector<std::pair<int32_t, int32_t>>> cols(h->nb);
  for (int r = 0; r < h->mb && ok; ++r) {
    const int d = (int)by_row[r].size();
    row_deg[r] = d;
    if (r < h->ncu && std::find(std::begin(kCnDegrees), std::end(kCnDegrees), d) == std::end(kCnDegrees)) ok = false;
    if (d > kRowStride) { ok = false; break; }
    for (int i = 0; i < d; ++i) {
      const int c = by_row[r][i].first, s = by_row[r][i].second;
      row_pad[(size_t)r * kRowStride + i] = (c * z * 4) | ((s * 4) << 18);
      if (r < h->ncu) cols[c].push_back({(r * z * 4) | ((s * 4) << 18), i});   // rows ascending
    }
  }
  // column tables: 2 dwords per slot, full chunks of 16 slots then a remainder class
  std::vector<int32_t> col_pad((size_t)h->nb * 2 * kColStride, 0), col_cls(h->nb, 0);
  const int32_t dummy = h->ncu * z * 4;                       // zero "dummy check node" block, shift 0
  for (int c = 0; c < h->nb && ok; ++c) {
    const int d = (int)cols[c].size();
    int nfull = d / 16;
    const int r0 = d % 16;
    int rem = -1;
    for (int k : kVnRemClasses) if (k >= r0) { rem = k; break; }
    if (rem < 0) { ++nfull; rem = 0; }                          // remainder 15: one more full chunk, one dummy slot
    if (nfull * 16 + rem > kColStride) { ok = false; break; }
    col_cls[c] = nfull | (rem << 4);
    for (int i = 0; i < kColStride; ++i) {
      col_pad[((size_t)c * kColStride + i) * 2] = i < d ? cols[c][i].first : dummy;
      col_pad[((size_t)c * kColStride + i) * 2 + 1] = i < d ? cols[c][i].second : 0;
    }
  }
  h->v2_ok = ok ? 1 : 0;
  if (!ok) return SAMD_OK;
  const int chunks = (z + 63) / 64;
  if (chunks > 255) { h->v2_ok = 0; return SAMD_OK; }
  // rows whose last edge is the only edge of its column, with shift 0 and degree 3..10: that
  // degree-1 VN is handled inside the CN phase (cn_row<..., FUSE1>)
  std::vector<char> row_fused(h->mb, 0), col_fused(h->nb, 0);
  for (int r = 0; r < h->ncu; ++r) {
    const int d = row_deg[r];
    const int c = by_row[r][d - 1].first, s = by_row[r][d - 1].second;
    if (cols[c].size() == 1 && s == 0 && d >= 3 && d <= 10) { row_fused[r] = 1; col_fused[c] = 1; }
  }
  // item descriptors: CN  r | chunk<<8 | degree<<16 | fused<<24 ;  VN  c | chunk<<8 | nfull<<16 | rem<<20
  // two consecutive fully valid 64-lane chunks of a row / column form one "pair" item (bit 25);
  // columns with >= 20 slots stay single-chunk items so that no item dwarfs a wave's fair share
  std::vector<std::pair<int, int32_t>> ci, vi, v1i;
  for (int r = 0; r < h->ncu; ++r)
    for (int q = 0; q < chunks; ++q) {
      if (r * z + q * 64 >= h->n_cn) continue;
      const int32_t d0 = r | (q << 8) | (row_deg[r] << 16) | (row_fused[r] << 24);
      const bool pair = (q + 2) * 64 <= z && r * z + (q + 2) * 64 <= h->n_cn;
      if (pair) { ci.push_back({2 * row_deg[r], d0 | (1 << 25)}); ++q; }
      else ci.push_back({row_deg[r], d0});
    }
  for (int c = 0; c < h->nbu; ++c)
    for (int q = 0; q < chunks; ++q) {
      if (c * z + q * 64 >= h->n_vn) continue;
      const int nfull = col_cls[c] & 0xF, rem = col_cls[c] >> 4, slots = nfull * 16 + rem;
      if (col_fused[c]) { v1i.push_back({1, c | (q << 8)}); continue; }
      const int32_t d0 = c | (q << 8) | (nfull << 16) | (rem << 20);
      const bool pair = slots < 20 && (q + 2) * 64 <= z && c * z + (q + 2) * 64 <= h->n_vn;
      if (pair) { vi.push_back({2 * slots, d0 | (1 << 25)}); ++q; }
      else vi.push_back({slots, d0});
    }
  // vn_sched_ptr = [16+1 offsets of the per-iteration lists | 16+1 offsets of the final degree-1 pass]
  std::vector<int32_t> cp, cl, vp, vl, v1p, v1l;
  // waves per workgroup from the LDS footprint: as many codewords per CU as fit, 16 waves in total
  size_t lds = ((size_t)2 * h->nbu + (size_t)3 * (h->ncu + 1)) * z * 4;
  h->llr_global = 0;
  if (lds > 160 * 1024 && ((size_t)h->nbu + (size_t)3 * (h->ncu + 1)) * z * 4 <= 160 * 1024) {
    h->llr_global = 1;                                          // channel LLRs move to the workspace (L2)
    lds = ((size_t)h->nbu + (size_t)3 * (h->ncu + 1)) * z * 4;
  } else if (lds > 160 * 1024 && (size_t)3 * (h->ncu + 1) * z * 4 <= 160 * 1024 && !opt_set("SAMD_NO_XTG")) {
    h->llr_global = 2;                                          // x_tot as well: LDS = check-node state only
    lds = (size_t)3 * (h->ncu + 1) * z * 4;
  } else if (lds > 160 * 1024 && (size_t)2 * (h->ncu + 1) * z * 4 <= 160 * 1024 && !opt_set("SAMD_NO_XTG")) {
    h->llr_global = 3;                                          // ... and the sign words: LDS = (M1, M2)
    lds = (size_t)2 * (h->ncu + 1) * z * 4;
  }
  h->dec_waves = 16;
  for (int nwc : {8, 4, 2, 1})
    if (lds * (size_t)(kDecWaves / nwc) <= 160 * 1024) h->dec_waves = nwc;
  if (opt_set("SAMD_ONCHIP_WAVES")) {
    const std::string e_s = opt_str("SAMD_ONCHIP_WAVES");
    const char* e = e_s.c_str();
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) h->dec_waves = v;      // experiments: force a workgroup size
  }
  const int nw = h->dec_waves;
  lpt_schedule(ci, nw, &cp, &cl);
  lpt_schedule(vi, nw, &vp, &vl);
  {
    const std::vector<int> pc = item_priorities(ci, cp, cl), pv = item_priorities(vi, vp, vl);   // bits 28-29, see ldpc5g.h
    for (size_t j = 0; j < cl.size(); ++j) cl[j] |= pc[j] << 28;
    for (size_t j = 0; j < vl.size(); ++j) vl[j] |= pv[j] << 28;
  }
  lpt_schedule(v1i, nw, &v1p, &v1l);
  for (int32_t o : v1p) vp.push_back(o + (int32_t)vl.size());
  vl.insert(vl.end(), v1l.begin(), v1l.end());
  if (vl.empty()) vl.push_back(0);
  int rc = upload(&h->row_pad, row_pad.data(), row_pad.size());
  if (rc == SAMD_OK) rc = upload(&h->row_deg, row_deg.data(), row_deg.size());
  if (rc == SAMD_OK) rc = upload(&h->col_pad, col_pad.data(), col_pad.size());
  if (rc == SAMD_OK) rc = upload(&h->col_cls, col_cls.data(), col_cls.size());
  if (rc == SAMD_OK) rc = upload(&h->cn_sched_ptr, cp.data(), cp.size());
  if (rc == SAMD_OK) rc = upload(&h->cn_sched, cl.data(), cl.size());
  if (rc == SAMD_OK) rc = upload(&h->vn_sched_ptr, vp.data(), vp.size());
  if (rc == SAMD_OK) rc = upload(&h->vn_sched, vl.data(), vl.size());
  return rc;
}

void free_onchip_tables(samd_ldpc5g* h) {
  (void)hipFree(h->row_pad); (void)hipFree(h->row_deg); (void)hipFree(h->col_pad); (void)hipFree(h->col_cls);
  (void)hipFree(h->cn_sched_ptr); (void)hipFree(h->cn_sched); (void)hipFree(h->vn_sched_ptr); (void)hipFree(h->vn_sched);
}

// llr_global: 0 all in LDS; 1 LLRs in L2; 2 LLRs and x_tot in L2; 3 also the packed sign words (LDS = M1, M2)
static size_t onchip_lds_bytes(const samd_ldpc5g* h) {
  const int g = h->llr_global;
  return ((size_t)(g >= 2 ? 0 : 2 - g) * h->nbu + (size_t)(g == 3 ? 2 : 3) * (h->ncu + 1)) * h->z * 4;
}

static int onchip_grid(const samd_ldpc5g* h, int batch) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const size_t per_cu = std::min<size_t>((size_t)(kDecWaves / h->dec_waves), std::max<size_t>(1, (160 * 1024) / onchip_lds_bytes(h)));
  return (int)std::min<size_t>((size_t)batch, (size_t)cus * per_cu);
}

size_t onchip_workspace_bytes(const samd_ldpc5g* h, int batch) {
  if (!h->v2_ok || !h->llr_global || batch <= 0) return 0;
  const size_t row = (size_t)std::min(h->llr_global, 2) * h->nbu * h->z + (h->llr_global == 3 ? (size_t)(h->ncu + 1) * h->z : 0);
  return (size_t)onchip_grid(h, batch) * row * sizeof(float) + 256;
}

int launch_onchip_v2(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                     float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                     size_t workspace_bytes, hipStream_t st) {
  const size_t lds = onchip_lds_bytes(h);
  if (lds > 160 * 1024) {
    set_error("code does not fit in LDS");
    return SAMD_ERR_UNSUPPORTED;
  }
  float* llr_ws = nullptr;
  if (h->llr_global) {
    if (!workspace || workspace_bytes < onchip_workspace_bytes(h, batch)) {
      set_error("workspace too small (samd_ldpc5g_decode_workspace_bytes)");
      return SAMD_ERR_WORKSPACE;
    }
    llr_ws = reinterpret_cast<float*>(align_up((size_t)workspace, 256));
  }
  const bool off = (cn_mode == SAMD_CN_OFFSET_MINSUM);
  const bool pow2 = (h->z & (h->z - 1)) == 0;
  typedef void (*kern_t)(const float*, float*, float*, RateMatch, int, int, int, int, int, float, float, int, int,
                         const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*,
                         const int32_t*, const int32_t*, const int32_t*);
  static const kern_t kerns[16] = {
      ldpc5g_decode_v2_kernel<false, 16>, ldpc5g_decode_v2_kernel<true, 16>, ldpc5g_decode_v2_kernel<false, 8>,
      ldpc5g_decode_v2_kernel<true, 8>,   ldpc5g_decode_v2_kernel<false, 4>, ldpc5g_decode_v2_kernel<true, 4>,
      ldpc5g_decode_v2_kernel<false, 2>,  ldpc5g_decode_v2_kernel<true, 2>,  ldpc5g_decode_v2_kernel<false, 1>,
      ldpc5g_decode_v2_kernel<true, 1>,   ldpc5g_decode_v2_kernel<false, 16, true>, ldpc5g_decode_v2_kernel<true, 16, true>,
      ldpc5g_decode_v2_kernel<false, 16, true, true>, ldpc5g_decode_v2_kernel<true, 16, true, true>,
      ldpc5g_decode_v2_kernel<false, 16, true, true, true>, ldpc5g_decode_v2_kernel<true, 16, true, true, true>};
  const int nw = h->dec_waves;
  const int ki = h->llr_global ? 8 + 2 * h->llr_global + (pow2 ? 1 : 0)
                               : ((nw == 16 ? 0 : nw == 8 ? 2 : nw == 4 ? 4 : nw == 2 ? 6 : 8) | (pow2 ? 1 : 0));
  // set on every launch: the attribute is per device and a process may drive several
  SAMD_SET_MAX_LDS(kerns[ki], 160 * 1024);
  const int grid = onchip_grid(h, batch);
  const RateMatch rm = make_rate_match(h);
  hipLaunchKernelGGL(kerns[ki], dim3(grid), dim3(nw * 64), lds, st, llr, out, llr_ws, rm, h->n_cn, h->ncu, h->nbu, batch,
                     num_iter, llr_max, (off ? offset : 0.f), hard_out, return_infobits, h->row_pad, h->row_deg,
                     h->col_pad, h->col_cls, h->cn_sched_ptr, h->cn_sched, h->vn_sched_ptr, h->vn_sched);
  return launch_status();
}

}  // namespace samd
