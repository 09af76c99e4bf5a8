// On-chip flooding min-sum / offset-min-sum decoder with EXPLICIT messages: one float per edge in LDS.
//
// Replaces LDPC5GDecoder.call = rate recovery + LDPCBPDecoder._bp_iter x num_iter +
// cn_update_(offset_)minsum + vn_update_sum + output mapping (reference
// src/sionna/phy/fec/ldpc/decoding.py:1427-1536, 416-524, 681-953) for codes whose E 4-byte messages
// fit in 160 KB (C2: 316 Z edges = 158 KB, channel LLRs in an L2 workspace row).
//
// Why a third min-sum engine: ldpc5g_onchip.hip keeps the check-node state COMPRESSED (M1, M2, index,
// signs - 12 B per check node) because messages + LLRs + totals of C2 do not fit in LDS together; the price
// is that every edge's c2v is rebuilt from the compressed state twice per iteration (27 VALU operations
// per edge, the kernel is 85 % VALU busy).  With the channel LLRs moved to L2 the 158 KB of messages fit
// by themselves, and the per-edge work drops to
//   CN: read v2c, m2 = med3(m1, m2, |v|), m1 = min(m1, |v|), sign ^= v;  then  mag = (|v| == m1) ? a2 : a1,
//       c2v = ((v ^ sign) & msb) | mag, write                                   (~9 VALU)
//   VN: read c2v, x += c;  then  v2c = med3(x - c, -L, L), write                (~5 VALU)
// Same arithmetic as the compressed engine and the oracle (bit-exact): the second minimum is tracked with
// multiplicity, a unique minimum gives min_e = (m2 - m1) + m1 (decoding.py:863), ties give min_e = m1;
// v2c is never -0 (the channel LLR is never -0), so the raw sign bit equals (v2c < 0).
//
// Layout, work split and tables are those of ldpc5g_onchip_bp.hip (edge blocks indexed by the check node's
// lifted copy; (row, chunk) / (column, chunk) items balanced over the waves on the host); every row /
// column runs an instantiation for its exact degree, table entries are wave-uniform scalars that feed
// the VALU operations directly (no scalar unpacking per edge).
#include "ldpc5g.h"
#include "bp_math.h"

namespace samd {

// LDS is addressed by plain byte offsets (address space 3): the kernel has no static __shared__ data, so
// the dynamic segment starts at offset 0 (checked once per workgroup) and no base has to be added per access.
typedef __attribute__((address_space(3))) float lds_f32;
__device__ __forceinline__ float lds_ld(unsigned a) { return *(lds_f32*)(uintptr_t)a; }
__device__ __forceinline__ void lds_st(unsigned a, float v) { *(lds_f32*)(uintptr_t)a = v; }

__device__ __forceinline__ float ms_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// one check node per lane and chunk: row of exact degree D, its messages are D lane-contiguous blocks from byte a0.
// NCH: consecutive 64-lane chunks of lifted copies handled in one pass (lane z and lane z+64 share every
// scalar and every address register - the second chunk is the same address + 256 bytes): two independent
// dependency chains per wave and half as many work items (their dispatch is scalar overhead).
// FUSE1: the row's last edge goes to a degree-1 variable node of the same lane (identity block of the base
// graph's extension part).  Its VN update - x = c2v + llr, v2c = clip(x - c2v) - is done right here, so
// these columns (42 of C2's 68) never appear in the VN phase; llr_v points at that VN's channel LLR.
// MODE: SAMD_CN_MINSUM (min-sum and, through `offset`, offset-min-sum: the arithmetic below) or one of the boxplus
// rules (bp_math.h's cn_update_col on the D messages of a chunk - the same function, hence the same bits, as the HBM
// engine and ldpc5g_onchip_bp.hip).
template <int D, int NCH, bool FUSE1, int MODE>
__device__ __forceinline__ void ms_cn_row(unsigned a0, unsigned z4, float llr_max, float offset,
                                          float* __restrict__ llr_v, bool last) {
  float v[NCH][D];
  unsigned a[D];
  float lf[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) lf[h] = FUSE1 ? llr_v[64 * h] : 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    a[i] = i ? a[i - 1] + z4 : a0;
#pragma unroll
    for (int h = 0; h < NCH; ++h) v[h][i] = lds_ld(a[i] + 256u * h);
  }
  if constexpr (MODE != SAMD_CN_MINSUM) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) cn_update_col<MODE, D>(v[h], D, llr_max, 0.f);    // in place: v[h][i] = c2v
#pragma unroll
    for (int i = 0; i < D; ++i) {
      float c2v[NCH];
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        c2v[h] = v[h][i];
        if (FUSE1 && i == D - 1) {
          const float x = c2v[h] + lf[h];
          if (last) llr_v[64 * h] = x;
          c2v[h] = ms_med3(x - c2v[h], -llr_max, llr_max);
        }
      }
#pragma unroll
      for (int h = 0; h < NCH; ++h) lds_st(a[i] + 256u * h, c2v[h]);
    }
    return;
  }
  float m1[NCH], m2[NCH];
  unsigned sx[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) { m1[h] = INFINITY; m2[h] = INFINITY; sx[h] = 0u; }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      m2[h] = ms_med3(m1[h], m2[h], fabsf(v[h][i]));      // second smallest, with multiplicity
      m1[h] = ms_med3(m1[h], fabsf(v[h][i]), 0.f);        // = min(m1, |v|) for non-negative values, one operation
      sx[h] ^= __float_as_uint(v[h][i]);                  // bit 31 = node sign
    }
  float a1[NCH], a2[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    // unique minimum <=> m2 > m1; (m2 - m1) + m1 is the reference's arithmetic (decoding.py:863)
    const float min_e = (m2[h] > m1[h]) ? ((m2[h] - m1[h]) + m1[h]) : m1[h];
    a1[h] = ms_med3(m1[h] - offset, 0.f, llr_max);      // plain min-sum: offset = 0 (exact)
    a2[h] = ms_med3(min_e - offset, 0.f, llr_max);
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float c2v[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const float mag = (fabsf(v[h][i]) == m1[h]) ? a2[h] : a1[h];
      const unsigned sg = (__float_as_uint(v[h][i]) ^ sx[h]) & 0x80000000u;   // own sign x node sign
      c2v[h] = __uint_as_float(sg | __float_as_uint(mag));
      if (FUSE1 && i == D - 1) {
        const float x = c2v[h] + lf[h];             // (0 + c2v) + llr; llr is never -0, so 0 + c2v needs no operation
        if (last) llr_v[64 * h] = x;
        c2v[h] = ms_med3(x - c2v[h], -llr_max, llr_max);   // the slot now holds the next v2c
      }
    }
#pragma unroll
    for (int h = 0; h < NCH; ++h) lds_st(a[i] + 256u * h, c2v[h]);   // adjacent: one ds_write2st64_b32 per pair
  }
}

// one variable node per lane and chunk, column of exact degree D.  ent[2i] = edge block byte offset,
// ent[2i+1] = 4 shift; zwv: 4Z-1 (POW2) or 4Z, in a VGPR so that (t & zw) | base is one v_and_or_b32
// l0 / l1: channel LLRs of the lane's VN in chunk 0 / 1 (fetched by the caller one item ahead)
typedef float ms_f32x2 __attribute__((ext_vector_type(2)));

template <int D, int NCH, bool POW2, bool INIT>
__device__ __forceinline__ void ms_vn_col(const int32_t* __restrict__ ent, unsigned zz4, unsigned zwv,
                                          float* __restrict__ llr_v, float l0, float l1, float llr_max, bool last) {
  unsigned a[NCH][D];
  float c[NCH][D];
  float l[NCH], x[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) { l[h] = h ? l1 : l0; x[h] = 0.f; }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const unsigned t = zz4 + 256u * h - (unsigned)ent[2 * i + 1];
      // edge blocks are aligned to 4Z when Z is a power of two: (t mod 4Z) | base
      a[h][i] = POW2 ? ((t & zwv) | (unsigned)ent[2 * i]) : (min(t, t + zwv) + (unsigned)ent[2 * i]);
      if (INIT) lds_st(a[h][i], l[h]);
      else c[h][i] = lds_ld(a[h][i]);
    }
  if (INIT) return;
  if constexpr (NCH == 2) {
    // both chunks of an edge in one packed-fp32 operation (v_pk_add_f32: two IEEE additions per issue slot, the
    // same results as two v_add_f32): x += c and x - c cost one VALU operation per edge instead of two
    ms_f32x2 xv = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < D; ++i) xv += ms_f32x2{c[0][i], c[1][i]};
    xv += ms_f32x2{l[0], l[1]};
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const ms_f32x2 e = xv - ms_f32x2{c[0][i], c[1][i]};
      lds_st(a[0][i], ms_med3(e.x, -llr_max, llr_max));
      lds_st(a[1][i], ms_med3(e.y, -llr_max, llr_max));
    }
    if (last) { llr_v[0] = xv.x; llr_v[64] = xv.y; }
  } else {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
#pragma unroll
      for (int i = 0; i < D; ++i) x[h] += c[h][i];
      x[h] += l[h];
#pragma unroll
      for (int i = 0; i < D; ++i) lds_st(a[h][i], ms_med3(x[h] - c[h][i], -llr_max, llr_max));
      if (last) llr_v[64 * h] = x[h];
    }
  }
}


template <bool POW2, bool INIT>
__device__ __forceinline__ void ms_vn_item(const int32_t* __restrict__ ent, int d, unsigned zz4, unsigned zwv,
                                           float* __restrict__ llr_v, float l0, float l1, float llr_max, bool last) {
#define SAMD_MS_VN(D) case D: ms_vn_col<D, 1, POW2, INIT>(ent, zz4, zwv, llr_v, l0, l1, llr_max, last); break
#define SAMD_MS_VN2(D) case 32 + D: ms_vn_col<D, 2, POW2, INIT>(ent, zz4, zwv, llr_v, l0, l1, llr_max, last); break
  switch (d) {                                                       // degree | pair << 5
    SAMD_MS_VN(1); SAMD_MS_VN(2); SAMD_MS_VN(3); SAMD_MS_VN(4); SAMD_MS_VN(5); SAMD_MS_VN(6); SAMD_MS_VN(7);
    SAMD_MS_VN(8); SAMD_MS_VN(9); SAMD_MS_VN(10); SAMD_MS_VN(11); SAMD_MS_VN(12); SAMD_MS_VN(13); SAMD_MS_VN(14);
    SAMD_MS_VN(15); SAMD_MS_VN(16); SAMD_MS_VN(17); SAMD_MS_VN(18); SAMD_MS_VN(19); SAMD_MS_VN(20); SAMD_MS_VN(21);
    SAMD_MS_VN(22); SAMD_MS_VN(23); SAMD_MS_VN(24); SAMD_MS_VN(25); SAMD_MS_VN(26); SAMD_MS_VN(27); SAMD_MS_VN(28);
    SAMD_MS_VN(29); SAMD_MS_VN(30);
    SAMD_MS_VN2(1); SAMD_MS_VN2(2); SAMD_MS_VN2(3); SAMD_MS_VN2(4); SAMD_MS_VN2(5); SAMD_MS_VN2(6); SAMD_MS_VN2(7);
    SAMD_MS_VN2(8); SAMD_MS_VN2(9); SAMD_MS_VN2(10); SAMD_MS_VN2(11); SAMD_MS_VN2(12);
    default: break;
  }
#undef SAMD_MS_VN
#undef SAMD_MS_VN2
}

// list entries are self-contained (no dependent table look-ups) and the next one is fetched while the current
// item runs:  VN (c | chunk<<8 | degree<<16,  dword offset of the column's edge table),
//             CN (row block byte offset | degree<<18 | fused<<23 | pair<<24,  r | chunk<<8 | fused column<<16)
// a pair item covers chunks (chunk, chunk+1), all 128 lanes valid; VN degree field = degree | pair<<5
// vn_ptr = [NW+1 offsets of the per-iteration lists | NW+1 offsets of the fused degree-1 columns (init only)]
#ifdef SAMD_MS_TRACE
// Development aid (tools/ms_trace.py; not part of the product build): per-wave timestamps of workgroup 0 at the phase
// boundaries of iterations 2..5 -> where the time of an iteration goes (CN items, barrier wait, VN items, barrier wait).
__device__ unsigned long long* g_ms_trace = nullptr;
#define SAMD_TRACE_MARK(slot)                                                                               \
  if (g_ms_trace && blockIdx.x == 0 && it >= 2 && it < 6 && lane == 0)                                        \
    g_ms_trace[((it - 2) * 5 + (slot)) * NW + w] = __builtin_readcyclecounter();
#else
#define SAMD_TRACE_MARK(slot)
#endif

template <bool POW2, int NW, bool LLRG, int MODE>
__global__ __launch_bounds__(NW * 64) void ldpc5g_decode_ms_kernel(
    const float* __restrict__ llr_in, float* __restrict__ out, float* __restrict__ llr_ws, RateMatch p, int n_cn,
    int nbu, int batch, int num_iter, float llr_max, float offset, int hard_out, int return_infobits,
    int msg_floats, const int32_t* __restrict__ col_ent, const int32_t* __restrict__ cn_ptr,
    const int2* __restrict__ cn_list, const int32_t* __restrict__ vn_ptr, const int2* __restrict__ vn_list) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((unsigned)(size_t)(lds_f32*)smem != 0u) __builtin_trap();      // see lds_ld
  constexpr int NT = NW * 64;
  const unsigned z = (unsigned)p.z, z4 = 4u * z;
  const unsigned zw = POW2 ? z4 - 1u : z4;
  unsigned zwv;
  asm volatile("v_mov_b32 %0, %1" : "=v"(zwv) : "s"(zw));
  const int n_vn = p.n_vn;
  const int nx = nbu * (int)z;
  float* llr = LLRG ? llr_ws + (size_t)blockIdx.x * nx : smem + msg_floats;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = cn_ptr[w], c1 = cn_ptr[w + 1];
  const int v0 = vn_ptr[w], v1 = vn_ptr[w + 1];
  const int f0 = vn_ptr[NW + 1 + w], f1 = vn_ptr[NW + 2 + w];

  for (int b = blockIdx.x; b < batch; b += gridDim.x) {
    const float* row = llr_in + (size_t)b * p.n;
    // decoding.py:552-565: clip, then logits -> LLR; "+ 0.f" turns -0 into +0 (numerically the same LLR)
    // so that no total and no v2c is ever -0 and the sign bit of a v2c equals (v2c < 0)
    for (int v = tid; v < nx; v += NT)
      llr[v] = (v < n_vn) ? (-1.f * clampf(recover_llr(p, row, v, llr_max), -llr_max, llr_max)) + 0.f : 0.f;
    __syncthreads();
    for (int seg = 0; seg < 2; ++seg) {                              // v2c of iteration 0 = channel LLR, all columns
      const int t0 = seg ? f0 : v0, t1 = seg ? f1 : v1;
      for (int t = t0; t < t1; ++t) {
        const int2 e = vn_list[t];
        const int d0 = __builtin_amdgcn_readfirstlane(e.x), d1 = __builtin_amdgcn_readfirstlane(e.y);
        const int c = d0 & 0xFF;
        const unsigned zz = (unsigned)(((d0 >> 8) & 0xFF) * 64 + lane);
        const int vn = c * (int)z + (int)zz;
        const bool pr = (d0 >> 21) & 1;
        if (pr || (zz < z && vn < n_vn))
          ms_vn_item<POW2, true>(col_ent + d1, (d0 >> 16) & 63, 4u * zz, zwv, llr + vn, llr[vn], pr ? llr[vn + 64] : 0.f, llr_max, false);
      }
    }
    __syncthreads();

    for (int it = 0; it < num_iter; ++it) {
      const bool last = (it == num_iter - 1);
      SAMD_TRACE_MARK(0)
      // vn_fetch: descriptor and channel LLRs of this wave's first VN item, in flight during the CN phase
      int2 vfirst = make_int2(0, 0);
      float lf0 = 0.f, lf1 = 0.f;
      if (v0 < v1) {
        vfirst = vn_list[v0];
        const int e0 = __builtin_amdgcn_readfirstlane(vfirst.x);
        const int vn2 = (e0 & 0xFF) * (int)z + ((e0 >> 8) & 0xFF) * 64 + lane;
        const bool pr2 = (e0 >> 21) & 1;
        if (pr2 || vn2 < n_vn) lf0 = llr[vn2];
        if (pr2) lf1 = llr[vn2 + 64];
      }
      {
        int2 nxt = c0 < c1 ? cn_list[c0] : make_int2(0, 0);
        for (int t = c0; t < c1; ++t) {
          const unsigned ro = (unsigned)__builtin_amdgcn_readfirstlane(nxt.x);
          const int d1 = __builtin_amdgcn_readfirstlane(nxt.y);
          if (t + 1 < c1) nxt = cn_list[t + 1];
          const int r = d1 & 0xFF;
          onchip_setprio(d1 >> 24);
          const unsigned zz = (unsigned)(((d1 >> 8) & 0xFF) * 64 + lane);
          const unsigned a0 = (ro & 0x3FFFFu) + 4u * zz;
          if (((ro >> 24) & 1u) || (zz < z && (unsigned)r * z + zz < (unsigned)n_cn)) {
            float* lv = llr + ((d1 >> 16) & 0xFF) * (int)z + (int)zz;   // channel LLR of the fused degree-1 VN
#define SAMD_MS_CN(D) case D: ms_cn_row<D, 1, false, MODE>(a0, z4, llr_max, offset, lv, last); break
#define SAMD_MS_CNF(D) case 32 + D: ms_cn_row<D, 1, true, MODE>(a0, z4, llr_max, offset, lv, last); break
#define SAMD_MS_CN2(D) case 64 + D: ms_cn_row<D, 2, false, MODE>(a0, z4, llr_max, offset, lv, last); break
#define SAMD_MS_CNF2(D) case 96 + D: ms_cn_row<D, 2, true, MODE>(a0, z4, llr_max, offset, lv, last); break
            switch (ro >> 18) {                                      // degree | fused << 5 | pair << 6
              SAMD_MS_CN(3); SAMD_MS_CN(4); SAMD_MS_CN(5); SAMD_MS_CN(6); SAMD_MS_CN(7); SAMD_MS_CN(8); SAMD_MS_CN(9);
              SAMD_MS_CN(10); SAMD_MS_CN(19);
              SAMD_MS_CNF(3); SAMD_MS_CNF(4); SAMD_MS_CNF(5); SAMD_MS_CNF(6); SAMD_MS_CNF(7); SAMD_MS_CNF(8);
              SAMD_MS_CNF(9); SAMD_MS_CNF(10);
              SAMD_MS_CN2(3); SAMD_MS_CN2(4); SAMD_MS_CN2(5); SAMD_MS_CN2(6); SAMD_MS_CN2(7); SAMD_MS_CN2(8);
              SAMD_MS_CN2(9); SAMD_MS_CN2(10); SAMD_MS_CN2(19);
              SAMD_MS_CNF2(3); SAMD_MS_CNF2(4); SAMD_MS_CNF2(5); SAMD_MS_CNF2(6); SAMD_MS_CNF2(7); SAMD_MS_CNF2(8);
              SAMD_MS_CNF2(9); SAMD_MS_CNF2(10);
              default: break;
            }
#undef SAMD_MS_CN2
#undef SAMD_MS_CNF2
#undef SAMD_MS_CN
#undef SAMD_MS_CNF
          } else if (zz < z) {
            // pruned check node of the last, partial base row: its edges do not exist - keep their slots at 0
            for (unsigned i = 0; i < ((ro >> 18) & 31u); ++i) lds_st(a0 + i * z4, 0.f);
          }
        }
      }
      SAMD_TRACE_MARK(1)
      __syncthreads();
      SAMD_TRACE_MARK(2)
      {
        // the channel LLRs of an item are fetched one item ahead (the first item's before the CN phase, see
        // vn_fetch above): an L2 round trip is longer than a whole VN item
        int2 cur = vfirst;
        float l0 = lf0, l1 = lf1;
        for (int t = v0; t < v1; ++t) {
          const int d0 = __builtin_amdgcn_readfirstlane(cur.x), d1 = __builtin_amdgcn_readfirstlane(cur.y);
          int2 nxt = make_int2(0, 0);
          float n0 = 0.f, n1 = 0.f;
          if (t + 1 < v1) {
            nxt = vn_list[t + 1];
            const int e0 = __builtin_amdgcn_readfirstlane(nxt.x);
            const int vn2 = (e0 & 0xFF) * (int)z + ((e0 >> 8) & 0xFF) * 64 + lane;
            const bool pr2 = (e0 >> 21) & 1;
            if (pr2 || vn2 < n_vn) n0 = llr[vn2];
            if (pr2) n1 = llr[vn2 + 64];
          }
          const int c = d0 & 0xFF;
          const unsigned zz = (unsigned)(((d0 >> 8) & 0xFF) * 64 + lane);
          const int vn = c * (int)z + (int)zz;
          onchip_setprio(d0 >> 24);
          if (((d0 >> 21) & 1) || (zz < z && vn < n_vn))
            ms_vn_item<POW2, false>(col_ent + d1, (d0 >> 16) & 63, 4u * zz, zwv, llr + vn, l0, l1, llr_max, last);
          cur = nxt; l0 = n0; l1 = n1;
        }
      }
      SAMD_TRACE_MARK(3)
      __syncthreads();
      SAMD_TRACE_MARK(4)
    }
    // ---------------- output (decoding.py:620-626, 1486-1531); llr[] now holds the marginals
    if (return_infobits) {
      float* o = out + (size_t)b * p.k;
      for (int v = tid; v < p.k; v += NT) {
        const float x = clampf(llr[v], -llr_max, llr_max);
        o[v] = hard_out ? ((0.f >= x) ? 1.f : 0.f) : -1.f * x;
      }
    } else {
      float* o = out + (size_t)b * p.n;
      for (int i = tid; i < p.n; i += NT) {
        const float x = clampf(llr[short_to_full(p, out_to_short(p, i))], -llr_max, llr_max);
        o[i] = hard_out ? ((0.f >= x) ? 1.f : 0.f) : -1.f * x;
      }
    }
    __syncthreads();
  }
}

#ifdef SAMD_MS_TRACE
}  // namespace samd
extern "C" int samd_debug_set_ms_trace(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(samd::g_ms_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
namespace samd {
#endif

int launch_onchip_ms(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                     float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                     size_t workspace_bytes, hipStream_t st) {
  if (!h->bp_ok || !h->ms_col_ent || !h->ms_cn_list || !h->ms_vn_list || !h->ms_vn_ptr) {
    set_error("messages of this code do not fit in LDS");
    return SAMD_ERR_UNSUPPORTED;
  }
  float* llr_ws = nullptr;
  if (h->bp_llr_global) {
    if (!workspace || workspace_bytes < onchip_bp_workspace_bytes(h, batch)) {
      set_error("workspace too small (samd_ldpc5g_decode_workspace_bytes)");
      return SAMD_ERR_WORKSPACE;
    }
    llr_ws = reinterpret_cast<float*>(align_up((size_t)workspace, 256));
  }
  const bool pow2 = (h->z & (h->z - 1)) == 0;
  typedef void (*kern_t)(const float*, float*, float*, RateMatch, int, int, int, int, float, float, int, int, int,
                         const int32_t*, const int32_t*, const int2*, const int32_t*, const int2*);
#define SAMD_MS_K(NWV, G, M) ldpc5g_decode_ms_kernel<false, NWV, G, M>, ldpc5g_decode_ms_kernel<true, NWV, G, M>
#define SAMD_MS_TAB(M) {SAMD_MS_K(16, false, M), SAMD_MS_K(8, false, M), SAMD_MS_K(4, false, M), \
                        SAMD_MS_K(2, false, M),  SAMD_MS_K(1, false, M), SAMD_MS_K(16, true, M)}
  static const kern_t kerns_all[3][12] = {SAMD_MS_TAB(SAMD_CN_MINSUM), SAMD_MS_TAB(SAMD_CN_BOXPLUS_PHI),
                                          SAMD_MS_TAB(SAMD_CN_BOXPLUS)};
#undef SAMD_MS_TAB
#undef SAMD_MS_K
  const kern_t* kerns = kerns_all[cn_mode == SAMD_CN_BOXPLUS_PHI ? 1 : cn_mode == SAMD_CN_BOXPLUS ? 2 : 0];
  const int nw = h->bp_waves;
  const int wi = h->bp_llr_global ? 5 : (nw == 16 ? 0 : nw == 8 ? 1 : nw == 4 ? 2 : nw == 2 ? 3 : 4);
  const int ki = 2 * wi + (pow2 ? 1 : 0);
  SAMD_HIP_CHECK(hipFuncSetAttribute((const void*)kerns[ki], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int nbu = (h->n_vn + h->z - 1) / h->z;
  const RateMatch rm{h->k, h->n, h->z, h->k_ldpc, h->n_vn, h->m_int};
  const float off = (cn_mode == SAMD_CN_OFFSET_MINSUM) ? offset : 0.f;
  hipLaunchKernelGGL(kerns[ki], dim3(onchip_bp_grid(h, batch)), dim3(nw * 64), onchip_bp_lds_bytes(h), st, llr, out,
                     llr_ws, rm, h->n_cn, nbu, batch, num_iter, llr_max, off, hard_out, return_infobits,
                     h->bp_edges * h->z, h->ms_col_ent, h->ms_cn_ptr, reinterpret_cast<const int2*>(h->ms_cn_list),
                     h->ms_vn_ptr, reinterpret_cast<const int2*>(h->ms_vn_list));
  return launch_status();
}

}  // namespace samd
