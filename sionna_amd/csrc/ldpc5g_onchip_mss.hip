// On-chip min-sum decoder with explicit messages for codes whose messages do NOT fit in LDS: the edge blocks
// of the first base rows stay in LDS, those of the last rows live in the workgroup's L2 workspace row.
//
// Same algorithm, arithmetic and results as ldpc5g_onchip_ms.hip (reference
// src/sionna/phy/fec/ldpc/decoding.py:1427-1536, 416-524, 681-953); this file only adds the second home of
// a message.  Why rows and why the LAST rows: a check node's messages are lane-contiguous, so a spilled row
// is read and written with fully coalesced global accesses; the extension rows at the bottom of the 5G base
// graphs have degree 3-5 plus one degree-1 column that is fused into the row (never touched by the VN phase);
// and because every LDS row precedes every spilled row, a variable node sums "LDS edges, then L2 edges" in
// ascending check-node order - the summation order of the oracle - with the LDS part in an instantiation for
// its exact degree and the L2 part in a short loop.
// The compressed-state engine (ldpc5g_onchip.hip) moves 12 B per check node and rebuilds every c2v twice
// (27 VALU operations per edge); here a spilled edge costs 16 B of L2 traffic per iteration and 15 operations.
// Used while at most ~27 % of the edges are spilled (codes with 160 KB < 4 E Z <~ 215 KB, e.g. k=6144 rate 2/3,
// k=5632 rate 1/2): +9 ... +16 % over the compressed engine; larger codes stay on the compressed engine.
#include "ldpc5g.h"
#include "bp_math.h"

namespace samd {

typedef __attribute__((address_space(3))) float mss_lds_f32;
__device__ __forceinline__ float mss_lds_ld(unsigned a) { return *(mss_lds_f32*)(uintptr_t)a; }
__device__ __forceinline__ void mss_lds_st(unsigned a, float v) { *(mss_lds_f32*)(uintptr_t)a = v; }
__device__ __forceinline__ float mss_med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// GLB: the row's blocks are at gbase + a (global), else at LDS byte offset a.  See ms_cn_row for the rest.
// MODE: SAMD_CN_MINSUM (also offset min-sum: offset is a runtime value) or a boxplus rule (bp_math.h arithmetic,
// identical bits to the HBM engine).
template <int D, int NCH, bool FUSE1, bool GLB, int MODE>
__device__ __forceinline__ void mss_cn_row(unsigned a0, unsigned z4, char* __restrict__ gbase, float llr_max,
                                           float offset, float* __restrict__ llr_v, bool last) {
  float v[NCH][D];
  unsigned a[D];
  float lf[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) lf[h] = FUSE1 ? llr_v[64 * h] : 0.f;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    a[i] = i ? a[i - 1] + z4 : a0;
#pragma unroll
    for (int h = 0; h < NCH; ++h)
      v[h][i] = GLB ? *reinterpret_cast<const float*>(gbase + a[i] + 256u * h) : mss_lds_ld(a[i] + 256u * h);
  }
  if constexpr (MODE != SAMD_CN_MINSUM) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) cn_update_col<MODE, D>(v[h], D, llr_max, 0.f);
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        float c2v = v[h][i];
        if (FUSE1 && i == D - 1) {
          const float x = c2v + lf[h];              // (0 + c2v) + llr up to the sign of a zero
          if (last) llr_v[64 * h] = x;
          c2v = clampf(-1.f * c2v + x, -llr_max, llr_max);
        }
        if (GLB) *reinterpret_cast<float*>(gbase + a[i] + 256u * h) = c2v;
        else mss_lds_st(a[i] + 256u * h, c2v);
      }
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
      m2[h] = mss_med3(m1[h], m2[h], fabsf(v[h][i]));
      m1[h] = mss_med3(m1[h], fabsf(v[h][i]), 0.f);
      sx[h] ^= __float_as_uint(v[h][i]);
    }
  float a1[NCH], a2[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    const float min_e = (m2[h] > m1[h]) ? ((m2[h] - m1[h]) + m1[h]) : m1[h];      // decoding.py:863
    a1[h] = mss_med3(m1[h] - offset, 0.f, llr_max);
    a2[h] = mss_med3(min_e - offset, 0.f, llr_max);
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    float c2v[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const float mag = (fabsf(v[h][i]) == m1[h]) ? a2[h] : a1[h];
      const unsigned sg = (__float_as_uint(v[h][i]) ^ sx[h]) & 0x80000000u;
      c2v[h] = __uint_as_float(sg | __float_as_uint(mag));
      if (FUSE1 && i == D - 1) {
        const float x = c2v[h] + lf[h];
        if (last) llr_v[64 * h] = x;
        c2v[h] = mss_med3(x - c2v[h], -llr_max, llr_max);
      }
    }
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      if (GLB) *reinterpret_cast<float*>(gbase + a[i] + 256u * h) = c2v[h];
      else mss_lds_st(a[i] + 256u * h, c2v[h]);
    }
  }
}

// column: DL edges in LDS (exact-degree instantiation), then dg edges in the L2 row (loop).
// ent[2i] = block byte offset (LDS edges first, then L2 edges, each in ascending row order), ent[2i+1] = 4 shift
template <int DL, int NCH, bool POW2, bool INIT>
__device__ __forceinline__ void mss_vn_col(const int32_t* __restrict__ ent, int dg, unsigned zz4, unsigned zwv,
                                           char* __restrict__ gbase, float* __restrict__ llr_v, float l0, float l1,
                                           float llr_max, bool last) {
  constexpr int DA = DL > 0 ? DL : 1;
  unsigned a[NCH][DA];
  float c[NCH][DA];
  float l[NCH], x[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) { l[h] = h ? l1 : l0; x[h] = 0.f; }
#pragma unroll
  for (int i = 0; i < DL; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const unsigned t = zz4 + 256u * h - (unsigned)ent[2 * i + 1];
      a[h][i] = POW2 ? ((t & zwv) | (unsigned)ent[2 * i]) : (min(t, t + zwv) + (unsigned)ent[2 * i]);
      if (INIT) {
        mss_lds_st(a[h][i], l[h]);
      } else {
        c[h][i] = mss_lds_ld(a[h][i]);
        x[h] += c[h][i];
      }
    }
  const int32_t* eg = ent + 2 * DL;
#pragma unroll 4
  for (int j = 0; j < dg; ++j) {
    const unsigned s4 = (unsigned)eg[2 * j + 1], base = (unsigned)eg[2 * j];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const unsigned t = zz4 + 256u * h - s4;
      const unsigned off = (POW2 ? (t & zwv) : min(t, t + zwv)) + base;
      if (INIT) *reinterpret_cast<float*>(gbase + off) = l[h];
      else x[h] += *reinterpret_cast<const float*>(gbase + off);
    }
  }
  if (INIT) return;
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    x[h] += l[h];
#pragma unroll
    for (int i = 0; i < DL; ++i) mss_lds_st(a[h][i], mss_med3(x[h] - c[h][i], -llr_max, llr_max));
  }
#pragma unroll 4
  for (int j = 0; j < dg; ++j) {
    const unsigned s4 = (unsigned)eg[2 * j + 1], base = (unsigned)eg[2 * j];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const unsigned t = zz4 + 256u * h - s4;
      const unsigned off = (POW2 ? (t & zwv) : min(t, t + zwv)) + base;
      float* q = reinterpret_cast<float*>(gbase + off);
      *q = mss_med3(x[h] - *q, -llr_max, llr_max);
    }
  }
  if (last) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) llr_v[64 * h] = x[h];
  }
}

template <bool POW2, bool INIT>
__device__ __forceinline__ void mss_vn_item(const int32_t* __restrict__ ent, int dl, int dg, unsigned zz4, unsigned zwv,
                                            char* __restrict__ gbase, float* __restrict__ llr_v, float l0, float l1,
                                            float llr_max, bool last) {
#define SAMD_MSS_VN(D) case D: mss_vn_col<D, 1, POW2, INIT>(ent, dg, zz4, zwv, gbase, llr_v, l0, l1, llr_max, last); break
#define SAMD_MSS_VN2(D) case 32 + D: mss_vn_col<D, 2, POW2, INIT>(ent, dg, zz4, zwv, gbase, llr_v, l0, l1, llr_max, last); break
  switch (dl) {                                                      // LDS degree | pair << 5
    SAMD_MSS_VN(0); SAMD_MSS_VN(1); SAMD_MSS_VN(2); SAMD_MSS_VN(3); SAMD_MSS_VN(4); SAMD_MSS_VN(5); SAMD_MSS_VN(6);
    SAMD_MSS_VN(7); SAMD_MSS_VN(8); SAMD_MSS_VN(9); SAMD_MSS_VN(10); SAMD_MSS_VN(11); SAMD_MSS_VN(12); SAMD_MSS_VN(13);
    SAMD_MSS_VN(14); SAMD_MSS_VN(15); SAMD_MSS_VN(16); SAMD_MSS_VN(17); SAMD_MSS_VN(18); SAMD_MSS_VN(19); SAMD_MSS_VN(20);
    SAMD_MSS_VN(21); SAMD_MSS_VN(22); SAMD_MSS_VN(23); SAMD_MSS_VN(24); SAMD_MSS_VN(25); SAMD_MSS_VN(26); SAMD_MSS_VN(27);
    SAMD_MSS_VN(28); SAMD_MSS_VN(29); SAMD_MSS_VN(30);
    SAMD_MSS_VN2(0); SAMD_MSS_VN2(1); SAMD_MSS_VN2(2); SAMD_MSS_VN2(3); SAMD_MSS_VN2(4); SAMD_MSS_VN2(5); SAMD_MSS_VN2(6);
    SAMD_MSS_VN2(7); SAMD_MSS_VN2(8); SAMD_MSS_VN2(9); SAMD_MSS_VN2(10); SAMD_MSS_VN2(11); SAMD_MSS_VN2(12);
    default: break;
  }
#undef SAMD_MSS_VN
#undef SAMD_MSS_VN2
}

// list entries (two dwords):
//   CN  byte offset of the row's first block (LDS or L2 row),
//       r | chunk<<8 | fused column<<11 | degree<<19 | fused<<24 | pair<<25 | in L2<<26
//   VN  c | chunk<<8 | LDS degree<<16 | pair<<21 | L2 degree<<22,  dword offset of the column's edge table
// vn_ptr = [17 offsets of the per-iteration lists | 17 offsets of the fused degree-1 columns (init only)]
// workspace row of a workgroup: [channel LLRs nx | spilled messages g_floats]
template <bool POW2, int MODE>
__global__ __launch_bounds__(1024) void ldpc5g_decode_mss_kernel(
    const float* __restrict__ llr_in, float* __restrict__ out, float* __restrict__ ws, RateMatch p, int n_cn,
    int nbu, int batch, int num_iter, float llr_max, float offset, int hard_out, int return_infobits, int g_floats,
    const int32_t* __restrict__ col_ent, const int32_t* __restrict__ cn_ptr, const int2* __restrict__ cn_list,
    const int32_t* __restrict__ vn_ptr, const int2* __restrict__ vn_list) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((unsigned)(size_t)(mss_lds_f32*)smem != 0u) __builtin_trap();    // LDS is addressed by plain offsets
  constexpr int NW = 16, NT = NW * 64;
  const unsigned z = (unsigned)p.z, z4 = 4u * z;
  const unsigned zw = POW2 ? z4 - 1u : z4;
  unsigned zwv;
  asm volatile("v_mov_b32 %0, %1" : "=v"(zwv) : "s"(zw));
  const int n_vn = p.n_vn;
  const int nx = nbu * (int)z;
  float* llr = ws + (size_t)blockIdx.x * ((size_t)nx + g_floats);
  char* gbase = reinterpret_cast<char*>(llr + nx);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = cn_ptr[w], c1 = cn_ptr[w + 1];
  const int v0 = vn_ptr[w], v1 = vn_ptr[w + 1];
  const int f0 = vn_ptr[NW + 1 + w], f1 = vn_ptr[NW + 2 + w];

  for (int b = blockIdx.x; b < batch; b += gridDim.x) {
    const float* row = llr_in + (size_t)b * p.n;
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
          mss_vn_item<POW2, true>(col_ent + d1, (d0 >> 16) & 63, (d0 >> 22) & 31, 4u * zz, zwv, gbase, llr + vn, llr[vn],
                                  pr ? llr[vn + 64] : 0.f, llr_max, false);
      }
    }
    __syncthreads();

    for (int it = 0; it < num_iter; ++it) {
      const bool last = (it == num_iter - 1);
      int2 vfirst = make_int2(0, 0);
      float lf0 = 0.f, lf1 = 0.f;
      if (v0 < v1) {                                                 // first VN item's LLRs, in flight during the CN phase
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
          onchip_setprio(d1 >> 27);
          const unsigned zz = (unsigned)(((d1 >> 8) & 7) * 64 + lane);
          const unsigned a0 = ro + 4u * zz;
          const int deg = (d1 >> 19) & 31;
          const bool glb = (d1 >> 26) & 1;
          if (((d1 >> 25) & 1) || (zz < z && (unsigned)r * z + zz < (unsigned)n_cn)) {
            float* lv = llr + ((d1 >> 11) & 0xFF) * (int)z + (int)zz;
#define SAMD_MSS_CN(D, G) \
  case (G ? 128 : 0) + D: mss_cn_row<D, 1, false, G, MODE>(a0, z4, gbase, llr_max, offset, lv, last); break; \
  case (G ? 128 : 0) + 64 + D: mss_cn_row<D, 2, false, G, MODE>(a0, z4, gbase, llr_max, offset, lv, last); break
#define SAMD_MSS_CNF(D, G) \
  case (G ? 128 : 0) + 32 + D: mss_cn_row<D, 1, true, G, MODE>(a0, z4, gbase, llr_max, offset, lv, last); break; \
  case (G ? 128 : 0) + 96 + D: mss_cn_row<D, 2, true, G, MODE>(a0, z4, gbase, llr_max, offset, lv, last); break
            switch (deg | (((d1 >> 24) & 3) << 5) | ((int)glb << 7)) {   // degree | fused<<5 | pair<<6 | L2<<7
              SAMD_MSS_CN(3, false); SAMD_MSS_CN(4, false); SAMD_MSS_CN(5, false); SAMD_MSS_CN(6, false);
              SAMD_MSS_CN(7, false); SAMD_MSS_CN(8, false); SAMD_MSS_CN(9, false); SAMD_MSS_CN(10, false);
              SAMD_MSS_CN(19, false);
              SAMD_MSS_CNF(3, false); SAMD_MSS_CNF(4, false); SAMD_MSS_CNF(5, false); SAMD_MSS_CNF(6, false);
              SAMD_MSS_CNF(7, false); SAMD_MSS_CNF(8, false); SAMD_MSS_CNF(9, false); SAMD_MSS_CNF(10, false);
              SAMD_MSS_CN(3, true); SAMD_MSS_CN(4, true); SAMD_MSS_CN(5, true); SAMD_MSS_CN(6, true);
              SAMD_MSS_CN(7, true); SAMD_MSS_CN(8, true); SAMD_MSS_CN(9, true); SAMD_MSS_CN(10, true);
              SAMD_MSS_CNF(3, true); SAMD_MSS_CNF(4, true); SAMD_MSS_CNF(5, true); SAMD_MSS_CNF(6, true);
              SAMD_MSS_CNF(7, true); SAMD_MSS_CNF(8, true); SAMD_MSS_CNF(9, true); SAMD_MSS_CNF(10, true);
              default: break;
            }
#undef SAMD_MSS_CN
#undef SAMD_MSS_CNF
          } else if (zz < z) {
            // pruned check node of the last, partial base row: its edges do not exist - keep their slots at 0
            for (int i = 0; i < deg; ++i) {
              if (glb) *reinterpret_cast<float*>(gbase + a0 + (unsigned)i * z4) = 0.f;
              else mss_lds_st(a0 + (unsigned)i * z4, 0.f);
            }
          }
        }
      }
      __syncthreads();
      {
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
          onchip_setprio(d0 >> 27);
          if (((d0 >> 21) & 1) || (zz < z && vn < n_vn))
            mss_vn_item<POW2, false>(col_ent + d1, (d0 >> 16) & 63, (d0 >> 22) & 31, 4u * zz, zwv, gbase, llr + vn, l0, l1,
                                     llr_max, last);
          cur = nxt; l0 = n0; l1 = n1;
        }
      }
      __syncthreads();
    }
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

// ---------------------------------------------------------------- host: tables, workspace, launch
int build_onchip_mss_tables(samd_ldpc5g* h, const std::vector<std::vector<std::pair<int, int>>>& by_row) {
  const int z = h->z;
  const int ncu = (h->n_cn + z - 1) / z, nbu = (h->n_vn + z - 1) / z;
  h->sp_ok = 0;
  if (h->bp_ok) return SAMD_OK;                               // everything fits in LDS: ldpc5g_onchip_ms.hip
  if (h->mb > 255 || h->nb > 255 || (z + 63) / 64 > 7) return SAMD_OK;
  static const int kDeg[] = {3, 4, 5, 6, 7, 8, 9, 10, 19};
  // longest prefix of rows whose messages fit in LDS
  int lds_rows = 0, e_lds = 0;
  for (int r = 0; r < ncu; ++r) {
    const int d = (int)by_row[r].size();
    if (std::find(std::begin(kDeg), std::end(kDeg), d) == std::end(kDeg)) return SAMD_OK;
    if (lds_rows == r && (size_t)(e_lds + d) * z * 4 <= 160 * 1024) { e_lds += d; lds_rows = r + 1; }
  }
  if (lds_rows < 4) return SAMD_OK;
  for (int r = lds_rows; r < ncu; ++r)
    if ((int)by_row[r].size() > 10) return SAMD_OK;          // spilled rows: extension part only
  std::vector<int32_t> row_off(h->mb, 0);
  std::vector<std::vector<std::pair<int32_t, int32_t>>> cl(h->nb), cg(h->nb);   // (block byte offset, 4 shift)
  int el = 0, eg = 0;
  for (int r = 0; r < ncu; ++r) {
    const int d = (int)by_row[r].size();
    const bool g = r >= lds_rows;
    row_off[r] = (g ? eg : el) * z * 4;
    for (int i = 0; i < d; ++i) {
      const int c = by_row[r][i].first, s = by_row[r][i].second;
      if (c >= nbu) return SAMD_OK;
      (g ? cg : cl)[c].push_back({((g ? eg : el) + i) * z * 4, s * 4});      // rows ascending
    }
    (g ? eg : el) += d;
  }
  std::vector<int32_t> col_ent, col_start(h->nb, 0);
  for (int c = 0; c < h->nb; ++c) {
    if (cl[c].size() > 30 || cg[c].size() > 31) return SAMD_OK;
    col_start[c] = (int32_t)col_ent.size();
    for (auto& e : cl[c]) { col_ent.push_back(e.first); col_ent.push_back(e.second); }
    for (auto& e : cg[c]) { col_ent.push_back(e.first); col_ent.push_back(e.second); }
  }
  col_ent.resize(col_ent.size() + 64, 0);
  std::vector<int> fused_col(h->mb, -1);
  std::vector<char> col_fused(h->nb, 0);
  for (int r = 0; r < ncu; ++r) {
    const int d = (int)by_row[r].size();
    const int c = by_row[r][d - 1].first, sft = by_row[r][d - 1].second;
    if (cl[c].size() + cg[c].size() == 1 && sft == 0 && d >= 3 && d <= 10) { fused_col[r] = c; col_fused[c] = 1; }
  }
  const int chunks = (z + 63) / 64;
  std::vector<std::pair<int, int32_t>> ci, vi, vf;
  for (int r = 0; r < ncu; ++r)
    for (int q = 0; q < chunks; ++q) {
      const int d = (int)by_row[r].size();
      const bool pair = (q + 2) * 64 <= z && r * z + (q + 2) * 64 <= h->n_cn &&
                        (fused_col[r] < 0 || fused_col[r] * z + (q + 2) * 64 <= h->n_vn);
      const int wgt = (r >= lds_rows ? 12 : 9) * d;           // L2 accesses issue more slowly than LDS ones
      if (pair) { ci.push_back({2 * wgt + 40, r | (q << 8) | (1 << 24)}); ++q; }
      else ci.push_back({wgt + 40, r | (q << 8)});
    }
  for (int c = 0; c < nbu; ++c)
    for (int q = 0; q < chunks; ++q) {
      if (c * z + q * 64 >= h->n_vn) continue;
      const int dl = (int)cl[c].size(), dg = (int)cg[c].size();
      const bool pair = dl <= 12 && (q + 2) * 64 <= z && c * z + (q + 2) * 64 <= h->n_vn;
      const int wgt = 5 * dl + 10 * dg;
      auto& dst = col_fused[c] ? vf : vi;
      if (pair) { dst.push_back({2 * wgt + 40, c | (q << 8) | (1 << 24)}); ++q; }
      else dst.push_back({wgt + 40, c | (q << 8)});
    }
  std::vector<int32_t> cp, cls, vp, vls, fp, fls, cl2, vl2;
  lpt_schedule(ci, 16, &cp, &cls);
  lpt_schedule(vi, 16, &vp, &vls);
  lpt_schedule(vf, 16, &fp, &fls);
  const std::vector<int> cprio = item_priorities(ci, cp, cls), vprio = item_priorities(vi, vp, vls);   // see ldpc5g.h
  for (int32_t o : fp) vp.push_back(o + (int32_t)vls.size());
  vls.insert(vls.end(), fls.begin(), fls.end());
  for (size_t j = 0; j < cls.size(); ++j) {
    const int32_t d = cls[j];
    const int r = d & 0xFF, q = (d >> 8) & 7, f = fused_col[r] >= 0, pr = (d >> 24) & 1, g = r >= lds_rows;
    cl2.push_back(row_off[r]);
    cl2.push_back(r | (q << 8) | ((f ? fused_col[r] : 0) << 11) | ((int)by_row[r].size() << 19) | (f << 24) | (pr << 25) |
                  (g << 26) | (cprio[j] << 27));
  }
  for (size_t j = 0; j < vls.size(); ++j) {
    const int32_t d = vls[j];
    const int c = d & 0xFF, pr = (d >> 24) & 1;
    vl2.push_back((d & 0xFFFF) | ((int)cl[c].size() << 16) | (pr << 21) | ((int)cg[c].size() << 22) |
                  ((j < vprio.size() ? vprio[j] : 0) << 27));
    vl2.push_back(col_start[c]);
  }
  cl2.resize(cl2.size() + 2, 0); vl2.resize(vl2.size() + 2, 0);
  // measured (tools/sweep_ldpc.py): with up to about a quarter of the edges in L2 this engine beats the compressed
  // state engine (+16 % at 4 %, +9 % at 26 %, even at 28 %); beyond that the L2 round trips of the VN phase dominate
  h->sp_spill_pct = (eg * 100 + el + eg - 1) / (el + eg);     // min-sum uses this engine up to 27 %, see ldpc5g.hip
  h->sp_lds_bytes = el * z * 4;
  h->sp_g_floats = eg * z;
  int rc = upload(&h->sp_col_ent, col_ent.data(), col_ent.size());
  if (rc == SAMD_OK) rc = upload(&h->sp_cn_ptr, cp.data(), cp.size());
  if (rc == SAMD_OK) rc = upload(&h->sp_vn_ptr, vp.data(), vp.size());
  if (rc == SAMD_OK) rc = upload(&h->sp_cn_list, cl2.data(), cl2.size());
  if (rc == SAMD_OK) rc = upload(&h->sp_vn_list, vl2.data(), vl2.size());
  if (rc == SAMD_OK) h->sp_ok = 1;
  return rc;
}

void free_onchip_mss_tables(samd_ldpc5g* h) {
  (void)hipFree(h->sp_col_ent); (void)hipFree(h->sp_cn_ptr); (void)hipFree(h->sp_vn_ptr);
  (void)hipFree(h->sp_cn_list); (void)hipFree(h->sp_vn_list);
}

static int mss_grid(const samd_ldpc5g* h, int batch) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return std::min(batch, cus);
}

size_t onchip_mss_workspace_bytes(const samd_ldpc5g* h, int batch) {
  if (!h->sp_ok || batch <= 0) return 0;
  const int nbu = (h->n_vn + h->z - 1) / h->z;
  return (size_t)mss_grid(h, batch) * ((size_t)nbu * h->z + h->sp_g_floats) * sizeof(float) + 256;
}

int launch_onchip_mss(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                      float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                      size_t workspace_bytes, hipStream_t st) {
  if (!h->sp_ok) {
    set_error("no spill schedule for this code");
    return SAMD_ERR_UNSUPPORTED;
  }
  if (!workspace || workspace_bytes < onchip_mss_workspace_bytes(h, batch)) {
    set_error("workspace too small (samd_ldpc5g_decode_workspace_bytes)");
    return SAMD_ERR_WORKSPACE;
  }
  float* ws = reinterpret_cast<float*>(align_up((size_t)workspace, 256));
  const bool pow2 = (h->z & (h->z - 1)) == 0;
  typedef void (*kern_t)(const float*, float*, float*, RateMatch, int, int, int, int, float, float, int, int, int,
                         const int32_t*, const int32_t*, const int2*, const int32_t*, const int2*);
  static const kern_t kerns[8] = {ldpc5g_decode_mss_kernel<false, SAMD_CN_MINSUM>, ldpc5g_decode_mss_kernel<true, SAMD_CN_MINSUM>,
                                  ldpc5g_decode_mss_kernel<false, SAMD_CN_BOXPLUS_PHI>, ldpc5g_decode_mss_kernel<true, SAMD_CN_BOXPLUS_PHI>,
                                  ldpc5g_decode_mss_kernel<false, SAMD_CN_BOXPLUS>, ldpc5g_decode_mss_kernel<true, SAMD_CN_BOXPLUS>,
                                  ldpc5g_decode_mss_kernel<false, SAMD_CN_BOXPLUS_PHI_FAST>, ldpc5g_decode_mss_kernel<true, SAMD_CN_BOXPLUS_PHI_FAST>};
  const int mi = cn_mode == SAMD_CN_BOXPLUS_PHI ? 1 : cn_mode == SAMD_CN_BOXPLUS ? 2 : cn_mode == SAMD_CN_BOXPLUS_PHI_FAST ? 3 : 0;
  const kern_t fn = kerns[2 * mi + (pow2 ? 1 : 0)];
  SAMD_SET_MAX_LDS(fn, 160 * 1024);
  const int nbu = (h->n_vn + h->z - 1) / h->z;
  const RateMatch rm = make_rate_match(h);
  const float off = (cn_mode == SAMD_CN_OFFSET_MINSUM) ? offset : 0.f;
  hipLaunchKernelGGL(fn, dim3(mss_grid(h, batch)), dim3(1024), h->sp_lds_bytes, st, llr, out, ws, rm, h->n_cn, nbu, batch,
                     num_iter, llr_max, off, hard_out, return_infobits, h->sp_g_floats, h->sp_col_ent, h->sp_cn_ptr,
                     reinterpret_cast<const int2*>(h->sp_cn_list), h->sp_vn_ptr,
                     reinterpret_cast<const int2*>(h->sp_vn_list));
  return launch_status();
}

}  // namespace samd
