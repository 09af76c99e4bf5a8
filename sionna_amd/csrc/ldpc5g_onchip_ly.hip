// On-chip LAYERED decoder for 5G-NR LDPC codes (round 3).
//
// Replaces LDPC5GDecoder(cn_schedule="layered").call (reference src/sionna/phy/fec/ldpc/decoding.py:1383-1389: one
// sub-iteration per base row = Z check nodes; _bp_iter with an array schedule :463-520: check-node update of the layer,
// then the variable-node update) for codes like those of the explicit-message engine's grouped kernel (Z a multiple of 64, no
// partially pruned base row, messages in LDS).  Until round 3 the schedule ran on the HBM-resident engine with two
// launches per layer: 920 launches and 79 k decodes/s for 10 iterations at config C2.
//
// State of one codeword (one workgroup, 16 waves):
//   LDS   c2v[e][z]   one float per edge of the NON-FUSED columns, block e = (row, position), indexed by the check node's
//                     lifted copy - the message a check node SENT last; it stays until the row's next update
//         xtot[c][z]  the unclipped total of every variable node of a non-fused column (sum of its c2v in ascending
//                     check-node order, channel LLR last - vn_update_sum, decoding.py:681-732)
//   L2 workspace row  channel LLRs of all columns, and cext[r][z]: the c2v of a row's fused edge (the degree-1 column of
//                     the base graph's extension part, private to its row)
// v2c messages are never stored: v2c_e = clip(xtot[v] - c2v_e) is what the reference's variable-node update leaves on an
// edge (decoding.py:724-731), recomputed when the row is updated.  After a layer, the reference updates EVERY variable
// node; only the nodes of the layer's columns see a changed input, so re-summing those - all their edges, in the defined
// order - gives the same bits (min-sum: bit-identical to oracle/ldpc_bp.py's literal form; boxplus rules: same function).
// Consecutive base rows that share no column are one group (BG1: 32 groups instead of 46 layers): their updates commute.
//
// Every wave walks a linear record list built on the host: CN items (row, both 64-lane chunks), a workgroup barrier, VN
// re-sum items (column; pairs of chunks for degree <= 12), a barrier, next group.  The operands a record needs from the
// L2 workspace (fused state, channel LLRs) are requested while the previous record runs.
#include <array>
#include "ldpc5g_onchip_ms.inc"

namespace samd {

enum { LY_END = 0, LY_BARRIER = 1, LY_CN = 2, LY_VN = 3 };

// check nodes (row r, lifted copies lane and lane + 64): D edges, the last one fused when F.
// ent: (byte offset of the column's xtot block, 4 shift) per non-fused edge; a0 = row block byte offset + 4 lane
// NCH = 2: both 64-lane chunks in one item (two dependency chains per wave); NCH = 1: one chunk - the row's two chunks
// then run on two waves side by side (the check-node update of a layer is the serial part of a group)
template <int D, bool F, bool POW2, int MODE, int NCH>
__device__ __forceinline__ void ly_cn_row(unsigned a0, unsigned z4, const int32_t* __restrict__ ent, unsigned lane4,
                                          unsigned zwv, float llr_max, float offset, float* __restrict__ cext,
                                          float co0, float co1, float lf0, float lf1) {
  constexpr int NF = F ? D - 1 : D;
  float v[NCH][D];
  float co[2] = {co0, co1};
#pragma unroll
  for (int i = 0; i < NF; ++i) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const unsigned t = lane4 + 256u * h + (unsigned)ent[2 * i + 1];
      const unsigned ax = POW2 ? ((t & zwv) | (unsigned)ent[2 * i]) : (min(t, t - zwv) + (unsigned)ent[2 * i]);
      const float x = lds_ld(ax);
      const float c = lds_ld(a0 + (unsigned)i * z4 + 256u * h);
      v[h][i] = ms_med3(x - c, -llr_max, llr_max);                     // the v2c the last variable-node update left
    }
  }
  if constexpr (F) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const float x = co[h] + (h ? lf1 : lf0);                          // (0 + c2v) + llr of the degree-1 node
      v[h][D - 1] = ms_med3(x - co[h], -llr_max, llr_max);
    }
  }
  if constexpr (MODE == SAMD_CN_MINSUM) {
    ms_minsum_inplace<D, NCH, 1>(v, llr_max, offset);
  } else {
    // boxplus rules: bp_math.h's node update on the D messages of a chunk - the function of every other engine
#pragma unroll
    for (int h = 0; h < NCH; ++h) cn_update_col<MODE, D>(v[h], D, llr_max, 0.f);
  }
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) lds_st(a0 + (unsigned)i * z4 + 256u * h, v[h][i]);
  if constexpr (F) {
    cext[0] = v[0][D - 1];
    if constexpr (NCH == 2) cext[64] = v[1][D - 1];
  }
}

// variable nodes of column c (lifted copies of chunk(s)): re-sum of all D messages + channel LLR -> xtot.
// ent: (edge block byte offset, 4 shift) per edge, rows ascending; ax = byte address of xtot[c][64 chunk + lane]
template <int D, int NCH, bool POW2>
__device__ __forceinline__ void ly_vn_col(const int32_t* __restrict__ ent, unsigned zz4, unsigned zwv, unsigned ax,
                                          float l0, float l1) {
  float c[NCH][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const unsigned t = zz4 + 256u * h - (unsigned)ent[2 * i + 1];
      const unsigned a = POW2 ? ((t & zwv) | (unsigned)ent[2 * i]) : (min(t, t + zwv) + (unsigned)ent[2 * i]);
      c[h][i] = lds_ld(a);
    }
  if constexpr (NCH == 2) {
    ms_f32x2 xv = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < D; ++i) xv += ms_f32x2{c[0][i], c[1][i]};
    xv += ms_f32x2{l0, l1};
    lds_st(ax, xv.x);
    lds_st(ax + 256u, xv.y);
  } else {
    float x = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) x += c[0][i];
    x += l0;
    lds_st(ax, x);
  }
}

template <bool POW2, int MODE>
__global__ __launch_bounds__(1024) void ldpc5g_decode_ly_kernel(
    const float* __restrict__ llr_in, float* __restrict__ out, float* __restrict__ ws, RateMatch p, int nbu, int batch,
    int num_iter, float llr_max, float offset, int hard_out, int return_infobits, int msg_floats, int n_ext,
    const int32_t* __restrict__ rec_ptr, const int4* __restrict__ recs, const int32_t* __restrict__ row_ent,
    const int32_t* __restrict__ col_ent, const int32_t* __restrict__ xt_index) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((unsigned)(size_t)(lds_f32*)smem != 0u) __builtin_trap();      // LDS addressed by plain byte offsets (lds_ld)
  constexpr int NT = 1024;
  const unsigned z = (unsigned)p.z, z4 = 4u * z;
  const unsigned zw = POW2 ? z4 - 1u : z4;
  unsigned zwv;
  asm volatile("v_mov_b32 %0, %1" : "=v"(zwv) : "s"(zw));
  const int nx = nbu * (int)z;
  float* llr = ws + (size_t)blockIdx.x * (size_t)(nx + n_ext * (int)z);
  float* cext = llr + nx;
  float* xtot = smem + msg_floats;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned lane4 = 4u * (unsigned)lane;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = rec_ptr[w];

  for (int b = blockIdx.x; b < batch; b += gridDim.x) {
    const float* row = llr_in + (size_t)b * p.n;
    // decoding.py:552-565 (clip, logits -> LLR; + 0.f: no -0 anywhere), messages start at 0 (:575-578)
    for (int v = tid; v < nx; v += NT) {
      const float l = (v < p.n_vn) ? (-1.f * clampf(recover_llr(p, row, v, llr_max), -llr_max, llr_max)) + 0.f : 0.f;
      llr[v] = l;
      const int xi = xt_index[v / (int)z];
      if (xi >= 0) xtot[xi * (int)z + v % (int)z] = l;                 // total of a node without messages = its LLR
    }
    for (int i = tid; i < msg_floats; i += NT) smem[i] = 0.f;
    for (int i = tid; i < n_ext * (int)z; i += NT) cext[i] = 0.f;
    __syncthreads();

    // Records are wave-uniform: scalar loads, one record ahead.  (Vector loads three ahead were tried: they share the
    // in-order vmcnt counter with the fused-column stores, so waiting for a record meant waiting for those stores to
    // reach L2 - 47.1 instead of 44.4 ms per 16384 decodes, profiles/r03b/layered_abl_r03t.txt.)
    for (int it = 0; it < num_iter; ++it) {
      int t = r0;
      int4 cur = recs[t];
      float pa0 = 0.f, pa1 = 0.f, pb0 = 0.f, pb1 = 0.f;                // prefetched operands of `cur`
      auto fetch = [&](int4 rc, float& a0, float& a1, float& b0, float& b1) __attribute__((always_inline)) {
        const int kind = __builtin_amdgcn_readfirstlane(rc.x) & 0xFF;
        const int wv = __builtin_amdgcn_readfirstlane(rc.w);
        if (kind == LY_CN) {
          const int rx = __builtin_amdgcn_readfirstlane(rc.x);
          if ((rx >> 13) & 1) {                                         // fused: c2v of the fused edge, its channel LLR
            const int q64 = ((rx >> 16) & 0xFF) * 64;
            const float* ce = cext + (wv >> 16) * (int)z + q64 + lane;
            const float* le = llr + (wv & 0xFFFF) * (int)z + q64 + lane;
            a0 = ce[0]; b0 = le[0];
            if ((rx >> 14) & 1) { a1 = ce[64]; b1 = le[64]; }
          }
        } else if (kind == LY_VN) {
          a0 = llr[wv + lane];
          if ((__builtin_amdgcn_readfirstlane(rc.x) >> 13) & 1) a1 = llr[wv + lane + 64];
        }
      };
      fetch(cur, pa0, pa1, pb0, pb1);
      for (;;) {
        const int cx = __builtin_amdgcn_readfirstlane(cur.x);
        const int kind = cx & 0xFF;
        if (kind == LY_END) break;
        const int4 nxt = recs[t + 1];
        float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
        fetch(nxt, na0, na1, nb0, nb1);
        const int cy = __builtin_amdgcn_readfirstlane(cur.y), cz = __builtin_amdgcn_readfirstlane(cur.z);
        const int cw = __builtin_amdgcn_readfirstlane(cur.w);
        if (kind == LY_BARRIER) {
          // LDS traffic only: wait for this wave's LDS operations, not for its global ones (__syncthreads would also wait
          // for the next record's operands, just requested from L2, and for the fused-column stores - 64 exposed L2
          // round trips per iteration).  cy = number of barriers in a row (a wave without items in a group)
          for (int q = 0; q < cy; ++q) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (kind == LY_CN) {
          const unsigned zq4 = ((unsigned)(cx >> 16) & 0xFFu) * 256u + lane4;      // 4 (64 chunk + lane)
          float* ce = cext + (cw >> 16) * (int)z + (int)(zq4 >> 2);
#define SAMD_LY_CN(KEY, D, F)                                                                                          \
  case KEY: ly_cn_row<D, F, POW2, MODE, 2>((unsigned)cy + zq4, z4, row_ent + cz, zq4, zwv, llr_max, offset, ce, pa0, pa1, pb0, pb1); break; \
  case 64 + KEY: ly_cn_row<D, F, POW2, MODE, 1>((unsigned)cy + zq4, z4, row_ent + cz, zq4, zwv, llr_max, offset, ce, pa0, pa1, pb0, pb1); break;
          switch (((cx >> 8) & 63) | (((cx >> 14) & 1) ? 0 : 64)) {
            SAMD_LY_CN(3, 3, false) SAMD_LY_CN(4, 4, false) SAMD_LY_CN(5, 5, false) SAMD_LY_CN(6, 6, false)
            SAMD_LY_CN(7, 7, false) SAMD_LY_CN(8, 8, false) SAMD_LY_CN(9, 9, false) SAMD_LY_CN(10, 10, false)
            SAMD_LY_CN(19, 19, false)
            SAMD_LY_CN(35, 3, true) SAMD_LY_CN(36, 4, true) SAMD_LY_CN(37, 5, true) SAMD_LY_CN(38, 6, true)
            SAMD_LY_CN(39, 7, true) SAMD_LY_CN(40, 8, true) SAMD_LY_CN(41, 9, true) SAMD_LY_CN(42, 10, true)
            default: break;
          }
#undef SAMD_LY_CN
        } else {
          const unsigned chunk = (unsigned)(cx >> 16) & 0xFFu;
          const unsigned zz4 = chunk * 256u + lane4;
          const unsigned ax = (unsigned)cy + zz4;
#define SAMD_LY_VN(KEY, D, NCHV) case KEY: ly_vn_col<D, NCHV, POW2>(col_ent + cz, zz4, zwv, ax, pa0, pa1); break;
          switch ((cx >> 8) & 63) {
            SAMD_LY_VN(1, 1, 1) SAMD_LY_VN(2, 2, 1) SAMD_LY_VN(3, 3, 1) SAMD_LY_VN(4, 4, 1) SAMD_LY_VN(5, 5, 1)
            SAMD_LY_VN(6, 6, 1) SAMD_LY_VN(7, 7, 1) SAMD_LY_VN(8, 8, 1) SAMD_LY_VN(9, 9, 1) SAMD_LY_VN(10, 10, 1)
            SAMD_LY_VN(11, 11, 1) SAMD_LY_VN(12, 12, 1) SAMD_LY_VN(13, 13, 1) SAMD_LY_VN(14, 14, 1)
            SAMD_LY_VN(15, 15, 1) SAMD_LY_VN(16, 16, 1) SAMD_LY_VN(17, 17, 1) SAMD_LY_VN(18, 18, 1)
            SAMD_LY_VN(19, 19, 1) SAMD_LY_VN(20, 20, 1) SAMD_LY_VN(21, 21, 1) SAMD_LY_VN(22, 22, 1)
            SAMD_LY_VN(23, 23, 1) SAMD_LY_VN(24, 24, 1) SAMD_LY_VN(25, 25, 1) SAMD_LY_VN(26, 26, 1)
            SAMD_LY_VN(27, 27, 1) SAMD_LY_VN(28, 28, 1) SAMD_LY_VN(29, 29, 1) SAMD_LY_VN(30, 30, 1)
            SAMD_LY_VN(33, 1, 2) SAMD_LY_VN(34, 2, 2) SAMD_LY_VN(35, 3, 2) SAMD_LY_VN(36, 4, 2) SAMD_LY_VN(37, 5, 2)
            SAMD_LY_VN(38, 6, 2) SAMD_LY_VN(39, 7, 2) SAMD_LY_VN(40, 8, 2) SAMD_LY_VN(41, 9, 2) SAMD_LY_VN(42, 10, 2)
            SAMD_LY_VN(43, 11, 2) SAMD_LY_VN(44, 12, 2)
            default: break;
          }
#undef SAMD_LY_VN
        }
        ++t;
        cur = nxt; pa0 = na0; pa1 = na1; pb0 = nb0; pb1 = nb1;
      }
    }
    __syncthreads();
    // ---------------- output (decoding.py:620-626, 1486-1531): marginal of a non-fused column = xtot, of a fused
    // (degree-1) node = c2v + llr
    auto marginal = [&](int v) __attribute__((always_inline)) -> float {
      const int c = v / (int)z, zv = v - c * (int)z;
      const int xi = xt_index[c];
      if (xi >= 0) return xtot[xi * (int)z + zv];
      if (xi == -0x7FFFFFFF) return llr[v];                              // column without edges in the pruned graph
      return cext[(-1 - xi) * (int)z + zv] + llr[v];
    };
    if (return_infobits) {
      float* o = out + (size_t)b * p.k;
      for (int v = tid; v < p.k; v += NT) {
        const float x = clampf(marginal(v), -llr_max, llr_max);
        o[v] = hard_out ? ((0.f >= x) ? 1.f : 0.f) : -1.f * x;
      }
    } else {
      float* o = out + (size_t)b * p.n;
      for (int i = tid; i < p.n; i += NT) {
        const float x = clampf(marginal(short_to_full(p, out_to_short(p, i))), -llr_max, llr_max);
        o[i] = hard_out ? ((0.f >= x) ? 1.f : 0.f) : -1.f * x;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ host tables
int build_onchip_ly_tables(samd_ldpc5g* h, const std::vector<std::vector<std::pair<int, int>>>& by_row) {
  h->ly_ok = 0;
  const int z = h->z, NW = 16;
  const int ncu = (h->n_cn + z - 1) / z, nbu = (h->n_vn + z - 1) / z;
  if (z % 64 != 0 || h->n_cn % z != 0 || h->n_vn % z != 0 || h->mb > 255 || h->nb > 255 || nbu > 0xFFFF) return SAMD_OK;
  static const int degs[] = {3, 4, 5, 6, 7, 8, 9, 10, 19};
  std::vector<int> col_deg(h->nb, 0);
  for (int r = 0; r < ncu; ++r)
    for (auto& e : by_row[r]) {
      if (e.first >= nbu) return SAMD_OK;
      ++col_deg[e.first];
    }
  // fused edges: the row's last edge, shift 0, to a degree-1 column (extension part), row degree 3..10
  std::vector<int> fused_col(ncu, -1), ext_of_row(ncu, -1);
  std::vector<char> col_fused(h->nb, 0);
  int n_ext = 0;
  for (int r = 0; r < ncu; ++r) {
    const int d = (int)by_row[r].size();
    if (std::find(std::begin(degs), std::end(degs), d) == std::end(degs)) return SAMD_OK;
    const int c = by_row[r][d - 1].first, sft = by_row[r][d - 1].second;
    if (col_deg[c] == 1 && sft == 0 && d >= 3 && d <= 10) { fused_col[r] = c; col_fused[c] = 1; ext_of_row[r] = n_ext++; }
  }
  // LDS layout: edge blocks of the non-fused edges, row-major; then xtot of the non-fused columns that have edges
  std::vector<int> row_off(ncu, 0), xt_of_col(h->nb, -0x7FFFFFFF);
  int edges = 0, ncore = 0;
  for (int r = 0; r < ncu; ++r) { row_off[r] = edges * z * 4; edges += (int)by_row[r].size() - (fused_col[r] >= 0 ? 1 : 0); }
  for (int c = 0; c < nbu; ++c) {
    if (col_fused[c]) continue;
    if (col_deg[c] == 0) continue;
    if (col_deg[c] > 30) return SAMD_OK;
    xt_of_col[c] = ncore++;
  }
  const size_t lds = ((size_t)edges + (size_t)ncore) * z * 4;
  if (lds > 160 * 1024) return SAMD_OK;
  const int xt_base = edges * z * 4;
  std::vector<int32_t> xt_index(h->nb, -0x7FFFFFFF);
  for (int c = 0; c < nbu; ++c) xt_index[c] = xt_of_col[c];
  for (int r = 0; r < ncu; ++r)
    if (fused_col[r] >= 0) xt_index[fused_col[r]] = -1 - ext_of_row[r];
  // per-row tables (xtot byte offset of the column, 4 shift) and per-column tables (edge block byte offset, 4 shift)
  std::vector<int32_t> row_ent, row_start(ncu, 0), col_ent, col_start(h->nb, 0);
  std::vector<std::vector<std::pair<int, int>>> col_edges(h->nb);       // (block byte offset, 4 shift), rows ascending
  for (int r = 0; r < ncu; ++r) {
    row_start[r] = (int32_t)row_ent.size();
    const int nf = (int)by_row[r].size() - (fused_col[r] >= 0 ? 1 : 0);
    for (int i = 0; i < nf; ++i) {
      const int c = by_row[r][i].first, s = by_row[r][i].second;
      if (xt_of_col[c] < 0) return SAMD_OK;
      row_ent.push_back(xt_base + xt_of_col[c] * z * 4);
      row_ent.push_back(4 * s);
      col_edges[c].push_back({row_off[r] + i * z * 4, 4 * s});
    }
  }
  row_ent.resize(row_ent.size() + 64, 0);
  for (int c = 0; c < nbu; ++c) {
    col_start[c] = (int32_t)col_ent.size();
    for (auto& e : col_edges[c]) { col_ent.push_back(e.first); col_ent.push_back(e.second); }
  }
  col_ent.resize(col_ent.size() + 64, 0);
  // groups of consecutive rows that share no non-fused column
  std::vector<std::vector<int>> groups;
  {
    std::vector<char> used(h->nb, 0);
    std::vector<int> cur;
    for (int r = 0; r < ncu; ++r) {
      bool clash = false;
      for (auto& e : by_row[r]) clash = clash || (!col_fused[e.first] && used[e.first]);
      if ((clash && !cur.empty()) || getenv("SAMD_LY_NOGROUP")) {
        if (!cur.empty()) groups.push_back(cur);
        cur.clear();
        std::fill(used.begin(), used.end(), 0);
      }
      cur.push_back(r);
      for (auto& e : by_row[r]) used[e.first] = 1;
    }
    if (!cur.empty()) groups.push_back(cur);
  }
  // per-wave record lists: CN items of the group, barrier, VN items, barrier
  std::vector<std::vector<std::array<int32_t, 4>>> per(NW);
  int rot = 0;
  for (auto& g : groups) {
    // CN items: one per row, on consecutive waves starting at a rotating position (so that the serial part does not
    // always load the same SIMD)
    // (SAMD_LY_CN_SPLIT=0: one item for both chunks; default: the chunks of a row as separate items on different waves)
    const bool cn_split = !(getenv("SAMD_LY_CN_SPLIT") && atoi(getenv("SAMD_LY_CN_SPLIT")) == 0);
    int slot = 0;
    for (size_t j = 0; j < g.size(); ++j) {
      const int r = g[j], d = (int)by_row[r].size(), f = fused_col[r] >= 0;
      const int32_t wcol = f ? (fused_col[r] | (ext_of_row[r] << 16)) : 0;
      for (int q = 0; q < z / 64; ++q) {
        if (cn_split || q + 1 >= z / 64) {                    // (an odd last chunk is a single-chunk item in any case)
          per[(rot + slot++) % NW].push_back({LY_CN | ((d | (f << 5)) << 8) | (q << 16), row_off[r], row_start[r], wcol});
        } else {
          per[(rot + slot++) % NW].push_back({LY_CN | ((d | (f << 5)) << 8) | (1 << 14) | (q << 16), row_off[r], row_start[r], wcol});
          ++q;
        }
      }
    }
    for (int wv = 0; wv < NW; ++wv) per[wv].push_back({LY_BARRIER, 0, 0, 0});
    // VN items of the columns the group touched, longest first onto the least loaded wave
    std::vector<std::pair<int, std::array<int32_t, 4>>> items;
    std::vector<char> seen(h->nb, 0);
    int ncols = 0;
    for (int r : g)
      for (auto& e : by_row[r])
        if (!col_fused[e.first] && !seen[e.first]) { seen[e.first] = 1; ++ncols; }
    std::fill(seen.begin(), seen.end(), 0);
    // few columns: single-chunk items spread over more waves (an item's time is mostly its fixed latency)
    const int single_max = getenv("SAMD_LY_VN_SINGLE_MAX") ? atoi(getenv("SAMD_LY_VN_SINGLE_MAX")) : 0;   // measured at C2: 0 -> 369 k, 10 -> 362 k, 32 -> 354 k decodes/s
    const bool all_single = ncols * (z / 64) <= 2 * single_max;
    for (int r : g)
      for (auto& e : by_row[r]) {
        const int c = e.first;
        if (col_fused[c] || seen[c]) continue;
        seen[c] = 1;
        const int dg = col_deg[c], chunks = z / 64;
        for (int q = 0; q < chunks; ++q) {
          const bool pair = dg <= 12 && !all_single && q + 1 < chunks;
          items.push_back({(pair ? 10 : 10) * dg + (pair ? 200 : 100),
                           {LY_VN | ((dg | ((pair ? 1 : 0) << 5)) << 8) | (q << 16), xt_base + xt_of_col[c] * z * 4, col_start[c], c * z + q * 64}});
          if (pair) ++q;
        }
      }
    std::stable_sort(items.begin(), items.end(), [](auto& a, auto& b) { return a.first > b.first; });
    std::vector<int> load(NW, 0);
    for (auto& it : items) {
      int wv = rot % NW;
      for (int q = 1; q < NW; ++q) {
        const int a = (rot + q) % NW;
        if (load[a] < load[wv]) wv = a;
      }
      per[wv].push_back(it.second);
      load[wv] += it.first;
    }
    for (int wv = 0; wv < NW; ++wv) per[wv].push_back({LY_BARRIER, 0, 0, 0});
    rot = (rot + 1) % NW;
  }
  // SAMD_LY_ABL (development, wrong results): 1 = lists without the VN re-sum items, 2 = without the CN items, 3 = barriers only
  const int abl = getenv("SAMD_LY_ABL") ? atoi(getenv("SAMD_LY_ABL")) : 0;
  std::vector<int32_t> rec_ptr, recs;
  for (int wv = 0; wv < NW; ++wv) {
    rec_ptr.push_back((int32_t)(recs.size() / 4));
    int last_barrier = -1;                                  // index (in recs) of a barrier record that can take one more
    for (auto& rc : per[wv]) {
      const int kind = rc[0] & 0xFF;
      if ((kind == LY_VN && (abl & 1)) || (kind == LY_CN && (abl & 2))) continue;
      if (kind == LY_BARRIER && last_barrier >= 0) { ++recs[last_barrier + 1]; continue; }   // barriers in a row: one record
      if (kind == LY_BARRIER) { last_barrier = (int)recs.size(); recs.insert(recs.end(), {LY_BARRIER, 1, 0, 0}); continue; }
      last_barrier = -1;
      recs.insert(recs.end(), rc.begin(), rc.end());
    }
    for (int q = 0; q < 4; ++q) recs.insert(recs.end(), {LY_END, 0, 0, 0});   // the walker reads three records ahead
  }
  rec_ptr.push_back((int32_t)(recs.size() / 4));
  h->ly_lds_bytes = (int)lds;
  h->ly_msg_floats = edges * z;
  h->ly_n_ext = n_ext;
  h->ly_groups = (int)groups.size();
  int rc = upload(&h->ly_rec_ptr, rec_ptr.data(), rec_ptr.size());
  if (rc == SAMD_OK) rc = upload(&h->ly_recs, recs.data(), recs.size());
  if (rc == SAMD_OK) rc = upload(&h->ly_row_ent, row_ent.data(), row_ent.size());
  if (rc == SAMD_OK) rc = upload(&h->ly_col_ent, col_ent.data(), col_ent.size());
  if (rc == SAMD_OK) rc = upload(&h->ly_xt_index, xt_index.data(), xt_index.size());
  if (rc == SAMD_OK) h->ly_ok = 1;
  return rc;
}

void free_onchip_ly_tables(samd_ldpc5g* h) {
  (void)hipFree(h->ly_rec_ptr); (void)hipFree(h->ly_recs); (void)hipFree(h->ly_row_ent); (void)hipFree(h->ly_col_ent);
  (void)hipFree(h->ly_xt_index);
}

static int ly_grid(const samd_ldpc5g* h, int batch) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int per_cu = std::max(1, (int)((160 * 1024) / std::max(1, h->ly_lds_bytes)));
  int grid = std::min(batch, cus * std::min(per_cu, 1));
  if (const char* e = getenv("SAMD_ONCHIP_GRID")) grid = std::min(grid, std::max(1, atoi(e)));
  return grid;
}

size_t onchip_ly_workspace_bytes(const samd_ldpc5g* h, int batch) {
  if (!h->ly_ok || batch <= 0) return 0;
  const int nbu = (h->n_vn + h->z - 1) / h->z;
  return (size_t)ly_grid(h, batch) * (size_t)(nbu + h->ly_n_ext) * h->z * sizeof(float) + 256;
}

int launch_onchip_ly(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode, float llr_max,
                     float offset, int hard_out, int return_infobits, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const bool minsum = cn_mode == SAMD_CN_MINSUM || cn_mode == SAMD_CN_OFFSET_MINSUM;
  const bool phi = cn_mode == SAMD_CN_BOXPLUS_PHI || cn_mode == SAMD_CN_BOXPLUS_PHI_FAST;
  if (!h->ly_ok || !(minsum || phi)) {
    set_error("layered on-chip engine: code or rule not covered (the HBM-resident scheduled engine takes it)");
    return SAMD_ERR_UNSUPPORTED;
  }
  if (minsum && (h->max_dc > 27 || !(llr_max >= 0.f) || (double)h->max_dc * 2.0 * (double)llr_max >= 99999.0)) {
    set_error("code / llr_max outside the on-chip decoder's envelope");
    return SAMD_ERR_UNSUPPORTED;
  }
  if (!workspace || workspace_bytes < onchip_ly_workspace_bytes(h, batch)) {
    set_error("workspace too small (samd_ldpc5g_decode_layered_workspace_bytes)");
    return SAMD_ERR_WORKSPACE;
  }
  float* ws = reinterpret_cast<float*>(align_up((size_t)workspace, 256));
  const bool pow2 = (h->z & (h->z - 1)) == 0;
  typedef void (*kern_t)(const float*, float*, float*, RateMatch, int, int, int, float, float, int, int, int, int,
                         const int32_t*, const int4*, const int32_t*, const int32_t*, const int32_t*);
  static const kern_t kerns[6] = {ldpc5g_decode_ly_kernel<false, SAMD_CN_MINSUM>, ldpc5g_decode_ly_kernel<true, SAMD_CN_MINSUM>,
                                  ldpc5g_decode_ly_kernel<false, SAMD_CN_BOXPLUS_PHI>, ldpc5g_decode_ly_kernel<true, SAMD_CN_BOXPLUS_PHI>,
                                  ldpc5g_decode_ly_kernel<false, SAMD_CN_BOXPLUS_PHI_FAST>, ldpc5g_decode_ly_kernel<true, SAMD_CN_BOXPLUS_PHI_FAST>};
  const kern_t fn = kerns[(minsum ? 0 : cn_mode == SAMD_CN_BOXPLUS_PHI ? 2 : 4) + (pow2 ? 1 : 0)];
  SAMD_HIP_CHECK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int nbu = (h->n_vn + h->z - 1) / h->z;
  const RateMatch rm{h->k, h->n, h->z, h->k_ldpc, h->n_vn, h->m_int};
  const float off = (cn_mode == SAMD_CN_OFFSET_MINSUM) ? offset : 0.f;
  hipLaunchKernelGGL(fn, dim3(ly_grid(h, batch)), dim3(1024), (size_t)h->ly_lds_bytes, st, llr, out, ws, rm, nbu, batch, num_iter,
                     llr_max, off, hard_out, return_infobits, h->ly_msg_floats, h->ly_n_ext, h->ly_rec_ptr,
                     reinterpret_cast<const int4*>(h->ly_recs), h->ly_row_ent, h->ly_col_ent, h->ly_xt_index);
  return launch_status();
}

}  // namespace samd
