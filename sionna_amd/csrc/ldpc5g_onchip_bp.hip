// On-chip flooding decoder for 5G-NR LDPC codes with the boxplus-phi (reference default) and
// boxplus (tanh) check-node rules: one float per edge, resident in LDS for all iterations.
//
// Replaces LDPC5GDecoder.call = rate recovery + LDPCBPDecoder._bp_iter x num_iter + cn_update_phi /
// cn_update_tanh + vn_update_sum + output mapping (reference src/sionna/phy/fec/ldpc/decoding.py:
// 1427-1536, 416-524, 681-732, 955-1166) for codes whose E 4-byte messages fit in 160 KB - C2
// (n=8448, rate 1/3, Z=128: 316 Z edges = 158 KB, channel LLRs in an L2 workspace row) and everything
// smaller.  The HBM-resident engine (ldpc_bp_generic.hip) moves 16 E + 4 N bytes per codeword and
// iteration through HBM and sits at ~41 % of the HBM roofline; here the messages never leave the CU.
//
// Layout: edge block e = (row r, position i in the row) holds msg[e][zr], zr = lifted copy of the
// CHECK node - the CN phase reads/writes lane-contiguous rows, the VN phase of column c reads the block
// rotated by the edge's shift.  A message slot holds v2c after the VN phase and c2v after the CN phase
// (in place, like the HBM engine).  Arithmetic and summation order are those of ldpc_bp_generic.hip
// (bp_math.h: ascending VN inside a CN, ascending CN inside a VN, channel LLR last), so the two engines
// return the same bits.
//
// Work split: (row, 64-lane chunk) and (column, chunk) items are assigned to the waves on the host
// (longest-processing-time first); every item is an instantiation for the row's exact degree / the
// column's degree class, all table entries are wave-uniform scalars.
#include "ldpc5g.h"
#include "ldpc5g_jit.h"

#include <cmath>
#include "bp_math.h"

namespace samd {

template <bool POW2>
__device__ __forceinline__ unsigned bp_wrap_sub(unsigned zz4, unsigned s4, unsigned zw) {
  const unsigned t = zz4 - s4;
  return POW2 ? (t & zw) : min(t, t + zw);
}

// one check node per lane: row of exact degree D; its messages are D lane-contiguous blocks
template <int MODE, int D>
__device__ __forceinline__ void bp_cn_row(char* __restrict__ msg_b, unsigned a0, unsigned z4, float llr_max) {
  float v[D];
#pragma unroll
  for (int i = 0; i < D; ++i) v[i] = *reinterpret_cast<const float*>(msg_b + a0 + (unsigned)i * z4);
  cn_update_col<MODE, D>(v, D, llr_max, 0.f);
#pragma unroll
  for (int i = 0; i < D; ++i) *reinterpret_cast<float*>(msg_b + a0 + (unsigned)i * z4) = v[i];
}

// one variable node per lane: column of degree d <= DMAX.  ent[i] = block byte offset | (4 shift) << 18.
// INIT: v2c of iteration 0 = channel LLR (decoding.py:571).  LAST: the marginal replaces the channel LLR.
template <int DMAX, bool POW2, bool INIT>
__device__ __forceinline__ void bp_vn_col(const int32_t* __restrict__ ent, int d, unsigned zz4, unsigned zw,
                                          char* __restrict__ msg_b, float* __restrict__ llr_v, float llr_max,
                                          bool last) {
  unsigned a[DMAX];
  float c[DMAX];
  const float l = *llr_v;
  float x = 0.f;
#pragma unroll
  for (int i = 0; i < DMAX; ++i)
    if (i < d) {
      const unsigned e = (unsigned)ent[i];
      a[i] = (e & 0x3FFFFu) + bp_wrap_sub<POW2>(zz4, e >> 18, zw);
      if (INIT) {
        *reinterpret_cast<float*>(msg_b + a[i]) = l;
      } else {
        c[i] = *reinterpret_cast<const float*>(msg_b + a[i]);
        x += c[i];
      }
    }
  if (INIT) return;
  x += l;
#pragma unroll
  for (int i = 0; i < DMAX; ++i)
    if (i < d) *reinterpret_cast<float*>(msg_b + a[i]) = clampf(-1.f * c[i] + x, -llr_max, llr_max);
  if (last) *llr_v = x;
}

template <bool POW2, bool INIT>
__device__ __forceinline__ void bp_vn_item(const int32_t* __restrict__ ent, int d, unsigned zz4, unsigned zw,
                                           char* __restrict__ msg_b, float* __restrict__ llr_v, float llr_max,
                                           bool last) {
  if (d <= 1) bp_vn_col<1, POW2, INIT>(ent, d, zz4, zw, msg_b, llr_v, llr_max, last);
  else if (d <= 3) bp_vn_col<3, POW2, INIT>(ent, d, zz4, zw, msg_b, llr_v, llr_max, last);
  else if (d <= 6) bp_vn_col<6, POW2, INIT>(ent, d, zz4, zw, msg_b, llr_v, llr_max, last);
  else if (d <= 10) bp_vn_col<10, POW2, INIT>(ent, d, zz4, zw, msg_b, llr_v, llr_max, last);
  else if (d <= 16) bp_vn_col<16, POW2, INIT>(ent, d, zz4, zw, msg_b, llr_v, llr_max, last);
  else bp_vn_col<kColStride, POW2, INIT>(ent, d, zz4, zw, msg_b, llr_v, llr_max, last);
}

// NW waves per workgroup, one codeword per workgroup at a time.  LLRG: channel LLRs (and, after the last
// iteration, the marginals) in a workspace row (L2) instead of LDS.
template <int MODE, bool POW2, int NW, bool LLRG>
__global__ __launch_bounds__(NW * 64) void ldpc5g_decode_bp_kernel(
    const float* __restrict__ llr_in, float* __restrict__ out, float* __restrict__ llr_ws, RateMatch p, int n_cn,
    int nbu, int batch, int num_iter, float llr_max, int hard_out, int return_infobits, int msg_floats,
    const int32_t* __restrict__ row_off, const int32_t* __restrict__ col_ent,
    const int32_t* __restrict__ col_deg, const int32_t* __restrict__ cn_ptr, const int32_t* __restrict__ cn_list,
    const int32_t* __restrict__ vn_ptr, const int32_t* __restrict__ vn_list) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = NW * 64;
  const unsigned z = (unsigned)p.z, z4 = 4u * z;
  const unsigned zw = POW2 ? z4 - 1u : z4;
  const int n_vn = p.n_vn;
  const int nx = nbu * (int)z;
  float* llr = LLRG ? llr_ws + (size_t)blockIdx.x * nx : smem + msg_floats;
  char* msg_b = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = cn_ptr[w], c1 = cn_ptr[w + 1];
  const int v0 = vn_ptr[w], v1 = vn_ptr[w + 1];

  for (int b = blockIdx.x; b < batch; b += gridDim.x) {
    const float* row = llr_in + (size_t)b * p.n;
    // decoding.py:552-565: clip, then logits -> LLR
    for (int v = tid; v < nx; v += NT)
      llr[v] = (v < n_vn) ? -1.f * clampf(recover_llr(p, row, v, llr_max), -llr_max, llr_max) : 0.f;
    __syncthreads();
    for (int t = v0; t < v1; ++t) {                                // v2c of iteration 0
      const int desc = __builtin_amdgcn_readfirstlane(vn_list[t]);  // c | chunk<<8
      const int c = desc & 0xFF;
      const unsigned zz = (unsigned)(((desc >> 8) & 0xFF) * 64 + lane);
      const int vn = c * (int)z + (int)zz;
      if (zz < z && vn < n_vn)
        bp_vn_item<POW2, true>(col_ent + c * kColStride, __builtin_amdgcn_readfirstlane(col_deg[c]), 4u * zz, zw,
                               msg_b, llr + vn, llr_max, false);
    }
    __syncthreads();

    for (int it = 0; it < num_iter; ++it) {
      for (int t = c0; t < c1; ++t) {
        const int desc = __builtin_amdgcn_readfirstlane(cn_list[t]);  // r | chunk<<8 | priority<<24
        const int r = desc & 0xFF;
        onchip_setprio(desc >> 24);                                   // longest remaining work first (ldpc5g.h)
        const unsigned zz = (unsigned)(((desc >> 8) & 0xFF) * 64 + lane);
        if (zz < z && (unsigned)r * z + zz < (unsigned)n_cn) {
          const unsigned ro = (unsigned)__builtin_amdgcn_readfirstlane(row_off[r]);
          const unsigned a0 = (ro & 0x3FFFFu) + 4u * zz;
#define SAMD_BP_CN(D) case D: bp_cn_row<MODE, D>(msg_b, a0, z4, llr_max); break
          switch (ro >> 18) {
            SAMD_BP_CN(3); SAMD_BP_CN(4); SAMD_BP_CN(5); SAMD_BP_CN(6); SAMD_BP_CN(7); SAMD_BP_CN(8); SAMD_BP_CN(9);
            SAMD_BP_CN(10); SAMD_BP_CN(19);
            default: break;
          }
#undef SAMD_BP_CN
        } else if (zz < z) {
          // pruned check node of the last, partial base row (decoding.py:1330-1370 prunes by node count): its
          // edges do not exist - keep their slots at 0 so that the VN sums of the neighbouring columns skip them
          const unsigned ro = (unsigned)__builtin_amdgcn_readfirstlane(row_off[r]);
          for (unsigned i = 0; i < (ro >> 18); ++i)
            *reinterpret_cast<float*>(msg_b + (ro & 0x3FFFFu) + 4u * zz + i * z4) = 0.f;
        }
      }
      __syncthreads();
      const bool last = (it == num_iter - 1);
      for (int t = v0; t < v1; ++t) {
        const int desc = __builtin_amdgcn_readfirstlane(vn_list[t]);
        const int c = desc & 0xFF;
        onchip_setprio(desc >> 24);
        const unsigned zz = (unsigned)(((desc >> 8) & 0xFF) * 64 + lane);
        const int vn = c * (int)z + (int)zz;
        if (zz < z && vn < n_vn)
          bp_vn_item<POW2, false>(col_ent + c * kColStride, __builtin_amdgcn_readfirstlane(col_deg[c]), 4u * zz, zw,
                                  msg_b, llr + vn, llr_max, last);
      }
      __syncthreads();
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

static const int kBpCnDegrees[] = {3, 4, 5, 6, 7, 8, 9, 10, 19};

int build_onchip_bp_tables(samd_ldpc5g* h, const std::vector<std::vector<std::pair<int, int>>>& by_row) {
  const int z = h->z;
  const int ncu = (h->n_cn + z - 1) / z, nbu = (h->n_vn + z - 1) / z;
  h->bp_ok = 0;
  if (h->mb > 255 || h->nb > 255 || (z + 63) / 64 > 255) return SAMD_OK;
  std::vector<int32_t> row_off(h->mb, 0), col_ent((size_t)h->nb * kColStride, 0), col_deg(h->nb, 0);

  int edges = 0;
  for (int r = 0; r < ncu; ++r) {
    const int d = (int)by_row[r].size();
    if (std::find(std::begin(kBpCnDegrees), std::end(kBpCnDegrees), d) == std::end(kBpCnDegrees)) return SAMD_OK;
    row_off[r] = (edges * z * 4) | (d << 18);               // byte offset of the first edge block | degree << 18
    for (int i = 0; i < d; ++i) {
      const int c = by_row[r][i].first, s = by_row[r][i].second;
      if (c >= nbu || col_deg[c] >= kColStride) return SAMD_OK;
      col_ent[(size_t)c * kColStride + col_deg[c]++] = ((edges + i) * z * 4) | ((s * 4) << 18);   // rows ascending
    }
    edges += d;
  }
  const size_t msg_bytes = (size_t)edges * z * 4;
  if (msg_bytes > 160 * 1024 || msg_bytes >= (1u << 18)) return SAMD_OK;   // the HBM-resident engine takes it
  h->bp_edges = edges;
  size_t lds = msg_bytes + (size_t)nbu * z * 4;
  h->bp_llr_global = 0;
  if (lds > 160 * 1024) { h->bp_llr_global = 1; lds = msg_bytes; }
  h->bp_waves = 16;
  if (!h->bp_llr_global)
    for (int nwc : {8, 4, 2, 1})
      if (lds * (size_t)(kDecWaves / nwc) <= 160 * 1024) h->bp_waves = nwc;
  if (opt_set("SAMD_ONCHIP_BP_WAVES")) {
    const std::string e_s = opt_str("SAMD_ONCHIP_BP_WAVES");
    const char* e = e_s.c_str();
    const int v = atoi(e);
    if (!h->bp_llr_global && (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) && lds * (size_t)(kDecWaves / v) <= 160 * 1024)
      h->bp_waves = v;
  }
  // items: CN cost ~ degree (two phi evaluations per edge), VN cost ~ degree
  const int chunks = (z + 63) / 64;
  std::vector<std::pair<int, int32_t>> ci, vi;
  for (int r = 0; r < ncu; ++r)
    for (int q = 0; q < chunks; ++q)
      ci.push_back({8 * (int)by_row[r].size(), r | (q << 8)});   // also chunks of pruned check nodes only: the item keeps their slots at 0
  for (int c = 0; c < nbu; ++c)
    for (int q = 0; q < chunks; ++q)
      if (c * z + q * 64 < h->n_vn) vi.push_back({col_deg[c] + 2, c | (q << 8)});
  std::vector<int32_t> cp, cl, vp, vl;
  lpt_schedule(ci, h->bp_waves, &cp, &cl);
  lpt_schedule(vi, h->bp_waves, &vp, &vl);
  {
    const std::vector<int> pc = item_priorities(ci, cp, cl), pv = item_priorities(vi, vp, vl);   // see ldpc5g.h
    for (size_t j = 0; j < cl.size(); ++j) cl[j] |= pc[j] << 24;
    for (size_t j = 0; j < vl.size(); ++j) vl[j] |= pv[j] << 24;
  }
  // ldpc5g_onchip_ms.hip: compact per-column edge tables (block byte offset, 4 shift) - a few KB, they stay in the
  // scalar cache - and self-contained two-dword list entries in the same order as the lists above
  std::vector<int32_t> col_ent2, col_start(h->nb, 0), cl2, vl2;
  for (int c = 0; c < h->nb; ++c) {
    col_start[c] = (int32_t)col_ent2.size();
    for (int i = 0; i < col_deg[c]; ++i) {
      const int32_t e = col_ent[(size_t)c * kColStride + i];
      col_ent2.push_back(e & 0x3FFFF);
      col_ent2.push_back((int32_t)((uint32_t)e >> 18));
    }
  }
  col_ent2.resize(col_ent2.size() + 64, 0);                    // wide scalar loads may run past the last edge
  // rows whose last edge is the only edge of its column with shift 0 (extension part of the base graph): that
  // degree-1 VN is updated inside the CN phase (ms_cn_row<D, true>) and leaves the per-iteration VN lists
  std::vector<int> fused_col(h->mb, -1);
  std::vector<char> col_fused(h->nb, 0);
  for (int r = 0; r < ncu; ++r) {
    const int d = (int)by_row[r].size();
    const int c = by_row[r][d - 1].first, sft = by_row[r][d - 1].second;
    if (col_deg[c] == 1 && sft == 0 && d >= 3 && d <= 10) { fused_col[r] = c; col_fused[c] = 1; }
  }
  // two consecutive fully valid chunks of a row / column (degree <= 12) form one pair item (bit 24);
  // weights ~ instructions: 9 (CN) / 5 (VN) per edge and chunk + ~40 for the item's dispatch
  std::vector<std::pair<int, int32_t>> ci2, vi2, vf2;
  // development knobs: per-item overhead of the cost model, capacities by wave launch order
  // (measured at C2 with tools/ms_sweep.py: a fixed cost of ~20 edges per item balances best)
  const int cn_ovh = (int)opt_int("SAMD_MS_CN_OVH", 400);
  const int vn_ovh = (int)opt_int("SAMD_MS_VN_OVH", 200);
  // cost per edge of a pair item / of a single-chunk item, and the fixed cost of a single-chunk VN item.  Round 3, item
  // trace of the grouped kernel (tools/ms_itrace.py, profiles/r03b/ms_itrace_r03d.txt): a CN pair item takes
  // 127 d + 1790 cycles, a VN pair item 115 d + 2240, a single-chunk VN item 111 d + 1040 - one chunk has one dependency
  // chain, so its edges cost what a pair's edge PAIRS cost.
  // 18: r02 model; 36 for the codes of the grouped kernel: +1.7 % at C2 (profiles/r03b/ms_cost_r03f.txt) - the other
  // lifting sizes lose 2-3 % with it (profiles/r03b/ldpc_sweep_r03_s36.json vs ..._s18.json)
  const bool grouped_code = z % 128 == 0 && h->n_cn % z == 0 && h->n_vn % z == 0;
  const int cn_slope = (int)opt_int("SAMD_MS_CN_SLOPE", (grouped_code ? 36 : 18));
  const int vn_single = (int)opt_int("SAMD_MS_VN_SINGLE", 0);   // 1: 10 d + vn_ovh / 2
  // Z not a multiple of 64: the last chunk of a row has `tail` < 64 lifted copies.  With tail <= 32 the tails of
  // 64 / gw rows of the same degree (and fused flag) are packed into one item (lane group g works for row g) - at
  // Z = 80 the 16-lane tails of four rows share a pass instead of running at 25 % lane utilisation each
  const int tail = z - 64 * (chunks - 1);
  int gw = 64;
  while (gw / 2 >= tail && gw > 8) gw /= 2;
  if (opt_set("SAMD_MS_NOPACK")) gw = 64;
  const int groups = 64 / gw;
  int tail_sh = 0;
  while ((1 << tail_sh) < gw) ++tail_sh;
  std::vector<std::vector<int>> packed;                       // rows of every packed-tail item
  std::vector<int> packed_key;                                // degree | fused << 5 of the item
  {
    std::vector<std::vector<int>> bucket(64);
    if (groups >= 2)
      for (int r = 0; r < ncu; ++r) bucket[(int)by_row[r].size() | ((fused_col[r] >= 0) << 5)].push_back(r);
    for (int key = 0; key < 64; ++key)
      for (size_t i = 0; i < bucket[key].size(); i += groups) {
        packed.emplace_back(bucket[key].begin() + i, bucket[key].begin() + std::min(bucket[key].size(), i + groups));
        packed_key.push_back(key);
      }
  }
  for (int r = 0; r < ncu; ++r)
    for (int q = 0; q < chunks; ++q) {
      const int d = (int)by_row[r].size();                    // chunks of pruned check nodes only still get an item (it zeroes their slots)
      if (groups >= 2 && q == chunks - 1) continue;           // the tail goes into a packed item
      const bool pair = (q + 2) * 64 <= z && r * z + (q + 2) * 64 <= h->n_cn &&
                        (fused_col[r] < 0 || fused_col[r] * z + (q + 2) * 64 <= h->n_vn);
      if (pair) { ci2.push_back({cn_slope * d + cn_ovh, r | (q << 8) | (1 << 24)}); ++q; }
      else ci2.push_back({cn_slope / 2 * d + cn_ovh, r | (q << 8)});
    }
  for (size_t i = 0; i < packed.size(); ++i)
    ci2.push_back({9 * (packed_key[i] & 31) + cn_ovh, (int32_t)i | (1 << 25)});
  // the same packing for the tails of the (not fused) columns, by column degree
  std::vector<std::vector<int>> vpacked;
  std::vector<int> vpacked_deg;
  {
    std::vector<std::vector<int>> bucket(64);
    if (groups >= 2)
      for (int c = 0; c < nbu; ++c)
        if (!col_fused[c] && c * z + (chunks - 1) * 64 < h->n_vn && col_deg[c] >= 1 && col_deg[c] <= 30)
          bucket[col_deg[c]].push_back(c);
    for (int dg = 0; dg < 64; ++dg)
      for (size_t i = 0; i < bucket[dg].size(); i += groups) {
        vpacked.emplace_back(bucket[dg].begin() + i, bucket[dg].begin() + std::min(bucket[dg].size(), i + groups));
        vpacked_deg.push_back(dg);
      }
  }
  for (int c = 0; c < nbu; ++c)
    for (int q = 0; q < chunks; ++q) {
      if (c * z + q * 64 >= h->n_vn) continue;
      if (groups >= 2 && q == chunks - 1 && !col_fused[c] && col_deg[c] >= 1 && col_deg[c] <= 30) continue;   // packed below
      const bool pair = col_deg[c] <= 12 && (q + 2) * 64 <= z && c * z + (q + 2) * 64 <= h->n_vn;
      auto& dst = col_fused[c] ? vf2 : vi2;
      if (pair) { dst.push_back({10 * col_deg[c] + vn_ovh, c | (q << 8) | (1 << 24)}); ++q; }
      else if (vn_single) dst.push_back({10 * col_deg[c] + vn_ovh / 2, c | (q << 8)});
      else dst.push_back({5 * col_deg[c] + vn_ovh, c | (q << 8)});
    }
  for (size_t i = 0; i < vpacked.size(); ++i) vi2.push_back({5 * vpacked_deg[i] + vn_ovh, (int32_t)i | (1 << 25)});
  std::vector<int32_t> mcp, mcl, mvp, mvl, mfp, mfl;
  // wave w runs on SIMD w % 4 as its (w / 4)-th oldest wave: optional capacities by age class (SAMD_MS_CAP="a,b,c,d")
  std::vector<double> cap(h->bp_waves, 1.0);
  {
    double cls[4] = {1.0, 1.0, 1.0, 1.0};
    if (opt_set("SAMD_MS_CAP")) sscanf(opt_str("SAMD_MS_CAP").c_str(), "%lf,%lf,%lf,%lf", &cls[0], &cls[1], &cls[2], &cls[3]);
    const int per_simd = std::max(1, h->bp_waves / 4);
    for (int wv = 0; wv < h->bp_waves; ++wv) cap[wv] = cls[std::min(3, (wv / 4) * 4 / per_simd)];
  }
  // Granularity: C2 has 29 variable-node items for 16 waves - the four single-chunk items of the degree-30 / 28 columns
  // are a wave's whole phase and leave it under-loaded, the other waves carry two pair items.  A pair item of the busiest
  // wave is cut into its two single-chunk items (a little more work in total: one dependency chain each) as long as
  // that shortens the longest wave under the LPT rule (SAMD_MS_VN_REFINE = number of cuts tried, 0 = off): +0.5 % at C2,
  // profiles/r03b/ms_refine_r03z.txt.  A half costs half the pair + `extra` (item trace: V12 pair 3.6 k cycles, a
  // single-chunk item of that degree ~2.4 k).
  // Lifting sizes of two chunks only (Z = 128: +1.3 % at C2, +1.1 % at k=2816 n=5632; codes of three or four chunks have
  // enough items per wave and lose 5 % - ms_refine_codes_r03z.txt).
  // development (tools/ms_autotune.py): item costs perturbed by +-SAMD_MS_PERTURB_PCT % from a seeded generator - a
  // search over LPT assignments near the model's by measurement
#ifdef SAMD_DEV                                               // changes the schedule search, not for product builds
  if (opt_set("SAMD_MS_PERTURB")) {
    unsigned long long st = 0x9E3779B97F4A7C15ull * (unsigned long long)(opt_int("SAMD_MS_PERTURB", 0) + 1);
    const int pct = (int)opt_int("SAMD_MS_PERTURB_PCT", 8);
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    for (auto* items : {&ci2, &vi2})
      for (auto& it : *items) it.first = std::max(1, (int)std::lround(it.first * (1.0 + pct / 100.0 * (2.0 * rnd() - 1.0))));
  }
#endif
  auto refine = [&](std::vector<std::pair<int, int32_t>>& items, int tries, int extra, auto&& may_cut) {
    auto makespan = [&](const std::vector<std::pair<int, int32_t>>& its, std::vector<int>* owner) {
      std::vector<size_t> order(its.size());
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return its[a].first > its[b].first; });
      std::vector<double> load(h->bp_waves, 0.0);
      if (owner) owner->assign(its.size(), 0);
      for (size_t i : order) {
        int w = 0;
        for (int q = 1; q < h->bp_waves; ++q)
          if ((load[q] + its[i].first + 3) / cap[q] < (load[w] + its[i].first + 3) / cap[w]) w = q;
        load[w] += its[i].first + 3;
        if (owner) (*owner)[i] = w;
      }
      int mw = 0;
      for (int q = 1; q < h->bp_waves; ++q) if (load[q] > load[mw]) mw = q;
      return std::make_pair(load[mw], mw);
    };
    for (int t = 0; t < tries; ++t) {
      std::vector<int> owner;
      const auto base = makespan(items, &owner);
      int best = -1;
      double best_m = base.first;
      for (size_t i = 0; i < items.size(); ++i) {
        if (owner[i] != base.second || !((items[i].second >> 24) & 1) || !may_cut(items[i].second & 0xFF)) continue;
        auto trial = items;
        const int c = trial[i].second & 0xFF, q = (trial[i].second >> 8) & 0xFF, half = trial[i].first / 2 + extra;
        trial[i] = {half, c | (q << 8)};
        trial.push_back({half, c | ((q + 1) << 8)});
        const double m = makespan(trial, nullptr).first;
        if (m < best_m) { best_m = m; best = (int)i; }
      }
      if (best < 0) break;
      const int c = items[best].second & 0xFF, q = (items[best].second >> 8) & 0xFF, half = items[best].first / 2 + extra;
      items[best] = {half, c | (q << 8)};
      items.push_back({half, c | ((q + 1) << 8)});
    }
  };
  refine(vi2, (int)opt_int("SAMD_MS_VN_REFINE", (chunks == 2 ? 8 : 0)),
         (int)opt_int("SAMD_MS_VN_REFINE_COST", 70), [](int) { return true; });
  // the same for the check-node pair items (the grouped kernel has the single-chunk bodies under key + 64)
  // (C2: +2 %, profiles/r03b/ms_refine_cn3_r03z.txt; rows of degree 5 / 6 with a fused column only - every further
  // single-chunk body in the kernel slows the items that do not use it: all 17 bodies -2 %, four -1.4 %, two -0.3 %)
  refine(ci2, (int)opt_int("SAMD_MS_CN_REFINE", (chunks == 2 ? 4 : 0)),
         (int)opt_int("SAMD_MS_CN_REFINE_COST", 150),
         [&](int r) { const int d = (int)by_row[r].size(); return fused_col[r] >= 0 && d >= 5 && d <= 6; });
  lpt_schedule(ci2, h->bp_waves, &mcp, &mcl, &cap);
  lpt_schedule(vi2, h->bp_waves, &mvp, &mvl, &cap);
  const std::vector<int> cprio = item_priorities(ci2, mcp, mcl), vprio = item_priorities(vi2, mvp, mvl);
  lpt_schedule(vf2, h->bp_waves, &mfp, &mfl);
  for (int32_t o : mfp) mvp.push_back(o + (int32_t)mvl.size());
  mvl.insert(mvl.end(), mfl.begin(), mfl.end());
  std::vector<int32_t> tail_tab;
  for (size_t i = 0; i < packed.size(); ++i)
    for (int g = 0; g < groups; ++g) {
      if (g < (int)packed[i].size()) {
        const int r = packed[i][g];
        tail_tab.push_back(row_off[r] & 0x3FFFF);
        tail_tab.push_back(r | ((fused_col[r] >= 0 ? fused_col[r] : 0) << 16));
      } else {
        tail_tab.push_back(0);
        tail_tab.push_back(0xFF);
      }
    }
  tail_tab.resize(tail_tab.size() + 2, 0);
  h->ms_tail_sh = tail_sh;
  for (size_t j = 0; j < mcl.size(); ++j) {
    const int32_t d = mcl[j];
    if ((d >> 25) & 1) {                                      // packed tails: table base | key, last chunk | flag
      const int i = d & 0xFFFF;
      cl2.push_back((int32_t)(i * groups) | (packed_key[i] << 18));
      cl2.push_back(((chunks - 1) << 8) | (1 << 26) | (cprio[j] << 24));
      continue;
    }
    const int r = d & 0xFF, f = fused_col[r] >= 0, pr = (d >> 24) & 1;
    cl2.push_back((row_off[r] & 0x3FFFF) | (((int)by_row[r].size() | (f << 5) | (pr << 6)) << 18));
    cl2.push_back((d & 0xFFFF) | ((f ? fused_col[r] : 0) << 16) | (cprio[j] << 24));
  }
  std::vector<int32_t> vtail_tab;
  for (size_t i = 0; i < vpacked.size(); ++i)
    for (int g = 0; g < groups; ++g) {
      const bool has = g < (int)vpacked[i].size();
      vtail_tab.push_back(has ? col_start[vpacked[i][g]] : 0);
      vtail_tab.push_back(has ? vpacked[i][g] : 0xFF);
    }
  vtail_tab.resize(vtail_tab.size() + 2, 0);
  for (size_t j = 0; j < mvl.size(); ++j) {
    const int32_t d = mvl[j];
    if ((d >> 25) & 1) {                                      // packed tails: last chunk | degree | flag, table base
      const int i = d & 0xFFFF;
      vl2.push_back(((chunks - 1) << 8) | (vpacked_deg[i] << 16) | (1 << 22) | ((j < vprio.size() ? vprio[j] : 0) << 24));
      vl2.push_back(i * groups);
      continue;
    }
    const int c = d & 0xFF, pr = (d >> 24) & 1;
    vl2.push_back((d & 0xFFFF) | ((col_deg[c] | (pr << 5)) << 16) | ((j < vprio.size() ? vprio[j] : 0) << 24));
    vl2.push_back(col_start[c]);
  }
  cl2.resize(cl2.size() + 2, 0); vl2.resize(vl2.size() + 2, 0);
  // ---- grouped dispatch (ldpc5g_decode_msg_kernel): the items of every wave sorted by body type.  Only for codes whose
  // items are all full chunk pairs (Z a multiple of 128, no partially pruned base row) on 16 waves.
  std::vector<int32_t> g_ptr, g_cn, g_vn, i_cn, i_vn, d_cn, d_vn;      // d_*: dataflow records, 4 ints per item (see below)
  h->ms_g_ok = 0;
  h->ms_df_ok = 0;
  // dataflow readiness (ldpc5g_decode_msg_kernel, experiment of round 4): instead of two workgroup barriers per iteration,
  // every item waits for the items it reads from and publishes its own completion through monotone counters in LDS -
  // rowcnt[r] / colcnt[compact c] count the 64-lane chunks of row r / column c updated so far (`chunks` per iteration).
  // A check-node item of row r in iteration `it` needs colcnt[c] >= chunks it for the non-fused columns c of its row, a
  // variable-node item of column c needs rowcnt[r] >= chunks (it + 1) for the rows r of its column: the masks over at most
  // 64 counters travel in the item's record {need lo, need hi, byte offset of the own counter, increment}.
  std::vector<int> col_compact(h->nb, -1);
  int n_compact = 0;
  for (int c = 0; c < nbu; ++c)
    if (!col_fused[c] && col_deg[c] > 0) col_compact[c] = n_compact++;
  std::vector<unsigned long long> row_need(ncu, 0ull), col_need(h->nb, 0ull);
  for (int r = 0; r < ncu; ++r)
    for (auto& e : by_row[r]) {
      const int c = e.first;
      if (c < nbu && col_compact[c] >= 0 && col_compact[c] < 64) row_need[r] |= 1ull << col_compact[c];
      if (c < nbu && r < 64) col_need[c] |= 1ull << r;
    }
  const bool df_shape = n_compact <= 64 && ncu <= 64;
  if (z % 128 == 0 && h->n_cn % z == 0 && h->n_vn % z == 0 && h->bp_waves == 16 && groups < 2) {
    bool ok = true;
    struct It { int cost, key; int32_t x, y; int idx, pr, q; };
    JitPlan* plan = new JitPlan();
    plan->cn.resize(h->bp_waves); plan->vn.resize(h->bp_waves);
    auto cost_in = [](const std::vector<std::pair<int, int32_t>>& items, int32_t id) {
      for (auto& it : items)
        if (it.second == id) return it.first;
      return 0;
    };
    const int nwv = h->bp_waves;
    for (int phase = 0; phase < 2 && ok; ++phase) {
      const std::vector<int32_t>& lp = phase ? mvp : mcp;
      const std::vector<int32_t>& ll = phase ? mvl : mcl;
      std::vector<std::vector<It>> per(nwv);
      double longest = 1.0;
      for (int wv = 0; wv < nwv; ++wv) {
        double tot = 0.0;
        for (int j = lp[wv]; j < lp[wv + 1]; ++j) {
          const int32_t d = ll[j];
          if ((d >> 25) & 1) { ok = false; break; }
          const int idx = d & 0xFF, q = (d >> 8) & 0xFF, pr = (d >> 24) & 1;
          It it;
          it.idx = idx; it.pr = pr; it.q = q;
          it.cost = cost_in(phase ? vi2 : ci2, d) + 3;
          if (!phase) {
            const int f = fused_col[idx] >= 0;
            it.key = (int)by_row[idx].size() | (f << 5) | (pr ? 0 : 64);      // (64: a single chunk of the row)
            it.x = (row_off[idx] & 0x3FFFF) + 256 * q;
            it.y = f ? fused_col[idx] * z + q * 64 : 0;
          } else {
            it.key = col_deg[idx] | (pr << 5);
            it.x = (idx * z + q * 64) | (pr << 23);
            it.y = col_start[idx] | (q << 20);
            if (col_start[idx] >= (1 << 20) || idx * z + q * 64 >= (1 << 23)) ok = false;
          }
          // every key must have a compiled body in ldpc5g_decode_msg_kernel (its switches end in `default: break`, which
          // would skip the item silently): CN degrees 3..10 and 19, plain or with a fused degree-1 column (+32, up to 10),
          // single-chunk CN items only as 64+37 / 64+38; VN degrees 1..30, chunk pairs (+32) up to degree 12
          if (!phase) {
            const int dgr = it.key & 31, fz = (it.key >> 5) & 1, single = (it.key >> 6) & 1;
            const bool body = single ? (fz && (dgr == 5 || dgr == 6)) : ((dgr >= 3 && dgr <= 10) || (dgr == 19 && !fz));
            if (!body) ok = false;
          } else {
            const int dgr = it.key & 31, pair = (it.key >> 5) & 1;
            if (dgr < 1 || dgr > 30 || (pair && dgr > 12)) ok = false;
          }
          tot += it.cost;
          per[wv].push_back(it);
        }
        longest = std::max(longest, tot);
      }
      if (!ok) break;
      std::vector<int32_t>& gl = phase ? g_vn : g_cn;
      std::vector<int32_t>& il = phase ? i_vn : i_cn;
      std::vector<int32_t>& dl = phase ? d_vn : d_cn;
      for (int wv = 0; wv < nwv; ++wv) {
        g_ptr.push_back((int32_t)gl.size() / 2);
        // groups by type, the type with the most expensive items first; the group order is rotated by the wave's index
        // on its SIMD so that the four waves of a SIMD start a phase in different bodies (cf. SAMD_MS_ORDER above)
        std::stable_sort(per[wv].begin(), per[wv].end(), [](const It& a, const It& b) {
          return a.cost != b.cost ? a.cost > b.cost : a.key > b.key; });
        std::vector<std::vector<It>> grp;
        for (const It& it : per[wv]) {
          if (grp.empty() || grp.back().front().key != it.key) {
            size_t k2 = 0;
            for (; k2 < grp.size(); ++k2)
              if (grp[k2].front().key == it.key) break;
            if (k2 == grp.size()) grp.emplace_back();
            grp[k2].push_back(it);
          } else {
            grp.back().push_back(it);
          }
        }
        if (grp.size() > 1 && !opt_set("SAMD_MS_NOROT")) std::rotate(grp.begin(), grp.begin() + ((wv >> 2) % grp.size()), grp.end());
        double rem = 0.0;
        for (auto& gq : grp)
          for (auto& it : gq) rem += it.cost;
        for (auto& gq : grp) {
          const size_t first = il.size() / 2;
          for (auto& it : gq) {
            const int prio = opt_set("SAMD_MS_NOPRIO") ? 0 : std::max(0, std::min(3, (int)std::ceil(4.0 * rem / longest) - 1));
            rem -= it.cost;
            il.push_back(it.x | (prio << 24));
            il.push_back(it.y);
            (phase ? plan->vn : plan->cn)[wv].push_back(JitItem{it.idx, it.q, it.pr ? 2 : 1, prio});
            const unsigned long long need = phase ? col_need[it.idx] : row_need[it.idx];
            dl.push_back((int32_t)(unsigned)(need & 0xFFFFFFFFull));
            dl.push_back((int32_t)(unsigned)(need >> 32));
            dl.push_back(phase ? 8 * col_compact[it.idx] + 4 : 8 * it.idx);   // counters interleaved: {rowcnt[i], colcnt[i]} per lane
            dl.push_back(it.pr ? 2 : 1);
          }
          gl.push_back(gq.front().key);
          gl.push_back((int32_t)(first | ((il.size() / 2) << 16)));
        }
      }
      g_ptr.push_back((int32_t)gl.size() / 2);
      if (il.size() / 2 >= 65536) ok = false;
    }
    h->ms_g_ok = ok && g_ptr.size() == (size_t)(2 * (nwv + 1)) ? 1 : 0;
    if (h->ms_g_ok) {
      // the same schedule as data for the source generator of the specialised kernel (ldpc5g_jit.cpp)
      plan->z = z; plan->edges = edges; plan->ncu = ncu; plan->nbu = nbu;
      plan->row_off.resize(ncu); plan->row_deg.resize(ncu); plan->fused_col.assign(fused_col.begin(), fused_col.begin() + ncu);
      for (int r = 0; r < ncu; ++r) { plan->row_off[r] = row_off[r] & 0x3FFFF; plan->row_deg[r] = (int)by_row[r].size(); }
      plan->col_edges.resize(nbu); plan->col_fused.assign(col_fused.begin(), col_fused.begin() + nbu);
      for (int c = 0; c < nbu; ++c)
        for (int i = 0; i < col_deg[c]; ++i)
          plan->col_edges[c].push_back({col_ent2[col_start[c] + 2 * i], col_ent2[col_start[c] + 2 * i + 1]});
      h->jit_plan = plan;
    } else {
      delete plan;
    }
    g_cn.resize(g_cn.size() + 2, 0); g_vn.resize(g_vn.size() + 2, 0);
    i_cn.resize(i_cn.size() + 4, 0); i_vn.resize(i_vn.size() + 4, 0);
    d_cn.resize(d_cn.size() + 8, 0); d_vn.resize(d_vn.size() + 8, 0);
    // every item of a row / column a full chunk pair's worth per iteration, the counters fit beside the messages
    h->ms_df_ok = (h->ms_g_ok && df_shape && onchip_bp_lds_bytes(h) + 512 <= 160 * 1024) ? 1 : 0;
    if (!h->ms_g_ok) g_ptr.assign(2 * (nwv + 1), 0);
  }
  int rc = upload(&h->bp_row_off, row_off.data(), row_off.size());
  if (rc == SAMD_OK && h->ms_g_ok) rc = upload(&h->ms_g_ptr, g_ptr.data(), g_ptr.size());
  if (rc == SAMD_OK && h->ms_g_ok) rc = upload(&h->ms_g_cn, g_cn.data(), g_cn.size());
  if (rc == SAMD_OK && h->ms_g_ok) rc = upload(&h->ms_g_vn, g_vn.data(), g_vn.size());
  if (rc == SAMD_OK && h->ms_g_ok) rc = upload(&h->ms_i_cn, i_cn.data(), i_cn.size());
  if (rc == SAMD_OK && h->ms_g_ok) rc = upload(&h->ms_i_vn, i_vn.data(), i_vn.size());
  if (rc == SAMD_OK && h->ms_df_ok) rc = upload(&h->ms_d_cn, d_cn.data(), d_cn.size());
  if (rc == SAMD_OK && h->ms_df_ok) rc = upload(&h->ms_d_vn, d_vn.data(), d_vn.size());
  if (rc == SAMD_OK) rc = upload(&h->bp_col_ent, col_ent.data(), col_ent.size());
  if (rc == SAMD_OK) rc = upload(&h->ms_col_ent, col_ent2.data(), col_ent2.size());
  if (rc == SAMD_OK) rc = upload(&h->ms_cn_ptr, mcp.data(), mcp.size());
  if (rc == SAMD_OK) rc = upload(&h->ms_vn_ptr, mvp.data(), mvp.size());
  if (rc == SAMD_OK) rc = upload(&h->ms_cn_list, cl2.data(), cl2.size());
  if (rc == SAMD_OK) rc = upload(&h->ms_vn_list, vl2.data(), vl2.size());
  if (rc == SAMD_OK) rc = upload(&h->ms_tail_tab, tail_tab.data(), tail_tab.size());
  if (rc == SAMD_OK) rc = upload(&h->ms_vtail_tab, vtail_tab.data(), vtail_tab.size());
  if (rc == SAMD_OK) rc = upload(&h->bp_col_deg, col_deg.data(), col_deg.size());
  if (rc == SAMD_OK) rc = upload(&h->bp_cn_ptr, cp.data(), cp.size());
  if (rc == SAMD_OK) rc = upload(&h->bp_cn_list, cl.data(), cl.size());
  if (rc == SAMD_OK) rc = upload(&h->bp_vn_ptr, vp.data(), vp.size());
  if (rc == SAMD_OK) rc = upload(&h->bp_vn_list, vl.data(), vl.size());
  if (rc == SAMD_OK) h->bp_ok = 1;
  return rc;
}

void free_onchip_bp_tables(samd_ldpc5g* h) {
  (void)hipFree(h->bp_row_off); (void)hipFree(h->bp_col_ent); (void)hipFree(h->bp_col_deg); (void)hipFree(h->ms_col_ent); (void)hipFree(h->ms_cn_list); (void)hipFree(h->ms_cn_ptr); (void)hipFree(h->ms_vn_ptr); (void)hipFree(h->ms_vn_list); (void)hipFree(h->ms_tail_tab); (void)hipFree(h->ms_vtail_tab);
  (void)hipFree(h->ms_g_ptr); (void)hipFree(h->ms_g_cn); (void)hipFree(h->ms_g_vn); (void)hipFree(h->ms_i_cn); (void)hipFree(h->ms_i_vn);
  (void)hipFree(h->ms_d_cn); (void)hipFree(h->ms_d_vn);
  (void)hipFree(h->bp_cn_ptr); (void)hipFree(h->bp_cn_list); (void)hipFree(h->bp_vn_ptr); (void)hipFree(h->bp_vn_list);
}

size_t onchip_bp_lds_bytes(const samd_ldpc5g* h) {
  const int nbu = (h->n_vn + h->z - 1) / h->z;
  return (size_t)h->bp_edges * h->z * 4 + (h->bp_llr_global ? 0 : (size_t)nbu * h->z * 4);
}

int onchip_bp_grid(const samd_ldpc5g* h, int batch) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const size_t per_cu = std::min<size_t>((size_t)(kDecWaves / h->bp_waves), std::max<size_t>(1, (160 * 1024) / onchip_bp_lds_bytes(h)));
  size_t grid = std::min<size_t>((size_t)batch, (size_t)cus * per_cu);
  // SAMD_ONCHIP_GRID=<n>: fewer workgroups than the chip holds (test hook: every workgroup then decodes several
  // codewords in sequence, which exercises the grid-stride loop and the reuse of its workspace row on small batches)
  if (h->opt.onchip_grid > 0) grid = std::min<size_t>(grid, (size_t)h->opt.onchip_grid);
  return (int)grid;
}

size_t onchip_bp_workspace_bytes(const samd_ldpc5g* h, int batch) {
  if (!h->bp_ok || !h->bp_llr_global || batch <= 0) return 0;
  const int nbu = (h->n_vn + h->z - 1) / h->z;
  return (size_t)onchip_bp_grid(h, batch) * nbu * h->z * sizeof(float) + 256;
}

int launch_onchip_bp(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                     float llr_max, int hard_out, int return_infobits, void* workspace, size_t workspace_bytes,
                     hipStream_t st) {
  if (!h->bp_ok) {
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
  const bool phi = (cn_mode == SAMD_CN_BOXPLUS_PHI);
  typedef void (*kern_t)(const float*, float*, float*, RateMatch, int, int, int, int, float, int, int, int,
                         const int32_t*, const int32_t*, const int32_t*, const int32_t*, const int32_t*,
                         const int32_t*, const int32_t*);
#define SAMD_BP_K(M, NWV, G) ldpc5g_decode_bp_kernel<M, false, NWV, G>, ldpc5g_decode_bp_kernel<M, true, NWV, G>
  static const kern_t kerns[24] = {
      SAMD_BP_K(SAMD_CN_BOXPLUS_PHI, 16, false), SAMD_BP_K(SAMD_CN_BOXPLUS_PHI, 8, false),
      SAMD_BP_K(SAMD_CN_BOXPLUS_PHI, 4, false),  SAMD_BP_K(SAMD_CN_BOXPLUS_PHI, 2, false),
      SAMD_BP_K(SAMD_CN_BOXPLUS_PHI, 1, false),  SAMD_BP_K(SAMD_CN_BOXPLUS_PHI, 16, true),
      SAMD_BP_K(SAMD_CN_BOXPLUS, 16, false),     SAMD_BP_K(SAMD_CN_BOXPLUS, 8, false),
      SAMD_BP_K(SAMD_CN_BOXPLUS, 4, false),      SAMD_BP_K(SAMD_CN_BOXPLUS, 2, false),
      SAMD_BP_K(SAMD_CN_BOXPLUS, 1, false),      SAMD_BP_K(SAMD_CN_BOXPLUS, 16, true)};
#undef SAMD_BP_K
  const int nw = h->bp_waves;
  const int wi = h->bp_llr_global ? 5 : (nw == 16 ? 0 : nw == 8 ? 1 : nw == 4 ? 2 : nw == 2 ? 3 : 4);
  const int ki = (phi ? 0 : 12) + 2 * wi + (pow2 ? 1 : 0);
  // set on every launch: the attribute is per device and a process may drive several
  SAMD_SET_MAX_LDS(kerns[ki], 160 * 1024);
  const int nbu = (h->n_vn + h->z - 1) / h->z;
  const RateMatch rm = make_rate_match(h);
  hipLaunchKernelGGL(kerns[ki], dim3(onchip_bp_grid(h, batch)), dim3(nw * 64), onchip_bp_lds_bytes(h), st, llr, out, llr_ws, rm,
                     h->n_cn, nbu, batch, num_iter, llr_max, hard_out, return_infobits, h->bp_edges * h->z,
                     h->bp_row_off, h->bp_col_ent, h->bp_col_deg, h->bp_cn_ptr, h->bp_cn_list,
                     h->bp_vn_ptr, h->bp_vn_list);
  return launch_status();
}

}  // namespace samd
