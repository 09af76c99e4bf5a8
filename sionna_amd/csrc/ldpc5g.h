// Shared declarations of the 5G-NR QC-LDPC kernels (handle layout, rate-matching index maps).
#pragma once
#include "common.h"
#include "options.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>
#include <utility>
#include <vector>

// Development switches a handle captured when it was created (options.h): its launch paths read these, never the
// environment, so a handle behaves the same for its whole life.
struct samd_ldpc5g_opt {
  bool enc_bytes = false, enc_persist = false, onchip_compressed = false, force_spill = false, no_spill = false;
  bool no_onchip_layered = false, bp_engine = false, onchip_v1 = false, ms_nogroup = false, ms_noz128 = false;
  int ms_var = 1, ms_ldsbar = 0, onchip_grid = 0, enc_dbg = 0, ms_dataflow = 0;
  // boxplus-phi on the generated kernel (bit-identical).  Round 5, loops unrolled: SLOWER than the generic explicit-message
  // kernel (437 KB of code for 16 wave programs, 1076 spilled registers: 144.7 against 119.9 ms per 65536 C2 codewords).
  // Round 6, check-node loops rolled (jit_cn_phi_rolled): 113.0 ms at C2 (0.580 M against 0.546 M decodes/s), 1.5 - 2 x the
  // generic engines on the codes that pack several codewords per workgroup (C4's code 3.45 M against 1.75 M), but 0.91 x where
  // ONE codeword leaves a quarter of the lanes of its only chunk idle (Z = 96: the rule is bound by the vector pipe, idle
  // lanes cost their full share) - profiles/r06m.  -1 (default): on, except for that geometry; 0 off; 1 on.
  int jit_phi = -1;
  // specialised kernels (ldpc5g_jit.cpp): 0 off, 1 from jit_min_batch codewords, 2 always.  The generated kernel is the faster
  // one at EVERY batch size (16 ... 4096 codewords, five codes: profiles/r06k_jit_small_batch.json); the threshold only keeps
  // one-off small decodes (unit tests over dozens of codes) from paying ~3 s of hipRTC each the first time a machine sees a
  // code - later processes load the cached code object (256 since round 6, 1024 before the disk cache)
  int jit = 1, jit_min_batch = 256;
  void capture() {
    using samd::opt_set; using samd::opt_int;
    enc_bytes = opt_set("SAMD_ENC_BYTES"); enc_persist = opt_set("SAMD_ENC_PERSIST");
    onchip_compressed = opt_set("SAMD_ONCHIP_COMPRESSED"); force_spill = opt_set("SAMD_FORCE_SPILL");
    no_spill = opt_set("SAMD_NO_SPILL"); no_onchip_layered = opt_set("SAMD_NO_ONCHIP_LAYERED");
    bp_engine = opt_set("SAMD_BP_ENGINE"); onchip_v1 = opt_set("SAMD_ONCHIP_V1");
    ms_nogroup = opt_set("SAMD_MS_NOGROUP"); ms_noz128 = opt_set("SAMD_MS_NOZ128");
    ms_var = (int)opt_int("SAMD_MS_VAR", 1) & 1; ms_ldsbar = (int)opt_int("SAMD_MS_LDSBAR", 0);
    onchip_grid = (int)opt_int("SAMD_ONCHIP_GRID", 0);
    ms_dataflow = (int)opt_int("SAMD_MS_DATAFLOW", 0);
    jit_phi = (int)opt_int("SAMD_LDPC_JIT_PHI", -1);
    jit = (int)opt_int("SAMD_LDPC_JIT", 1); jit_min_batch = (int)opt_int("SAMD_LDPC_JIT_MIN_BATCH", 256);
#ifdef SAMD_DEV
    enc_dbg = (int)opt_int("SAMD_ENC_DBG", 0);        // skips encoder phases: wrong results, development builds only
#endif
  }
};

namespace samd { struct JitPlan; struct JitState; }

struct samd_ldpc5g {
  samd_ldpc5g_opt opt;
  int host_only = 0;           // built without a device (SAMD_HOST_ONLY): tables and schedules only, no launch
  samd::JitPlan* jit_plan = nullptr;     // schedule of the specialised kernel (ldpc5g_jit.h); null: code outside its class
  samd::JitState* jit_state = nullptr;   // compiled modules, created lazily (mutable behind a const handle)
  int bg = 0, z = 0, k = 0, n = 0, m_int = 0, nb_pruned = 0;
  int mb = 0, nb = 0, k_b = 0, k_ldpc = 0, n_ldpc = 0, n_vn = 0, n_cn = 0;
  int s_a = 0, s_b = 0;  // shifts of the core entries P_A, P_B (encoding.py:476-481)
  int nnz = 0, max_dc = 0, max_dv = 0;
  // base graph, CSR by row (entries ascending column) - device
  int32_t* row_ptr = nullptr;  // [mb+1]
  int32_t* row_ent = nullptr;  // [nnz]  col | shift<<16   (shift already mod Z)
  // base graph, CSC by column (entries ascending row) - device
  int32_t* col_ptr = nullptr;  // [nb+1]
  int32_t* col_ent = nullptr;  // [nnz]  row | shift<<8 | pos<<20  (pos = index of the edge inside its row)
  // work items for the decoder, longest first: (index | chunk<<16)
  uint16_t* enc_out_idx = nullptr;   // [n] bit-packed encoder (Z % 32 == 0): output position -> position in the full codeword
  int32_t* cn_items = nullptr; int n_cn_items = 0;
  int32_t* vn_items = nullptr; int n_vn_items = 0;
  // ---- tables of the statically scheduled on-chip decoder (csrc/ldpc5g_onchip.hip)
  int v2_ok = 0;               // all row degrees have an unrolled instantiation
  int ncu = 0, nbu = 0;        // base rows / columns that hold at least one un-pruned node
  int32_t* row_pad = nullptr;  // [mb*20]  (c*z) | shift<<16, entries ascending column
  int32_t* row_deg = nullptr;  // [mb]
  int32_t* col_pad = nullptr;  // [nb*32]  (r*z) | shift<<16 | pos<<25, padded with the zero dummy CN block
  int32_t* col_cls = nullptr;  // [nb]     unrolled class size (>= column degree)
  int32_t* cn_sched_ptr = nullptr; int32_t* cn_sched = nullptr;   // per-wave item lists (LPT balanced)
  int32_t* vn_sched_ptr = nullptr; int32_t* vn_sched = nullptr;
  // ---- on-chip boxplus / boxplus-phi engine (csrc/ldpc5g_onchip_bp.hip): one float per edge in LDS
  int bp_ok = 0, bp_waves = 16, bp_llr_global = 0, bp_edges = 0;   // bp_edges: base-graph edges of the used rows
  int32_t* bp_row_off = nullptr;   // [mb]     byte offset of the row's first edge block (blocks are Z floats) | degree << 18
  int32_t* bp_col_ent = nullptr;   // [nb*32]  (edge block byte offset) | (4*shift) << 18, rows ascending
  // explicit-message min-sum engine (ldpc5g_onchip_ms.hip): compact edge tables + self-contained list entries
  int32_t* ms_col_ent = nullptr;   // [2 E + pad] per column, rows ascending: edge block byte offset, 4*shift
  int32_t* ms_cn_ptr = nullptr; int32_t* ms_vn_ptr = nullptr;   // per-wave list offsets (VN: per-iteration lists, then fused columns)
  int32_t* ms_cn_list = nullptr;   // [2 items]  row block byte offset | (degree | fused<<5)<<18,  r | chunk<<8 | fused column<<16
  int32_t* ms_vn_list = nullptr;   // [2 items]  c | chunk<<8 | degree<<16,  dword offset into ms_col_ent
  int32_t* ms_tail_tab = nullptr;  // [2 x groups] packed-tail items: row block byte offset, r | fused column<<16 (r = 0xFF: no row)
  int ms_tail_sh = 6;              // log2 of the lane-group width of a packed-tail item
  int32_t* ms_vtail_tab = nullptr; // [2 x groups] packed-tail VN items: dword offset of the column's edge table, c (0xFF: no column)
  // grouped dispatch of the same items (ldpc5g_decode_msg_kernel): items of a wave sorted by body type
  int ms_g_ok = 0;
  int32_t* ms_g_ptr = nullptr;     // [2 (NW+1)] group offsets per wave: CN groups, then VN groups
  int32_t* ms_g_cn = nullptr; int32_t* ms_g_vn = nullptr;   // [2 groups] body type, first item | (last + 1) << 16
  int32_t* ms_i_cn = nullptr; int32_t* ms_i_vn = nullptr;   // [2 items]  see ldpc5g_onchip_ms.inc
  int ms_df_ok = 0;                // dataflow readiness instead of the two barriers per iteration (SAMD_MS_DATAFLOW)
  int32_t* ms_d_cn = nullptr; int32_t* ms_d_vn = nullptr;   // [4 items]  need mask lo / hi, own counter byte offset, increment
  int32_t* bp_col_deg = nullptr;   // [nb]
  int32_t* bp_cn_ptr = nullptr; int32_t* bp_cn_list = nullptr;     // per-wave item lists (LPT balanced)
  int32_t* bp_vn_ptr = nullptr; int32_t* bp_vn_list = nullptr;
  // explicit-message min-sum engine with the last rows' messages in the L2 workspace row (ldpc5g_onchip_mss.hip)
  int sp_ok = 0, sp_lds_bytes = 0, sp_g_floats = 0, sp_spill_pct = 0;   // sp_spill_pct: share of the edges in L2
  int32_t* sp_col_ent = nullptr; int32_t* sp_cn_ptr = nullptr; int32_t* sp_vn_ptr = nullptr;
  int32_t* sp_cn_list = nullptr; int32_t* sp_vn_list = nullptr;
  // on-chip layered decoder (ldpc5g_onchip_ly.hip): record lists per wave, row / column edge tables
  int ly_ok = 0, ly_lds_bytes = 0, ly_msg_floats = 0, ly_n_ext = 0, ly_groups = 0, ly_zero_off = 0, ly_waves = 16;
  int32_t* ly_rec_ptr = nullptr; int32_t* ly_recs = nullptr; int32_t* ly_ent_tab = nullptr;
  int32_t* ly_xt_index = nullptr; int32_t* ly_slot_tab = nullptr;
  int32_t* ly_bp_rec_ptr = nullptr; int32_t* ly_bp_recs = nullptr; int32_t* ly_bp_slot_tab = nullptr; int ly_bp_lds_bytes = 0;
  int dec_waves = 16;          // waves per workgroup of the on-chip decoder (16 / 8 / 4: small codes share a CU)
  int llr_global = 0;          // 1: channel LLRs in the caller's workspace (L2) instead of LDS (larger codes fit)
};

namespace samd {
constexpr int kRowStride = 20;   // max row degree of BG1 is 19
constexpr int kColStride = 32;   // max column degree of BG1 is 30
constexpr int kDecWaves = 16;    // waves of one CU's decoder workgroups (1 x 16, 2 x 8 or 4 x 4)
// build the v2 tables (host); defined in ldpc5g_onchip.hip
int build_onchip_tables(samd_ldpc5g* h, const std::vector<std::vector<std::pair<int, int>>>& by_row);
void free_onchip_tables(samd_ldpc5g* h);
size_t onchip_workspace_bytes(const samd_ldpc5g* h, int batch);
int build_onchip_bp_tables(samd_ldpc5g* h, const std::vector<std::vector<std::pair<int, int>>>& by_row);
void free_onchip_bp_tables(samd_ldpc5g* h);
size_t onchip_bp_workspace_bytes(const samd_ldpc5g* h, int batch);
size_t onchip_bp_lds_bytes(const samd_ldpc5g* h);
int onchip_bp_grid(const samd_ldpc5g* h, int batch);
// explicit-message min-sum engine (ldpc5g_onchip_ms.hip), same tables as the boxplus engine
int launch_onchip_ms(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                     float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                     size_t workspace_bytes, hipStream_t st);
int launch_onchip_bp(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                     float llr_max, int hard_out, int return_infobits, void* workspace, size_t workspace_bytes,
                     hipStream_t st);
int build_onchip_mss_tables(samd_ldpc5g* h, const std::vector<std::vector<std::pair<int, int>>>& by_row);
void free_onchip_mss_tables(samd_ldpc5g* h);
size_t onchip_mss_workspace_bytes(const samd_ldpc5g* h, int batch);
int launch_onchip_mss(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                      float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                      size_t workspace_bytes, hipStream_t st);
int build_onchip_ly_tables(samd_ldpc5g* h, const std::vector<std::vector<std::pair<int, int>>>& by_row);
void free_onchip_ly_tables(samd_ldpc5g* h);
size_t onchip_ly_workspace_bytes(const samd_ldpc5g* h, int batch);
int launch_onchip_ly(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode, float llr_max,
                     float offset, int hard_out, int return_infobits, void* workspace, size_t workspace_bytes, hipStream_t st);
int launch_onchip_v2(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                     float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                     size_t workspace_bytes, hipStream_t st);
}

namespace samd {

// longest-processing-time-first assignment of items to the waves of the workgroup.
// cap (optional, one entry per wave): relative capacity of a wave - an item goes to the wave whose load / capacity
// is smallest after taking it.  The SIMD arbiter serves its OLDEST wave first (measured with the trace build of
// ldpc5g_onchip_ms.hip: of four equally loaded waves of a SIMD the first-launched finished a phase after 4.3 k
// cycles, the last after 8.4 k - it mostly runs in the gaps of the older ones and then alone, latency-bound), so
// waves that are launched later get less work.
inline void lpt_schedule(const std::vector<std::pair<int, int32_t>>& items, int nw, std::vector<int32_t>* ptr,
                         std::vector<int32_t>* list, const std::vector<double>* cap = nullptr) {
  std::vector<size_t> order(items.size());
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return items[a].first > items[b].first; });
  std::vector<std::vector<int32_t>> per(nw);
  std::vector<double> load(nw, 0.0);
  for (size_t i : order) {
    const double cost = items[i].first + 3;                // + fixed per-item overhead
    int w = 0;
    double best = 0.0;
    for (int q = 0; q < nw; ++q) {
      const double t = (load[q] + cost) / (cap ? (*cap)[q] : 1.0);
      if (q == 0 || t < best) { best = t; w = q; }
    }
    per[w].push_back(items[i].second);
    load[w] += cost;
  }
  // order of a wave's items (SAMD_MS_ORDER; default 2: +2.5 % at C2 - the four waves of a SIMD start a phase on items of
  // different size instead of all on their largest): 0 = largest first on every wave; 1 = odd waves
  // smallest first; 2 = rotated by the wave's index on its SIMD; 3 = waves 2, 3 (mod 4) smallest first
  const int order_mode = (int)opt_int("SAMD_MS_ORDER", 2);       // handle-build time
  for (int w = 0; w < nw && order_mode; ++w) {
    if (per[w].size() < 2) continue;
    if ((order_mode == 1 && (w & 1)) || (order_mode == 3 && (w & 2))) std::reverse(per[w].begin(), per[w].end());
    if (order_mode == 2) std::rotate(per[w].begin(), per[w].begin() + ((w >> 2) % per[w].size()), per[w].end());
    if (order_mode == 4 && ((w >> 2) & 1)) std::reverse(per[w].begin(), per[w].end());
    if (order_mode == 5) std::rotate(per[w].begin(), per[w].begin() + (w % per[w].size()), per[w].end());
    if (order_mode == 6) std::rotate(per[w].begin(), per[w].begin() + ((2 * (w >> 2)) % per[w].size()), per[w].end());
    if (order_mode == 7) std::rotate(per[w].begin(), per[w].begin() + ((w & 3) % per[w].size()), per[w].end());
    if (order_mode == 8) std::rotate(per[w].begin(), per[w].begin() + (((w >> 2) + (w & 3)) % per[w].size()), per[w].end());
  }
  ptr->assign(1, 0);
  list->clear();
  for (int w = 0; w < nw; ++w) {
    list->insert(list->end(), per[w].begin(), per[w].end());
    ptr->push_back((int32_t)list->size());
  }
}

// Issue priority by remaining work.  The SIMD arbiter serves its oldest wave first, so equally loaded waves do not
// finish a phase together (trace build of ldpc5g_onchip_ms.hip at C2: 4.3 k vs 8.4 k cycles for the first- and the
// last-launched wave of a SIMD, 31 % of an iteration spent waiting at the two barriers).  Every item of a wave's list
// therefore carries a priority 0..3 = the quarter of the phase's longest list that is still ahead of the wave; the
// kernels apply it with s_setprio before the item (onchip_setprio), which turns the arbiter into "longest remaining
// work first": the waves of a SIMD finish together (C2 min-sum: 2.10 -> 2.30 M decodes/s with the same lists).
// items: (cost, id) as given to lpt_schedule; ptr / list: its result.  SAMD_MS_NOPRIO=1 disables (development).
inline std::vector<int> item_priorities(const std::vector<std::pair<int, int32_t>>& items, const std::vector<int32_t>& ptr,
                                        const std::vector<int32_t>& list) {
  std::vector<int> prio(list.size(), 0);
  if (opt_set("SAMD_MS_NOPRIO")) return prio;
  auto cost_of = [&](int32_t id) {
    for (auto& it : items)
      if (it.second == id) return it.first + 3;
    return 3;
  };
  double longest = 1.0;
  std::vector<double> tot(ptr.size() - 1, 0.0);
  for (size_t wv = 0; wv + 1 < ptr.size(); ++wv) {
    for (int j = ptr[wv]; j < ptr[wv + 1]; ++j) tot[wv] += cost_of(list[j]);
    longest = std::max(longest, tot[wv]);
  }
  for (size_t wv = 0; wv + 1 < ptr.size(); ++wv) {
    double rem = tot[wv];
    for (int j = ptr[wv]; j < ptr[wv + 1]; ++j) {
      prio[j] = std::max(0, std::min(3, (int)std::ceil(4.0 * rem / longest) - 1));
      rem -= cost_of(list[j]);
    }
  }
  return prio;
}

// s_setprio takes an immediate
__device__ __forceinline__ void onchip_setprio(int p) {
  switch (p & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
}

// ------------------------------------------------------------------ index maps (shared)
struct RateMatch {
  int k, n, z, k_ldpc, n_vn, m_int;
  // q = n / m_int (rows of the output interleaver) and ceil(2^30 / q): t / q = umulhi(4 t, q_magic) exactly for t, q < 2^15
  // (t (q_magic q - 2^30) < t q < 2^30) - the two integer divisions of rate recovery cost ~80 instructions per LLR, a third
  // of the fixed cost per codeword of the on-chip decoders.  0: divide.
  int q = 0;
  unsigned q_magic = 0;
};
inline RateMatch make_rate_match(const samd_ldpc5g* h) {
  RateMatch rm{h->k, h->n, h->z, h->k_ldpc, h->n_vn, h->m_int};
  if (h->m_int > 0 && h->n < (1 << 15)) {
    rm.q = h->n / h->m_int;
    rm.q_magic = (unsigned)(((1ull << 30) + (unsigned long long)rm.q - 1ull) / (unsigned long long)rm.q);
  }
  return rm;
}

// position t of c_short / x_short (before the output interleaver) for output index o
// (encoding.py:238-244: out[o] = c_short[perm[o]], perm[i + j*m] = i*(n/m) + j)
__device__ __forceinline__ int out_to_short(const RateMatch& p, int o) {
  if (p.m_int <= 0) return o;
  const int i = o % p.m_int, j = o / p.m_int;
  return i * (p.n / p.m_int) + j;
}
// index into the full (filler-including) codeword for position t of c_short
// (encoding.py:645-655 / decoding.py:1508-1521)
__device__ __forceinline__ int short_to_full(const RateMatch& p, int t) {
  const int u = t + 2 * p.z;
  return u < p.k ? u : u + (p.k_ldpc - p.k);
}
// rate recovery (decoding.py:1438-1475): value for VN v given the received llr row
__device__ __forceinline__ float recover_llr(const RateMatch& p, const float* __restrict__ llr_row, int v,
                                             float llr_max) {
  int u;
  if (v < p.k) u = v;
  else if (v < p.k_ldpc) return -llr_max;          // filler bits: logit -llr_max
  else u = v - (p.k_ldpc - p.k);
  const int t = u - 2 * p.z;
  if (t < 0 || t >= p.n) return 0.f;               // punctured
  int o = t;
  if (p.m_int > 0) {                               // out_int_inv[t]
    if (p.q_magic) {
      const int tq = (int)__umulhi((unsigned)t << 2, p.q_magic);
      o = tq + (t - tq * p.q) * p.m_int;
    } else {
      const int q = p.n / p.m_int;
      o = (t / q) + (t % q) * p.m_int;
    }
  }
  return llr_row[o];
}


}  // namespace samd
