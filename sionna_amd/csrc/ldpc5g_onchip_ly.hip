// On-chip LAYERED decoder for 5G-NR LDPC codes (round 3).
//
// Replaces LDPC5GDecoder(cn_schedule="layered").call (reference src/sionna/phy/fec/ldpc/decoding.py:1383-1389: one
// sub-iteration per base row = Z check nodes; _bp_iter with an array schedule :463-520: check-node update of the layer,
// then the variable-node update) for the codes without a partially pruned base row whose messages fit in LDS (any lifting
// size; a chunk = 64 lifted copies, the last one partly filled when Z is not a multiple of 64).  Until round 3 the schedule ran on the HBM-resident engine with two
// launches per layer: 920 launches and 79 k decodes/s for 10 iterations at config C2.
//
// State of one codeword (one workgroup, 16 waves):
//   LDS   c2v[e][z]   one float per edge of the NON-FUSED columns, block e = (row, position), indexed by the check node's
//                     lifted copy - the message a check node SENT last; it stays until the row's next update
//         xtot[c][z]  the unclipped total of every variable node of a non-fused column (sum of its c2v in ascending
//                     check-node order, channel LLR last - vn_update_sum, decoding.py:681-732)
//   VGPRs of the wave that OWNS an item (the item -> wave map is the same in every iteration):
//         check-node item (row, chunk) with a fused edge: the c2v of that edge (the degree-1 column of the base graph's
//                     extension part, private to its row) and the channel LLR of its variable node
//         variable-node unit (column, chunk): the channel LLRs and the PREFIX of the running total - the sum of the
//                     column's messages whose rows were already updated in this iteration, in check-node order
//   L2 workspace row  channel LLRs of all columns (read once per codeword into the owners' registers, and by the
//                     output phase) and the final c2v of the fused edges (written once for the output phase)
// v2c messages are never stored: v2c_e = clip(xtot[v] - c2v_e) is what the reference's variable-node update leaves on an
// edge (decoding.py:724-731), recomputed when the row is updated.  After a layer, the reference updates EVERY variable
// node; only the nodes of the layer's columns see a changed input, so re-summing those - all their edges, in the defined
// order - gives the same bits (min-sum: bit-identical to oracle/ldpc_bp.py's literal form; boxplus rules: same function).
// Rows are updated in ascending order, so the messages of a column's edges ABOVE the updated row are final for this
// iteration: their left-to-right sum (the prefix) is kept, and a re-sum is prefix + new message + the edges below + LLR -
// the same additions in the same order as summing everything again, half the LDS reads on average.
// Consecutive base rows that share no column are one group (BG1: 32 groups instead of 46 layers): their updates commute.
//
// Every wave walks a linear record list built on the host.  A step = the records between two workgroup barriers: the
// check-node items of a group (one 64-lane chunk of a row per wave; for the boxplus rules cut into parts of 2 or 4 edges
// on different waves, ly_cns_part), then the re-sums the next group waits for (column; pairs of chunks for small degrees);
// re-sums nobody waits for yet fill idle waves of later steps.  No global memory operand inside the iteration loop:
// until round 3x the fused state and the channel LLRs came from the L2 workspace one record ahead, and every record then
// lasted at least one L2 round trip (~1.6 k cycles measured per item, profiles/r03b/ly_itrace_r03x_before.txt, against
// ~30 cycles per edge of actual work).
#include <array>
#include "ldpc5g_onchip_ms.inc"

namespace samd {

// A record = one body (or none) followed by `nb` workgroup barriers.  Word x: bits 0-1 body kind, 2-7 body key (CN: row
// degree | fused << 5; VN: reads / 4 | pair << 4), 8-13 nb, 14 VN `first`, 15 last record of the list, 16-23 chunk;
// y: LDS byte offset (row block / xtot block); z: byte offset of the body's (byte offset, 4 shift) entries in the entry
// table; w: where the item's private state sits in the wave's register file.  One record per step and wave instead of
// separate item and barrier records: a lone wave issues an instruction every 6-7 cycles (the other 15 wait at the
// barrier), so a record costs what its instruction count says - the walker's share is ~60 instructions.
enum { LY_NOP = 0, LY_CNS = 1, LY_CN = 2, LY_VN = 3, LY_LAST = 1 << 15, LY_MAX_NB = 63 };

#ifdef SAMD_LY_TRACE
// Development aid (tools/ly_itrace.py, `make -C sionna_amd/csrc lytrace`; not part of the product build): iteration 3 of
// workgroup 0, per wave one (record word, s_memtime) pair at the start of every record and one at the end of the list.
__device__ unsigned long long* g_ly_trace = nullptr;
__device__ unsigned long long* g_ly_sub = nullptr;      // 4 times inside the item that precedes a 0xF0 record
#define SAMD_LY_MARK(tag)                                                                                  \
  if (g_ly_trace && blockIdx.x == 0 && it == 3 && lane == 0 && tr_slot < 512) {                              \
    g_ly_trace[(w * 512 + tr_slot) * 2] = (unsigned long long)(unsigned)(tag);                               \
    g_ly_trace[(w * 512 + tr_slot) * 2 + 1] = __builtin_readcyclecounter();                                  \
    ++tr_slot;                                                                                             \
  }
#define SAMD_LY_SUB(i) tm[i] = __builtin_readcyclecounter();
#else
#define SAMD_LY_MARK(tag)
#define SAMD_LY_SUB(i)
#endif

// per-wave register file of the items' private state: [0, 8) c2v of the fused edge of the wave's CN items, [8, 16) channel
// LLR of their degree-1 nodes, [16, 24) prefixes of the wave's VN units, [24, 32) their channel LLRs.  One 32-wide vector:
// a dynamic (wave-uniform) index becomes s_set_gpr_idx / v_mov, 3 instructions, where four 8-wide vectors became chains
// of 8 compares + 8 v_cndmask per access (~70 instructions in front of every re-sum body).
typedef float ly_f32x32 __attribute__((ext_vector_type(32)));
enum { LY_CN_SLOTS = 8, LY_VN_SLOTS = 8, LY_SLOT_INTS = 2 * LY_CN_SLOTS + LY_VN_SLOTS };
enum { LY_ST_CO = 0, LY_ST_LF = 8, LY_ST_P = 16, LY_ST_L = 24 };

// One record of a wave's list = 4 dwords (x, LDS byte offset, offset of the body's entries in the entry table, extra);
// the (byte offset, 4 shift) entries of a body come from ONE table shared by all records: a row's list, or a column's
// list from edge j on.  Records (~450 x 16 B per iteration at C2) + table (~5 KB) stay in the 16 KB scalar data cache.
// (Tried and dropped, profiles/r03b: records with their entries inline - one 64-byte line per record, 110 KB per
// iteration, every record load a scalar-cache miss that the body's first LDS wait, lgkmcnt(0), has to sit out: 555 k
// decodes/s against 635 k, layered_abl_r03zb/zc.txt; the first entries of the next record requested a record ahead:
// 12 more scalar moves per record, 596 k against 657 k, layered_abl_r03zg.txt; the re-sums of columns 0 / 1 run by the
// check-node wave itself behind its item, which saves the re-sum step but puts three dependent records on one wave:
// 497 k, layered_abl_r03zc.txt.)
#define LY_ENT(i, k) ent[2 * (i) + (k)]

// check nodes (row r, lifted copies 64 chunk + lane): D edges, the last one fused when F.
// entries: (byte offset of the column's xtot block, 4 shift) per non-fused edge; a0 = row block byte offset + 4 (64 chunk + lane)
// co: c2v of the fused edge (in: the one sent last, out: the new one), lf: channel LLR of the fused degree-1 node
template <int D, bool F, bool POW2, int MODE>
__device__ __forceinline__ void ly_cn_row(unsigned a0, unsigned z4, const int32_t* __restrict__ ent, unsigned lane4, unsigned zwv, float llr_max, float offset, float& co, float lf,
                                          [[maybe_unused]] unsigned long long* tm) {
  constexpr int NF = F ? D - 1 : D;
  float v[1][D];
  SAMD_LY_SUB(0)
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const unsigned t = lane4 + (unsigned)LY_ENT(i, 1);
    const unsigned ax = POW2 ? ((t & zwv) | (unsigned)LY_ENT(i, 0)) : (min(t, t - zwv) + (unsigned)LY_ENT(i, 0));
    const float x = lds_ld(ax);
    const float c = lds_ld(a0 + (unsigned)i * z4);
    v[0][i] = ms_med3(x - c, -llr_max, llr_max);                       // the v2c the last variable-node update left
  }
  if constexpr (F) {
    const float x = co + lf;                                            // (0 + c2v) + llr of the degree-1 node
    v[0][D - 1] = ms_med3(x - co, -llr_max, llr_max);
  }
  SAMD_LY_SUB(1)
  if constexpr (MODE == SAMD_CN_MINSUM) {
    ms_minsum_inplace<D, 1, 1>(v, llr_max, offset);
  } else {
    // boxplus rules: bp_math.h's node update on the D messages of a chunk - the function of every other engine
    cn_update_col<MODE, D>(v[0], D, llr_max, 0.f);
  }
  SAMD_LY_SUB(2)
#pragma unroll
  for (int i = 0; i < NF; ++i) lds_st(a0 + (unsigned)i * z4, v[0][i]);
  if constexpr (F) co = v[0][D - 1];
  SAMD_LY_SUB(3)
}

// PART of a check-node item for the boxplus rules: E consecutive edges (from edge e0) of the D edges of (row, chunk), the
// last one fused when F.  A step works one or two waves and a boxplus-phi edge costs ~100 instructions (two exp, four
// log), so the edges of a row are spread over several waves: every part computes phi of its own incoming messages and
// leaves them (sign of the message in the sign bit - phi is positive) in a scratch block, one workgroup barrier, then
// every part sums ALL D values in edge order - the order of cn_update_col, so the same bits - and finishes its own edges.
// The other waves of the workgroup pass the same barrier from their record lists.  scr = byte address of the item's
// scratch slot 0 for this lane; zero_a = address of a zero for this lane (reads past D).
template <int E, bool F, bool POW2, int MODE>
__device__ __forceinline__ void ly_cns_part(unsigned a0, unsigned z4, const int32_t* __restrict__ ent, unsigned lane4,
                                            unsigned zwv, float llr_max, float& co, float lf, unsigned scr, int e0, int D,
                                            unsigned zero_a) {
  constexpr int NF = F ? E - 1 : E;
  float v[E];
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    const unsigned t = lane4 + (unsigned)ent[2 * i + 1];
    const unsigned ax = POW2 ? ((t & zwv) | (unsigned)ent[2 * i]) : (min(t, t - zwv) + (unsigned)ent[2 * i]);
    const float x = lds_ld(ax);
    const float c = lds_ld(a0 + (unsigned)i * z4);
    v[i] = ms_med3(x - c, -llr_max, llr_max);
  }
  if constexpr (F) {
    const float x = co + lf;
    v[E - 1] = ms_med3(x - co, -llr_max, llr_max);
  }
  unsigned sg[E];
#pragma unroll
  for (int i = 0; i < E; i += 2) {
    if (i + 1 < E) {
      sg[i] = (v[i] < 0.f) ? 0x80000000u : 0u;
      sg[i + 1] = (v[i + 1] < 0.f) ? 0x80000000u : 0u;
      const f32x2 ph = phi2_f32<MODE>(fabsf(v[i]), fabsf(v[i + 1]));
      v[i] = ph.x; v[i + 1] = ph.y;
    } else {
      sg[i] = (v[i] < 0.f) ? 0x80000000u : 0u;
      v[i] = phi1_f32<MODE>(fabsf(v[i]));
    }
  }
#pragma unroll
  for (int i = 0; i < E; ++i) lds_st(scr + (unsigned)(e0 + i) * 256u, __uint_as_float(__float_as_uint(v[i]) | sg[i]));
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  float sum = 0.f;
  unsigned node = 0u;
  for (int b = 0; b < D; b += 4) {                                       // (wave-uniform trip count)
    float w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = lds_ld((b + q < D) ? scr + (unsigned)(b + q) * 256u : zero_a);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned u = __float_as_uint(w[q]);
      node ^= u & 0x80000000u;
      sum += __uint_as_float(u & 0x7FFFFFFFu);                           // (reads past D add +0: exact)
    }
  }
#pragma unroll
  for (int i = 0; i < E; i += 2) {
    if (i + 1 < E) {
      const f32x2 q = phi2_f32<MODE>(-1.f * v[i] + sum, -1.f * v[i + 1] + sum);
      v[i] = __uint_as_float(__float_as_uint(fminf(q.x, llr_max)) ^ (sg[i] ^ node));
      v[i + 1] = __uint_as_float(__float_as_uint(fminf(q.y, llr_max)) ^ (sg[i + 1] ^ node));
    } else {
      const float q = phi1_f32<MODE>(-1.f * v[i] + sum);
      v[i] = __uint_as_float(__float_as_uint(fminf(q, llr_max)) ^ (sg[i] ^ node));
    }
  }
#pragma unroll
  for (int i = 0; i < NF; ++i) lds_st(a0 + (unsigned)i * z4, v[i]);
  if constexpr (F) co = v[E - 1];
}

// variable nodes of column c (lifted copies of chunk(s)) after the update of the row of its edge j: 1 + the number of
// edges below j reads, padded to NL (a multiple of 4) with entries that point at a block of zeros (x + 0 = x exactly,
// and no total is ever -0) - few bodies, hence a shallow dispatch tree.  entries: (edge block byte offset, 4 shift) of
// edge j and the edges below it, rows ascending; ax = byte address of xtot[c][64 chunk + lane]; p = prefix (sum of the
// edges above j; 0 when j is the column's first edge), l = channel LLR (added last)
template <int NL, int NCH, bool POW2>
__device__ __forceinline__ void ly_vn_col(const int32_t* __restrict__ ent, unsigned zz4, unsigned zwv, unsigned ax, float& p0, float& p1, float l0, float l1,
                                          [[maybe_unused]] unsigned long long* tm) {
  float c[NCH][NL];
  SAMD_LY_SUB(0)
#pragma unroll
  for (int i = 0; i < NL; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const unsigned t = zz4 + 256u * h - (unsigned)LY_ENT(i, 1);
      const unsigned a = POW2 ? ((t & zwv) | (unsigned)LY_ENT(i, 0)) : (min(t, t + zwv) + (unsigned)LY_ENT(i, 0));
      c[h][i] = lds_ld(a);
    }
  SAMD_LY_SUB(1)
  if constexpr (NCH == 2) {
    ms_f32x2 xv = ms_f32x2{p0, p1} + ms_f32x2{c[0][0], c[1][0]};
    p0 = xv.x;
    p1 = xv.y;
#pragma unroll
    for (int i = 1; i < NL; ++i) xv += ms_f32x2{c[0][i], c[1][i]};
    xv += ms_f32x2{l0, l1};
    lds_st(ax, xv.x);
    lds_st(ax + 256u, xv.y);
  } else {
    float x = p0 + c[0][0];
    p0 = x;
#pragma unroll
    for (int i = 1; i < NL; ++i) x += c[0][i];
    x += l0;
    lds_st(ax, x);
  }
  SAMD_LY_SUB(3)
}

// PART: the lifting size is not a multiple of 64 - the last 64-lane chunk of a row / column is partly filled, and the
// bodies of its items run under `lane < Z - 64 chunk` (a template flag: the full-chunk codes, C2 among them, keep a
// walker without the exec mask).
template <bool POW2, int MODE, bool PART>
__global__ __launch_bounds__(1024) void ldpc5g_decode_ly_kernel(
    const float* __restrict__ llr_in, float* __restrict__ out, float* __restrict__ ws, RateMatch p, int nbu, int batch,
    int num_iter, float llr_max, float offset, int hard_out, int return_infobits, int msg_floats, int n_ext,
    int zero_off, int nt, const int32_t* __restrict__ rec_ptr, const int4* __restrict__ recs, const int32_t* __restrict__ ent_tab,
    const int32_t* __restrict__ xt_index, const int32_t* __restrict__ slot_tab) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((unsigned)(size_t)(lds_f32*)smem != 0u) __builtin_trap();      // LDS addressed by plain byte offsets (lds_ld)
  const int NT = nt;                                                   // threads per codeword: 16, 8 or 4 waves
  const unsigned z = (unsigned)p.z, z4 = 4u * z;
  const unsigned zw = POW2 ? z4 - 1u : z4;
  unsigned zwv;
  asm volatile("v_mov_b32 %0, %1" : "=v"(zwv) : "s"(zw));
  const int nx = nbu * (int)z;
  float* llr = ws + (size_t)blockIdx.x * (size_t)(nx + n_ext * (int)z);
  float* cext = llr + nx;
  float* xtot = smem + msg_floats;
  const int n_xt = zero_off - msg_floats;                              // floats of the xtot blocks
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned lane4 = 4u * (unsigned)lane;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = rec_ptr[w];
  const int32_t* slots = slot_tab + w * LY_SLOT_INTS;

  for (int b = blockIdx.x; b < batch; b += gridDim.x) {
    const float* row = llr_in + (size_t)b * p.n;
    // decoding.py:552-565 (clip, logits -> LLR; + 0.f: no -0 anywhere), messages start at 0 (:575-578)
    // A channel LLR is kept where it is used: a non-fused column's in its xtot block (LDS; its owner copies it into
    // registers below), a fused degree-1 node's in its owner's registers (read from the input row below).  Only the
    // codeword output (return_infobits = 0) needs the LLRs of the columns without an xtot block once more, and only then
    // are they written to the workspace row: until round 4 every LLR went through it (and every final fused c2v), 45 KB
    // per codeword written and read back through L2 - 6.7 GB per launch at C2 on the memory-side counters against
    // 2.95 GB of compulsory input + output (profiles/r03zy_pmc).
    auto chan_llr = [&](int v) __attribute__((always_inline)) -> float {
      return (v < p.n_vn) ? (-1.f * clampf(recover_llr(p, row, v, llr_max), -llr_max, llr_max)) + 0.f : 0.f;
    };
    for (int v = tid; v < nx; v += NT) {
      const float l = chan_llr(v);
      const int xi = xt_index[v / (int)z];
      if (xi >= 0) xtot[xi * (int)z + v % (int)z] = l;                 // total of a node without messages = its LLR
      else if (!return_infobits) llr[v] = l;
    }
    // the registers of the items this wave owns: channel LLRs; c2v of the fused edges and the prefixes start at 0.
    // The fused nodes' LLRs come straight from the input row: requested BEFORE the LDS initialisation and its barrier, so
    // that their latency is hidden behind them
    ly_f32x32 st;
#pragma unroll
    for (int q = 0; q < LY_CN_SLOTS; ++q) {
      const int li = slots[q];
      st[LY_ST_CO + q] = 0.f;
      st[LY_ST_LF + q] = (li >= 0) ? chan_llr(min(li + lane, nx - 1)) : 0.f;   // (lanes past a partial chunk: never used)
    }
    for (int i = tid; i < msg_floats; i += NT) smem[i] = 0.f;
    for (int i = tid; i < (int)z; i += NT) smem[zero_off + i] = 0.f;   // the block the padded re-sum entries read
    __syncthreads();
#pragma unroll
    for (int q = 0; q < LY_VN_SLOTS; ++q) {
      const int li = slots[2 * LY_CN_SLOTS + q];
      st[LY_ST_P + q] = 0.f;
      st[LY_ST_L + q] = (li >= 0) ? xtot[min(li + lane, n_xt - 1)] : 0.f;   // li: float index into xtot (host: xt block * Z + 64 chunk)
    }

    // Records are wave-uniform: scalar loads, one record ahead.
    for (int it = 0; it < num_iter; ++it) {
      const int4* rp = recs + r0;
      [[maybe_unused]] int tr_slot = 0;
      int4 cur = rp[0];
      for (;;) {
        const int cx = cur.x, cy = cur.y;
        SAMD_LY_MARK(cx)
        const int4 nxt = rp[1];                                          // (a list ends with one spare record)
        const int32_t* ent = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(ent_tab) + cur.z);
        const unsigned zq4 = ((unsigned)(cx >> 16) & 0xFFu) * 256u + lane4;      // 4 (64 chunk + lane)
        [[maybe_unused]] unsigned long long tm[4] = {0, 0, 0, 0};
        // The body's private state: four indexed reads in front of the dispatch, two indexed writes behind it, in ONE
        // place each (reads / writes inside the branches made the compiler copy the 32 registers once per path).
        // VN: s0, s1 = prefixes of the unit's chunk(s), s2, s3 = their channel LLRs; CN: s0 = c2v of the fused edge,
        // s2 = its channel LLR.  A body that does not use a value leaves it as read, so writing it back is harmless.
        // (w = the indices i0 | i1 << 8 of s0 / s1 in the register file; s2 / s3 are 8 above)
        const bool act = !PART || (zq4 >> 2) < z;                        // lifted copy 64 chunk + lane exists
        const bool is_vn = (cx & 3) == LY_VN;
        const int i0 = cur.w & 31, i1 = (cur.w >> 8) & 31;
        float s0 = st[i0], s1 = st[i1], s2 = st[i0 + 8], s3 = st[i1 + 8];
        if (!act) {
          // (a lane past the lifting size: no body; the barrier of a split check-node part is executed by the wave's
          // active lanes)
        } else if (is_vn) {
          const unsigned ax = (unsigned)cy + zq4;
          if ((cx >> 14) & 1) {                                          // the column's first edge: the prefix restarts
            // (a branch on the wave-uniform record - the empty asm keeps it one: as selections the two resets read a vcc
            // written by s_cselect_b64, ~24 cycles of the vector pipe each on gfx950, profiles/r06w_valu_rate2.txt)
            asm volatile("");
            s0 = 0.f;
            if ((cx >> 6) & 1) { asm volatile(""); s1 = 0.f; }
          }
#define SAMD_LY_VN(KEY, NL, NCHV) case KEY: ly_vn_col<NL, NCHV, POW2>(ent, zq4, zwv, ax, s0, s1, s2, s3, tm); break;
          switch ((cx >> 2) & 63) {
            SAMD_LY_VN(1, 4, 1) SAMD_LY_VN(2, 8, 1) SAMD_LY_VN(3, 12, 1) SAMD_LY_VN(4, 16, 1) SAMD_LY_VN(5, 20, 1)
            SAMD_LY_VN(6, 24, 1) SAMD_LY_VN(7, 28, 1) SAMD_LY_VN(8, 32, 1)
            SAMD_LY_VN(17, 4, 2) SAMD_LY_VN(18, 8, 2) SAMD_LY_VN(19, 12, 2)
            default: break;
          }
#undef SAMD_LY_VN
        } else if ((cx & 3) == LY_CNS) {
          if constexpr (MODE != SAMD_CN_MINSUM) {
            // w: bits 16-23 D, 24-31 first edge; x bits 24-31: the item's scratch area, in 256-byte slots
            const unsigned scr = 4u * (unsigned)zero_off + z4 + 256u * ((unsigned)cx >> 24) + lane4;
            const unsigned zero_a = 4u * (unsigned)zero_off + lane4;
            const int dd = (cur.w >> 16) & 255, e0 = (cur.w >> 24) & 255;
#define SAMD_LY_CNS(KEY, E, F) \
  case KEY: ly_cns_part<E, F, POW2, MODE>((unsigned)cy + zq4, z4, ent, zq4, zwv, llr_max, s0, s2, scr, e0, dd, zero_a); break;
            switch ((cx >> 2) & 63) {
              SAMD_LY_CNS(1, 1, false) SAMD_LY_CNS(2, 2, false) SAMD_LY_CNS(3, 3, false) SAMD_LY_CNS(4, 4, false)
              SAMD_LY_CNS(9, 1, true) SAMD_LY_CNS(10, 2, true) SAMD_LY_CNS(11, 3, true) SAMD_LY_CNS(12, 4, true)
              default: break;
            }
#undef SAMD_LY_CNS
          }
        } else if ((cx & 3) == LY_CN) {
#define SAMD_LY_CN(KEY, D, F) \
  case KEY: ly_cn_row<D, F, POW2, MODE>((unsigned)cy + zq4, z4, ent, zq4, zwv, llr_max, offset, s0, s2, tm); break;
          switch ((cx >> 2) & 63) {
            SAMD_LY_CN(3, 3, false) SAMD_LY_CN(4, 4, false) SAMD_LY_CN(5, 5, false) SAMD_LY_CN(6, 6, false)
            SAMD_LY_CN(7, 7, false) SAMD_LY_CN(8, 8, false) SAMD_LY_CN(9, 9, false) SAMD_LY_CN(10, 10, false)
            SAMD_LY_CN(19, 19, false)
            SAMD_LY_CN(35, 3, true) SAMD_LY_CN(36, 4, true) SAMD_LY_CN(37, 5, true) SAMD_LY_CN(38, 6, true)
            SAMD_LY_CN(39, 7, true) SAMD_LY_CN(40, 8, true) SAMD_LY_CN(41, 9, true) SAMD_LY_CN(42, 10, true)
            default: break;
          }
#undef SAMD_LY_CN
        }
        st[i1] = s1;
        st[i0] = s0;
#ifdef SAMD_LY_TRACE
        if ((cx & 3) >= LY_CN) {
          SAMD_LY_MARK(0xF0)                                             // end of the body; its inner times beside it
          if (g_ly_trace && blockIdx.x == 0 && it == 3 && lane == 0 && tr_slot <= 512)
            for (int q = 0; q < 4; ++q) g_ly_sub[(w * 512 + tr_slot - 1) * 4 + q] = tm[q];
        }
#endif
        // LDS traffic only: wait for this wave's LDS operations, then the barrier(s) (several: steps without a body)
        for (int q = (cx >> 8) & 63; q > 0; --q) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (cx & LY_LAST) break;
        ++rp;
        cur = nxt;
      }
    }
    // the final c2v of the fused edges, for the output phase
#pragma unroll
    for (int q = 0; q < LY_CN_SLOTS; ++q) {
      const int ci = slots[LY_CN_SLOTS + q];
      if (!return_infobits && ci >= 0 && (!PART || ci % (int)z + lane < (int)z)) cext[ci + lane] = st[LY_ST_CO + q];
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
  const int z = h->z;
  const int ncu = (h->n_cn + z - 1) / z, nbu = (h->n_vn + z - 1) / z;
  // (lifting sizes below 16 stay on the HBM-resident engine, which runs many codewords side by side: Z = 11 is a tie,
  // 7.8 against 8.1 M decodes/s; from Z = 18 on the on-chip engine with 4 waves per codeword wins, 5.2 against 3.9 M -
  // layered_rate_smallz_r03z.txt)
  if (h->n_cn % z != 0 || h->n_vn % z != 0 || h->mb > 255 || h->nb > 255 || nbu > 0xFFFF ||
      z < ((int)opt_int("SAMD_LY_MINZ", 16)) || z > 64 * 255) return SAMD_OK;
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
  // groups of consecutive rows that share no non-fused column
  std::vector<std::vector<int>> groups;
  {
    std::vector<char> used(h->nb, 0);
    std::vector<int> cur;
    for (int r = 0; r < ncu; ++r) {
      bool clash = false;
      for (auto& e : by_row[r]) clash = clash || (!col_fused[e.first] && used[e.first]);
      if ((clash && !cur.empty()) || opt_set("SAMD_LY_NOGROUP")) {
        if (!cur.empty()) groups.push_back(cur);
        cur.clear();
        std::fill(used.begin(), used.end(), 0);
      }
      cur.push_back(r);
      for (auto& e : by_row[r]) used[e.first] = 1;
    }
    if (!cur.empty()) groups.push_back(cur);
  }
  const int G = (int)groups.size();
  std::vector<std::vector<char>> in_group(G, std::vector<char>(h->nb, 0));
  for (int g = 0; g < G; ++g)
    for (int r : groups[g])
      for (auto& e : by_row[r])
        if (!col_fused[e.first]) in_group[g][e.first] = 1;
  // LDS layout: edge blocks of the non-fused edges, row-major; xtot of the non-fused columns that have edges; one block of
  // zeros (padding of the re-sums)
  std::vector<int> row_off(ncu, 0), xt_of_col(h->nb, -0x7FFFFFFF);
  int edges = 0, ncore = 0;
  for (int r = 0; r < ncu; ++r) { row_off[r] = edges * z * 4; edges += (int)by_row[r].size() - (fused_col[r] >= 0 ? 1 : 0); }
  for (int c = 0; c < nbu; ++c) {
    if (col_fused[c]) continue;
    if (col_deg[c] == 0) continue;
    if (col_deg[c] > 30) return SAMD_OK;
    xt_of_col[c] = ncore++;
  }
  const size_t lds = ((size_t)edges + (size_t)ncore + 1) * z * 4;
  if (lds > 160 * 1024) return SAMD_OK;
  const int xt_base = edges * z * 4, zero_base = (edges + ncore) * z * 4;
  std::vector<int32_t> xt_index(h->nb, -0x7FFFFFFF);
  for (int c = 0; c < nbu; ++c) xt_index[c] = xt_of_col[c];
  for (int r = 0; r < ncu; ++r)
    if (fused_col[r] >= 0) xt_index[fused_col[r]] = -1 - ext_of_row[r];
  // per-row entries (xtot byte offset of the column, 4 shift) and per-column entries (edge block byte offset, 4 shift)
  std::vector<std::vector<int32_t>> row_ent(ncu);
  std::vector<std::vector<std::pair<int, int>>> col_edges(h->nb);       // (block byte offset, 4 shift), rows ascending
  for (int r = 0; r < ncu; ++r) {
    const int nf = (int)by_row[r].size() - (fused_col[r] >= 0 ? 1 : 0);
    for (int i = 0; i < nf; ++i) {
      const int c = by_row[r][i].first, s = by_row[r][i].second;
      if (xt_of_col[c] < 0) return SAMD_OK;
      row_ent[r].push_back(xt_base + xt_of_col[c] * z * 4);
      row_ent[r].push_back(4 * s);
      col_edges[c].push_back({row_off[r] + i * z * 4, 4 * s});
    }
  }
  // the entry table: the rows' lists, then the columns' lists, each column's followed by three reads of the zero block (a
  // re-sum reads from edge j on, rounded up to a multiple of 4 entries)
  std::vector<int32_t> ent_tab, row_start(ncu, 0), col_start(h->nb, 0);
  for (int r = 0; r < ncu; ++r) {
    row_start[r] = (int32_t)ent_tab.size();
    ent_tab.insert(ent_tab.end(), row_ent[r].begin(), row_ent[r].end());
  }
  for (int c = 0; c < nbu; ++c) {
    col_start[c] = (int32_t)ent_tab.size();
    for (auto& e : col_edges[c]) { ent_tab.push_back(e.first); ent_tab.push_back(e.second); }
    for (int q = 0; q < 3; ++q) { ent_tab.push_back(zero_base); ent_tab.push_back(0); }
  }
  ent_tab.resize(ent_tab.size() + 64, 0);
  const int chunks = (z + 63) / 64;                          // (the last one partly filled when Z is not a multiple of 64)
  struct Lists { std::vector<int32_t> rec_ptr, recs, slot_tab; int scratch_bytes = 0; };
  // SAMD_LY_ABL (development, wrong results): 1 = lists without the re-sum items, 2 = without the CN items, 3 = barriers only
#ifdef SAMD_DEV
  const int abl = (int)opt_int("SAMD_LY_ABL", 0);
#else
  const int abl = 0;                                         // wrong results by construction: development builds only
#endif
  // The lists are built twice: whole check-node items (min-sum: ~13 instructions per edge), and - `split` - items cut into
  // parts of 2 (row degree <= 10) or 4 edges for the boxplus rules (~100 instructions per edge), see ly_cns_part.
  // Returns 1 when the code does not fit the engine.
  auto build_lists = [&](bool split, int NW, Lists& L) -> int {
  L = Lists();
  // ---- ownership.  A check-node item (row, chunk) - or each of its parts - and a variable-node unit (column, chunk or pair
  // of chunks) run on the same wave in every iteration, which keeps their private state in that wave's registers.
  // CN items: the waves of a group's step are distinct; among those the one with the fewest fused slots, then the
  // least check-node work so far
  struct Part { int r, q, e0, ne, wave, scr; };                             // edges [e0, e0 + ne) of (row, chunk)
  std::vector<std::vector<Part>> cn_of_group(groups.size());
  std::vector<char> group_split(groups.size(), 0);
  std::vector<int> cn_slots(NW, 0), cn_work(NW, 0);
  std::vector<std::vector<int>> cn_slot_llr(NW), cn_slot_ext(NW);
  std::vector<std::vector<int>> cn_slot_of(ncu, std::vector<int>(chunks, 0));
  const int free_lds = 160 * 1024 - (int)lds;
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    // parts of the group's items; the group is split only if its parts fit the 16 waves and its scratch the free LDS
    std::vector<Part> parts;
    int scr = 0;
    for (int r : groups[gi])
      for (int q = 0; q < chunks; ++q) {
        const int d = (int)by_row[r].size(), pe = d <= 10 ? 2 : 4;
        for (int e0 = 0; e0 < d; e0 += pe) parts.push_back({r, q, e0, std::min(pe, d - e0), -1, scr});
        scr += d * 256;
      }
    if (split && (int)parts.size() <= NW && scr <= free_lds) {
      group_split[gi] = 1;
      L.scratch_bytes = std::max(L.scratch_bytes, scr);
    } else {
      parts.clear();
      for (int r : groups[gi])
        for (int q = 0; q < chunks; ++q) parts.push_back({r, q, 0, (int)by_row[r].size(), -1, 0});
    }
    std::vector<char> taken(NW, 0);
    int ntaken = 0;
    for (auto& pt : parts) {
      if (ntaken == NW) { std::fill(taken.begin(), taken.end(), 0); ntaken = 0; }
      const int f = fused_col[pt.r] >= 0 && pt.e0 + pt.ne == (int)by_row[pt.r].size();   // the part with the fused edge
      int best = -1;
      for (int k2 = 0; k2 < NW; ++k2) {
        const int wv = (int)((gi + (size_t)k2) % NW);
        if (taken[wv] || (f && cn_slots[wv] >= LY_CN_SLOTS)) continue;
        if (best < 0 || std::make_pair(f ? cn_slots[wv] : 0, cn_work[wv]) < std::make_pair(f ? cn_slots[best] : 0, cn_work[best])) best = wv;
      }
      if (best < 0) return 1;                                             // more fused items than slots
      taken[best] = 1; ++ntaken;
      cn_work[best] += 10 + pt.ne;
      if (f) {
        cn_slot_of[pt.r][pt.q] = cn_slots[best]++;
        cn_slot_llr[best].push_back(fused_col[pt.r] * z + pt.q * 64);
        cn_slot_ext[best].push_back(ext_of_row[pt.r] * z + pt.q * 64);
      }
      pt.wave = best;
      cn_of_group[gi].push_back(pt);
    }
  }
  // ---- steps, separated by workgroup barriers: one per group for its check-node items, and behind it one for the re-sums
  // the next group waits for (the columns both groups touch: at least one, or the rows would be one group).
  std::vector<std::vector<std::pair<int, int>>> touch(h->nb);              // per column: (group, index j of the edge)
  {
    std::vector<int> next_edge(h->nb, 0);
    for (int g = 0; g < G; ++g)
      for (int r : groups[g])
        for (auto& e : by_row[r])
          if (!col_fused[e.first]) touch[e.first].push_back({g, next_edge[e.first]++});
  }
  std::vector<int> cn_step(G, 0);
  int nsteps = 0;
  for (int g = 0; g < G; ++g) {
    cn_step[g] = nsteps++;
    bool need = false;
    for (int c = 0; c < nbu; ++c)
      if (in_group[g][c] && (g + 1 == G || in_group[g + 1][c])) need = true;
    if (need || opt_set("SAMD_LY_NODEFER")) ++nsteps;
  }
  // VN units: a column's chunk (or pair of chunks for degree <= 8 / 12); its item after the update of edge j reads 1 + (the
  // edges below j) messages.  Units go, heaviest first, to the wave where they add least to the sum over the groups of
  // the squared load.
  struct Unit { int c, q, pair, wave, slot; long cost; };
  std::vector<Unit> units;
  // (cost model of the schedule, in cycles of a lone wave; development knobs for tools/ly_grid.py)
  const long ly_vn_slope = opt_int("SAMD_LY_VN_SLOPE", 18);
  const long ly_vn_ovh = opt_int("SAMD_LY_VN_OVH", 250);
  const long ly_cn_slope = opt_int("SAMD_LY_CN_SLOPE", 60);
  const long ly_cn_ovh = opt_int("SAMD_LY_CN_OVH", 300);
  auto item_cost = [&](int c, int j, int pair) { return (pair ? 2 : 1) * ly_vn_slope * ((col_deg[c] - j + 3) / 4 * 4) + ly_vn_ovh; };
  // (two-chunk lifting sizes with the split lists of the boxplus rules: 8 is +0.9 % over 12 at C2, ly_grid_r03z.txt; with
  // more chunks 12 stays - k=3000 n=6000, Z = 320, loses 2 % (boxplus-phi 7 %) with 8.  Whole-item lists under the
  // SIMD-aware schedule of round 4: 12 is +0.7 % over 8 at C2, profiles/r04b/ly_pair_max.txt - a pair halves the walker's
  // share per chunk, and the steps are bound by the SIMD's instruction total, not by the longest wave)
  const int pair_max = (int)opt_int("SAMD_LY_PAIR_MAX", (chunks == 2 && split ? 8 : 12));
  for (int c = 0; c < nbu; ++c) {
    if (xt_of_col[c] < 0) continue;
    for (int q = 0; q < chunks; ++q) {
      const int pair = (col_deg[c] <= std::min(12, pair_max) && z - 64 * (q + 1) >= 64) ? 1 : 0;   // two FULL chunks
      long cost = 0;
      for (auto& tj : touch[c]) cost += item_cost(c, tj.second, pair);
      units.push_back({c, q, pair, -1, -1, cost});
      if (pair) ++q;
    }
  }
  std::vector<int> order(units.size());
  for (size_t i = 0; i < units.size(); ++i) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return units[x].cost > units[y].cost; });
  std::vector<std::vector<long>> load(groups.size(), std::vector<long>(NW, 0));
  // (the same objective per SIMD - waves w, w + 4, ... share one issue pipe - weighted by simd_beta percent; development knob)
  const long simd_beta = opt_int("SAMD_LY_SIMD_BETA", split ? 0 : 10);
  std::vector<std::array<long, 4>> sload_g(groups.size(), std::array<long, 4>{0, 0, 0, 0});
  std::vector<int> vn_slots(NW, 0);
  std::vector<std::vector<int>> vn_slot_llr(NW);
  for (int ui : order) {
    Unit& u = units[ui];
    int best = -1;
    long best_inc = 0;
    for (int wv = 0; wv < NW; ++wv) {
      if (vn_slots[wv] + 1 + u.pair > LY_VN_SLOTS) continue;
      long inc = 0;
      for (auto& tj : touch[u.c]) {
        const long cst = item_cost(u.c, tj.second, u.pair), l0 = load[tj.first][wv], s0 = sload_g[tj.first][wv & 3];
        inc += (l0 + cst) * (l0 + cst) - l0 * l0 + simd_beta * ((s0 + cst) * (s0 + cst) - s0 * s0) / 100;
      }
      if (best < 0 || inc < best_inc) { best = wv; best_inc = inc; }
    }
    if (best < 0) return 1;                                               // more units than slots
    u.wave = best;
    u.slot = vn_slots[best];
    vn_slots[best] += 1 + u.pair;
    if (xt_of_col[u.c] < 0) return 1;                                     // a unit's column keeps its total in LDS
    vn_slot_llr[best].push_back(xt_of_col[u.c] * z + u.q * 64);          // float index into the xtot blocks
    if (u.pair) vn_slot_llr[best].push_back(xt_of_col[u.c] * z + (u.q + 1) * 64);
    for (auto& tj : touch[u.c]) {
      load[tj.first][best] += item_cost(u.c, tj.second, u.pair);
      sload_g[tj.first][best & 3] += item_cost(u.c, tj.second, u.pair);
    }
  }
  std::vector<int32_t>& slot_tab = L.slot_tab;
  slot_tab.assign((size_t)NW * LY_SLOT_INTS, -1);
  for (int wv = 0; wv < NW; ++wv) {
    for (size_t i = 0; i < cn_slot_llr[wv].size(); ++i) {
      slot_tab[(size_t)wv * LY_SLOT_INTS + i] = cn_slot_llr[wv][i];
      slot_tab[(size_t)wv * LY_SLOT_INTS + LY_CN_SLOTS + i] = cn_slot_ext[wv][i];
    }
    for (size_t i = 0; i < vn_slot_llr[wv].size(); ++i) slot_tab[(size_t)wv * LY_SLOT_INTS + 2 * LY_CN_SLOTS + i] = vn_slot_llr[wv][i];
  }
  // ---- schedule of the re-sums.  The re-sum of column c after its edge j (row in
  // group g) is needed by the next row that touches c (group g' > g) and by nothing else, and nothing it reads or writes
  // is touched by the groups in between - so it may run in any step behind group g's and before group g''s, also beside
  // the check-node items of other rows, where it fills waves that would wait at the barrier.  Items with the smallest
  // window are placed first, each into the step of its window where it lengthens the step least (a step lasts as long as
  // its busiest wave).
  struct Item { long key; int32_t x, y, zz, w; };
  std::vector<std::vector<std::vector<Item>>> step_items(nsteps, std::vector<std::vector<Item>>(NW));
  std::vector<std::vector<long>> sload(nsteps, std::vector<long>(NW, 0));
  std::vector<long> smax(nsteps, 0);
  for (int g = 0; g < G; ++g)
    for (auto& pt : cn_of_group[g]) {
      const int r = pt.r, q = pt.q, wv = pt.wave, d = (int)by_row[r].size(), st = cn_step[g];
      const int f = fused_col[r] >= 0 && pt.e0 + pt.ne == d;
      const int sidx = (LY_ST_CO + cn_slot_of[r][q]) | ((LY_ST_CO + ((cn_slot_of[r][q] + 1) & 7)) << 8);
      if (group_split[g]) {
        step_items[st][wv].push_back({1L << 40, LY_CNS | ((pt.ne | (f << 3)) << 2) | (q << 16) | (int32_t)((unsigned)(pt.scr / 256) << 24),
                                      row_off[r] + pt.e0 * z * 4, 4 * (row_start[r] + 2 * pt.e0), sidx | (d << 16) | (pt.e0 << 24)});
        sload[st][wv] += 500 + 350L * ((pt.ne + 1) / 2) + 8L * d;
      } else {
        step_items[st][wv].push_back({1L << 40, LY_CN | ((d | (f << 5)) << 2) | (q << 16), row_off[r], 4 * row_start[r], sidx});
        sload[st][wv] += split ? 300 + 330L * ((d + 1) / 2) : ly_cn_ovh + ly_cn_slope * d;
      }
      smax[st] = std::max(smax[st], sload[st][wv]);
    }
  struct VTask { int unit, j, lo, hi; long cost; };
  std::vector<VTask> vt;
  const bool defer = !(opt_int("SAMD_LY_NODEFER", 0));
  for (size_t ui = 0; ui < units.size(); ++ui) {
    const Unit& u = units[ui];
    for (size_t k2 = 0; k2 < touch[u.c].size(); ++k2) {
      const int g0 = touch[u.c][k2].first, j = touch[u.c][k2].second;
      const int hi = (k2 + 1 < touch[u.c].size()) ? cn_step[touch[u.c][k2 + 1].first] - 1 : nsteps - 1;
      if (hi < cn_step[g0] + 1) return 1;                                  // (cannot happen: such a column got a step)
      vt.push_back({(int)ui, j, cn_step[g0] + 1, defer ? hi : cn_step[g0] + 1, item_cost(u.c, j, u.pair)});
    }
  }
  std::stable_sort(vt.begin(), vt.end(), [](const VTask& x, const VTask& y) {
    return (x.hi - x.lo) != (y.hi - y.lo) ? (x.hi - x.lo) < (y.hi - y.lo) : x.cost > y.cost;
  });
  // A step lasts as long as its busiest WAVE (a lone wave issues a dependent instruction every ~6.5 cycles) - or as its
  // busiest SIMD: wave w runs on SIMD w % 4, which issues one wave64 operation per 4 cycles for all its waves together, so
  // a re-sum placed beside a check-node item of the same SIMD slows that item down (profiles/r04b/layered_owed_*.txt).
  // simd_alpha (percent; development knob, 0 = waves only) weighs the SIMD's summed load against the busiest wave's:
  // 45 (+ simd_beta 10 in the ownership objective above) is worth +3.9 % at C2 for the whole-item lists (653 -> 678 k
  // decodes/s, profiles/r04b/ly_simd_*.txt); the split lists of the boxplus rules lose 1 % with it and keep 0.
  const long simd_alpha = opt_int("SAMD_LY_SIMD_ALPHA", split ? 0 : 45);
  std::vector<std::array<long, 4>> simd(nsteps, std::array<long, 4>{0, 0, 0, 0});
  for (int st = 0; st < nsteps; ++st)
    for (int wv = 0; wv < NW; ++wv) simd[st][wv & 3] += sload[st][wv];
  auto step_cost = [&](int st) {
    long c = smax[st];
    for (int q = 0; q < 4; ++q) c = std::max(c, simd_alpha * simd[st][q] / 100);
    return c;
  };
  for (auto& tk : vt) {
    const Unit& u = units[tk.unit];
    int best = -1;
    long best_inc = 0, best_load = 0;
    for (int st = tk.lo; st <= tk.hi; ++st) {
      const long after = std::max(sload[st][u.wave] + tk.cost, simd_alpha * (simd[st][u.wave & 3] + tk.cost) / 100);
      const long inc = std::max(0L, after - step_cost(st));
      if (best < 0 || inc < best_inc || (inc == best_inc && after < best_load)) { best = st; best_inc = inc; best_load = after; }
    }
    const int nl4 = (col_deg[u.c] - tk.j + 3) / 4 * 4;
    // (sort key inside a wave's step: the check-node item first, then the re-sums in the order of their columns' edges)
    step_items[best][u.wave].push_back({((long)u.c << 8) | tk.j,
        LY_VN | (((nl4 / 4) | (u.pair << 4)) << 2) | ((tk.j == 0 ? 1 : 0) << 14) | (u.q << 16),
        xt_base + xt_of_col[u.c] * z * 4, 4 * (col_start[u.c] + 2 * tk.j), (LY_ST_P + u.slot) | ((LY_ST_P + ((u.slot + 1) & 7)) << 8)});
    sload[best][u.wave] += tk.cost;
    simd[best][u.wave & 3] += tk.cost;
    smax[best] = std::max(smax[best], sload[best][u.wave]);
  }
  if (opt_set("SAMD_LY_DUMP")) {                             // development (tools/ly_dump.py): the schedule, estimated cycles
    fprintf(stderr, "%s check-node items\n", split ? "split" : "whole");
    long total = 0;
    for (int st = 0; st < nsteps; ++st) {
      fprintf(stderr, "step %3d max %5ld |", st, smax[st]);
      for (int wv = 0; wv < NW; ++wv) {
        fprintf(stderr, " w%d:", wv);
        for (auto& it : step_items[st][wv])
          fprintf(stderr, "%s%d%s", (it.x & 3) == LY_CN ? "C" : (it.x & 3) == LY_CNS ? "P" : "V",
                  (it.x & 3) == LY_CN ? ((it.x >> 2) & 31) : (it.x & 3) == LY_CNS ? ((it.x >> 2) & 7) : 4 * ((it.x >> 2) & 15),
                  (it.x & 3) == LY_VN && ((it.x >> 6) & 1) ? "p," : ",");
      }
      fprintf(stderr, "\n");
      total += smax[st];
    }
    fprintf(stderr, "sum of step maxima %ld\n", total);
  }
  std::vector<int32_t>& rec_ptr = L.rec_ptr;
  std::vector<int32_t>& recs = L.recs;                     // 4 dwords each
  std::vector<char> split_step(nsteps, 0);
  for (int g = 0; g < G; ++g) split_step[cn_step[g]] = group_split[g];
  for (int wv = 0; wv < NW; ++wv) {
    rec_ptr.push_back((int32_t)(recs.size() / 4));
    long last = -1;                                         // position of the wave's last record
    auto push = [&](int32_t x, int32_t y, int32_t zz, int32_t w) {
      last = (long)recs.size();
      recs.insert(recs.end(), {x, y, zz, w});
    };
    auto barrier = [&]() {                                  // one more barrier behind the wave's last record
      if (last < 0 || ((recs[last] >> 8) & 63) == LY_MAX_NB) push(LY_NOP, 0, 0, 0);
      recs[last] += 1 << 8;
    };
    for (int st = 0; st < nsteps; ++st) {
      auto& its = step_items[st][wv];
      std::stable_sort(its.begin(), its.end(), [](const Item& x, const Item& y) {
        const bool cx2 = x.key >= (1L << 40), cy2 = y.key >= (1L << 40);
        return cx2 != cy2 ? cx2 : x.key < y.key;
      });
      // a step with split check-node items has a barrier in its middle (inside ly_cns_part); a wave without a part passes
      // it from its list, BEFORE its re-sums of the step (they then run beside the parts' second half)
      bool has_part = false;
      for (auto& it : its) has_part = has_part || (it.x & 3) == LY_CNS;
      if (split_step[st] && !has_part) barrier();
      for (auto& it : its) {
        const int kind = it.x & 3;
        if ((kind == LY_VN && (abl & 1)) || ((kind == LY_CN || kind == LY_CNS) && (abl & 2))) continue;
        push(it.x, it.y, it.zz, it.w);
      }
      barrier();                                            // the step's barrier
    }
    recs[last] |= LY_LAST;
    push(LY_NOP | LY_LAST, 0, 0, 0);                        // the walker reads one record ahead
  }
  rec_ptr.push_back((int32_t)(recs.size() / 4));
  // every record's key must have a compiled body in ldpc5g_decode_ly_kernel (its switches end in `default: break`, which
  // would skip the record silently and decode wrong LLRs): LY_VN re-sums of 1..8 lanes-of-4 (chunk pairs: 1..3), split
  // check-node parts of 1..4 edges (plain / fused), whole rows of degree 3..10 and 19 (plain) or 3..10 (fused)
  for (size_t t = 0; t < recs.size() / 4; ++t) {
    const int32_t x = recs[4 * t];
    const int kind = x & 3, key = (x >> 2) & 63;
    bool body = true;
    if (kind == LY_VN) body = (key >= 1 && key <= 8) || (key >= 17 && key <= 19);
    else if (kind == LY_CNS) body = (key >= 1 && key <= 4) || (key >= 9 && key <= 12);
    else if (kind == LY_CN) body = (key >= 3 && key <= 10) || key == 19 || (key >= 35 && key <= 42);
    if (!body) return 1;
  }
  // every wave must pass the same number of workgroup barriers per iteration (a split check-node part has one inside):
  // a list that does not is a deadlock on the GPU, so it is checked here and the code left to the HBM-resident engine
  long want = -1;
  for (int wv = 0; wv < NW; ++wv) {
    long nbar = 0;
    for (int32_t t = rec_ptr[wv]; t < rec_ptr[wv + 1]; ++t)
      nbar += ((recs[4 * (size_t)t] >> 8) & 63) + ((recs[4 * (size_t)t] & 3) == LY_CNS ? 1 : 0);
    if (want < 0) want = nbar;
    if (nbar != want) return 1;
  }
  return 0;
  };   // build_lists
  // Waves per codeword: 16 waves are resident per CU (128 registers each); a codeword whose state is small shares the CU
  // with others - 2 workgroups of 8 waves or 4 of 4 - instead of leaving most of 16 waves waiting at its barriers (a step
  // works one to four waves).  The footprint counts the scratch of the split check-node items.
  int max_scr = 0;
  for (auto& g : groups) {
    int scr = 0;
    for (int r : g) scr += (int)by_row[r].size() * 256 * chunks;
    max_scr = std::max(max_scr, scr);
  }
  Lists whole, parts;
  int NW = 0;
  for (int nw : {4, 8, 16}) {
    if (opt_set("SAMD_LY_WAVES") && (int)opt_int("SAMD_LY_WAVES", 0) != nw) continue;
    else if (nw < 16 && ((int)lds + max_scr) * (16 / nw) > 160 * 1024) continue;
    if (build_lists(false, nw, whole) == 0) { NW = nw; break; }
  }
  if (NW == 0) return SAMD_OK;
  const bool have_parts = abl == 0 && !opt_set("SAMD_LY_NOSPLIT") && build_lists(true, NW, parts) == 0 && parts.scratch_bytes > 0;
  h->ly_waves = NW;
  h->ly_lds_bytes = (int)lds;
  h->ly_zero_off = zero_base / 4;
  h->ly_msg_floats = edges * z;
  h->ly_n_ext = n_ext;
  h->ly_groups = (int)groups.size();
  h->ly_bp_lds_bytes = have_parts ? (int)lds + parts.scratch_bytes : 0;
  int rc = upload(&h->ly_rec_ptr, whole.rec_ptr.data(), whole.rec_ptr.size());
  if (rc == SAMD_OK) rc = upload(&h->ly_recs, whole.recs.data(), whole.recs.size());
  if (rc == SAMD_OK && have_parts) rc = upload(&h->ly_bp_rec_ptr, parts.rec_ptr.data(), parts.rec_ptr.size());
  if (rc == SAMD_OK && have_parts) rc = upload(&h->ly_bp_recs, parts.recs.data(), parts.recs.size());
  if (rc == SAMD_OK && have_parts) rc = upload(&h->ly_bp_slot_tab, parts.slot_tab.data(), parts.slot_tab.size());
  if (rc == SAMD_OK) rc = upload(&h->ly_ent_tab, ent_tab.data(), ent_tab.size());
  if (rc == SAMD_OK) rc = upload(&h->ly_xt_index, xt_index.data(), xt_index.size());
  if (rc == SAMD_OK) rc = upload(&h->ly_slot_tab, whole.slot_tab.data(), whole.slot_tab.size());
  if (rc == SAMD_OK) h->ly_ok = 1;
  return rc;
}

void free_onchip_ly_tables(samd_ldpc5g* h) {
  (void)hipFree(h->ly_rec_ptr); (void)hipFree(h->ly_recs); (void)hipFree(h->ly_ent_tab);
  (void)hipFree(h->ly_xt_index); (void)hipFree(h->ly_slot_tab);
  (void)hipFree(h->ly_bp_rec_ptr); (void)hipFree(h->ly_bp_recs); (void)hipFree(h->ly_bp_slot_tab);
}

static int ly_grid(const samd_ldpc5g* h, int batch) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int grid = std::min(batch, cus * (16 / std::max(4, h->ly_waves)));
  if (h->opt.onchip_grid > 0) grid = std::min(grid, h->opt.onchip_grid);
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
  typedef void (*kern_t)(const float*, float*, float*, RateMatch, int, int, int, float, float, int, int, int, int, int, int,
                         const int32_t*, const int4*, const int32_t*, const int32_t*, const int32_t*);
#define SAMD_LY_K(M) {{ldpc5g_decode_ly_kernel<false, M, false>, ldpc5g_decode_ly_kernel<true, M, false>}, \
                      {ldpc5g_decode_ly_kernel<false, M, true>, ldpc5g_decode_ly_kernel<true, M, true>}}
  static const kern_t kerns[3][2][2] = {SAMD_LY_K(SAMD_CN_MINSUM), SAMD_LY_K(SAMD_CN_BOXPLUS_PHI), SAMD_LY_K(SAMD_CN_BOXPLUS_PHI_FAST)};
#undef SAMD_LY_K
  const kern_t fn = kerns[minsum ? 0 : cn_mode == SAMD_CN_BOXPLUS_PHI ? 1 : 2][h->z % 64 != 0 ? 1 : 0][pow2 ? 1 : 0];
  SAMD_SET_MAX_LDS(fn, 160 * 1024);
  const int nbu = (h->n_vn + h->z - 1) / h->z;
  const RateMatch rm = make_rate_match(h);
  const float off = (cn_mode == SAMD_CN_OFFSET_MINSUM) ? offset : 0.f;
  // the boxplus rules walk the lists with split check-node items where the code has them (more LDS: the parts' scratch)
  const bool parts = !minsum && h->ly_bp_lds_bytes > 0;
  hipLaunchKernelGGL(fn, dim3(ly_grid(h, batch)), dim3(64 * h->ly_waves), (size_t)(parts ? h->ly_bp_lds_bytes : h->ly_lds_bytes), st, llr, out, ws,
                     rm, nbu, batch, num_iter, llr_max, off, hard_out, return_infobits, h->ly_msg_floats, h->ly_n_ext,
                     h->ly_zero_off, 64 * h->ly_waves, parts ? h->ly_bp_rec_ptr : h->ly_rec_ptr,
                     reinterpret_cast<const int4*>(parts ? h->ly_bp_recs : h->ly_recs), h->ly_ent_tab, h->ly_xt_index,
                     parts ? h->ly_bp_slot_tab : h->ly_slot_tab);
  return launch_status();
}

}  // namespace samd

#ifdef SAMD_LY_TRACE
extern "C" int samd_debug_set_ly_trace(unsigned long long* p) {
  unsigned long long* q = p ? p + 16 * 512 * 2 : nullptr;
  if (hipMemcpyToSymbol(HIP_SYMBOL(samd::g_ly_sub), &q, sizeof(q)) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(samd::g_ly_trace), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
