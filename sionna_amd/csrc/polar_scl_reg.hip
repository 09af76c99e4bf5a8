// Polar SC / SC-list decoder whose low decoding stages live in registers (north-star config C5).
//
// Replaces (reference src/sionna/phy/fec/polar/decoding.py): PolarSCLDecoder, default TF path with use_fast_scl
// (:525-723 metric / boxplus arithmetic, :919-1045 decoding recursion, :1345-1437 final CRC-aided selection), list
// sizes 1..32, and PolarSCDecoder (:122-263) - codes with n >= 64; the generic engine of polar.hip takes the rest.
// Same schedule, same float32 arithmetic (scl_math.h) and therefore the same bits as the generic engine in
// polar.hip and as oracle/polar_scl.c; what changes is where the state of the lowest tree levels lives.
//
// 7/8 of the ~2800 schedule operations of an n = 1024 code work on the 8-leaf subtrees at the bottom of the
// decoding tree, on 8..64 numbers each.  The generic engine keeps every stage in LDS behind per-slot pointer
// tables (lazy path copies), so each of those operations is a chain of 3-4 dependent LDS round trips around a
// few dozen arithmetic instructions - PMC: half of all wave cycles waiting, as many scalar as vector
// instructions.  Here a wave still owns one codeword, but lane = slot * W + j (W = 64 / L lanes per list slot):
//   * the LLRs of stages 0..R (R = log2 W; 2^s values at stage s, in lanes j < 2^s) and the partial sums of
//     those stages (two bits per stage in one register) are registers of the slot's lanes: f / g / combine
//     at these stages are a lane shift inside a 16-lane row (DPP) plus the arithmetic, no memory, no pointers;
//   * block metrics of rate-0 / repetition nodes at these stages are a DPP tree with the summation order of
//     scl_math.h (lane 0 of the halving tree);
//   * a fork ranks the 2L candidates with one LDS exchange (W/2 lanes per candidate share the comparisons),
//     matches dead slots to the second survivors with ballots and a scalar loop, and a dead slot pulls the
//     R + 3 registers of its new parent with ds_bpermute; path metric and sorted position of a slot are
//     registers of its first lane;
//   * stages above R keep the design of the generic engine (LDS for the next stages, L2 scratch for the top
//     G stages, lazy copies through the pointer tables), iterated by slot - the position order is only needed
//     for the tie-break of the ranking;
//   * the schedule hands over whole nodes of stage R + 1 as ONE record (SUBTREE): f / g from LDS around the two
//     register-stage subtrees, whose recursion is unrolled at compile time into straight-line code with
//     wave-uniform branches on the frozen pattern - the scalar unit (one per CU, shared by the 32 resident waves)
//     dispatches 213 instead of 2161 operations per n = 1024 decode, which was worth more than everything above
//     (3.0 -> 5.2 M decodes/s at C5).
// Measured at C5 (n = 1024, k = 512 + CRC11, L = 8, batch 32768): 5.3 M decodes/s against 1.3 M of the generic
// engine; 113 k vector + 47 k scalar instructions per decode (247 k + 237 k before), VALU issue 0.49 of peak.
#include "polar_scl.h"
#include <type_traits>

namespace samd {

// lane i <- lane i + K inside a row of 16 lanes (lanes past the row read 0)
template <int K>
__device__ __forceinline__ int dpp_up_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x100 + K, 0xF, 0xF, true);   // row_shl:K
}
template <int K>
__device__ __forceinline__ float dpp_up(float v) { return __int_as_float(dpp_up_i<K>(__float_as_int(v))); }
// lane i <- lane i - K inside a row of 16 lanes
template <int K>
__device__ __forceinline__ int dpp_down_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x110 + K, 0xF, 0xF, true);   // row_shr:K
}

// LDS-typed pointers: loads and stores through them are ds_* instructions whatever the optimiser merges
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

template <int MAX, class F>
__device__ __forceinline__ void stage_switch(int s, F&& f) {   // f(integral_constant<s>) for the wave-uniform s <= MAX
  if constexpr (MAX >= 0) {
    if (s == MAX) f(std::integral_constant<int, MAX>{});
    else stage_switch<MAX - 1>(s, f);
  }
}

// sum of the 2^S values in lanes j < 2^S of a slot, in the order of scl_math.h's halving tree; result in lane j = 0
template <int S>
__device__ __forceinline__ float slot_tree(float v) {
  if constexpr (S >= 4) v += dpp_up<8>(v);
  if constexpr (S >= 3) v += dpp_up<4>(v);
  if constexpr (S >= 2) v += dpp_up<2>(v);
  if constexpr (S >= 1) v += dpp_up<1>(v);
  return v;
}

template <int L>
struct SclRegLayout {
  static constexpr int W = 64 / L;                                       // lanes per slot
  static constexpr int R = W >= 16 ? 4 : W == 8 ? 3 : W == 4 ? 2 : 1;     // register stages 0..R
  static constexpr int H = 1 << R;                                       // values of stage R: the lanes j < H of a slot
};                                                                        // hold register stages (H == W for L >= 4)

static inline size_t scl_reg_wstride(int n) { return (size_t)(((n + 31) / 32 + 3) / 4 * 4); }

// SC: PolarSCDecoder (one path, hard decisions by the sign of the leaf LLR, rate-0 shortcut only - decoding.py:122-263):
// no metrics, no forks.  List sizes 1 and 2 use 16 of their 64 / 32 lanes per slot for the register stages.
template <int L, bool SC>
__global__ __launch_bounds__(64, 8) void polar_scl_reg_kernel(SclArgs p) {
  constexpr int W = SclRegLayout<L>::W, R = SclRegLayout<L>::R, H = SclRegLayout<L>::H;
  static_assert(H <= W && W >= 2 && (!SC || L == 1), "register stages live in the first H lanes of a slot");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int n = p.n, m = p.m;
  const unsigned lane0 = threadIdx.x;
  const int top = m - p.gstages;                       // stages (R, top) in LDS, [top, m) in L2 scratch, m = channel
  const int hn = 1 << top;
  const int wstride = (((n + 31) / 32 + 3) / 4) * 4, wq = wstride / 4;
  // fixed-size arrays first (compile-time LDS offsets), the arrays sized by n and the L2 split last
  unsigned char* tab = reinterpret_cast<unsigned char*>(smem);               // [L][3][16] slot holding the stage-s
  uint2* ck = reinterpret_cast<uint2*>(tab + (size_t)L * 48);                // LLRs / left sums / right sums; [2L] sort keys
  float* cv = reinterpret_cast<float*>(ck + 2 * L);                          // [L] penalised final metrics
  float* pm_s = cv + L;                                                      // [L] final metrics by position
  int* order = reinterpret_cast<int*>(pm_s + L);                             // [L] position -> slot
  float* blk = reinterpret_cast<float*>(order + L);                          // [2L]
  // the header is 84 L bytes: round it up to 16 so that fzb / bits / llr keep the alignment their 128-bit accesses want
  // for L = 1, 2 as well (scl_reg_lds_bytes reserves 88 L + 64)
  uint32_t* fzb = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(smem) + (((size_t)L * 84 + 15) & ~(size_t)15));   // [wstride] frozen-bit map
  uint32_t* bits = fzb + wstride;                                            // [L][wstride] decided u bits
  float* llr = reinterpret_cast<float*>(bits + (size_t)L * wstride);         // [L][hn]  stage s at [2^s, 2^(s+1))
  unsigned char* beta = reinterpret_cast<unsigned char*>(llr + (size_t)L * hn);   // [L][hn]  bit 0 left, bit 1 right result
  lds_f32* llr3 = (lds_f32*)llr;
  lds_u8* beta3 = (lds_u8*)beta;
  const bool ch_aligned = (reinterpret_cast<uintptr_t>(p.llr_in) & 15u) == 0;   // rows of n >= 32 floats keep it
  float* gsc = p.gscratch + (size_t)blockIdx.x * L * (n - hn);
  unsigned char* gbe = p.gbeta + (size_t)blockIdx.x * L * (n - hn);

  for (int i = lane0; i < wstride; i += 64) fzb[i] = 0xFFFFFFFFu;
  __syncthreads();
  for (int i = lane0; i < p.k; i += 64) atomicAnd(&fzb[p.info_pos[i] >> 5], ~(1u << (p.info_pos[i] & 31)));
  __syncthreads();

  for (int b = blockIdx.x; b < p.batch; b += gridDim.x) {
    const float* llr_ch = p.llr_in + (size_t)b * (p.src_a ? p.n_in : n);    // logits: negated where they are read (LLR = -logit)
    // channel value of mother-code position idx: the received row itself, or Polar5GDecoder's rate recovery as an index
    // (the torch gathers in front of the kernel were four launches, ~3 % of a C5 step; decoding.py:2018-2052)
    auto ch_val = [&](int idx) __attribute__((always_inline)) -> float {
      if (!p.src_a) return llr_ch[idx];
      const int ia = p.src_a[idx];
      float v = ia >= 0 ? llr_ch[ia] : 0.f;
      v = ia == -2 ? -p.rm_fill : v;
      if (p.src_b) {
        const int ib = p.src_b[idx];
        v = v + (ib >= 0 ? llr_ch[ib] : 0.f);
      }
      return v;
    };
    // value idx of the stage-s LLRs held by slot sl; every branch is wave-uniform and has its own address space
    auto ld_llr = [&](int sl, int s, int idx) __attribute__((always_inline)) -> float {
      if (s == m) return -ch_val(idx);
      if (s < top) return llr3[sl * hn + (1 << s) + idx];
      return gsc[(size_t)sl * (n - hn) + ((1 << s) - hn) + idx];
    };
    // result bit `v` of side a1 into byte idx of the stage-s partial sums of slot sl (the other side's bit stays)
    auto put_beta = [&](int sl, int s, int idx, uint32_t v, int a1) __attribute__((always_inline)) {
      const uint32_t keep = a1 ? 1u : 2u;
      if (s < top) {
        lds_u8* d = beta3 + sl * hn + (1 << s) + idx;
        *d = (unsigned char)((*d & keep) | (v << a1));
      } else {
        unsigned char* d = gbe + (size_t)sl * (n - hn) + ((1 << s) - hn) + idx;
        *d = (unsigned char)((*d & keep) | (v << a1));
      }
    };

    // four consecutive values (idx a multiple of 4) - the upper stages hold >= 4 values per half
    auto ld_llr4 = [&](int sl, int s, int idx) __attribute__((always_inline)) -> float4 {
      if (s == m) {
        if (p.src_a || !ch_aligned) return make_float4(-ch_val(idx), -ch_val(idx + 1), -ch_val(idx + 2), -ch_val(idx + 3));
        const float4 v = *reinterpret_cast<const float4*>(llr_ch + idx);
        return make_float4(-v.x, -v.y, -v.z, -v.w);
      }
      if (s < top) {
        const f32x4 v = *(lds_f32x4*)(llr3 + sl * hn + (1 << s) + idx);
        return make_float4(v.x, v.y, v.z, v.w);
      }
      return *reinterpret_cast<const float4*>(gsc + (size_t)sl * (n - hn) + ((1 << s) - hn) + idx);
    };
    auto st_llr4 = [&](int sl, int s, int idx, float4 v) __attribute__((always_inline)) {
      if (s < top) *(lds_f32x4*)(llr3 + sl * hn + (1 << s) + idx) = f32x4{v.x, v.y, v.z, v.w};
      else *reinterpret_cast<float4*>(gsc + (size_t)sl * (n - hn) + ((1 << s) - hn) + idx) = v;
    };
    auto ld_beta4 = [&](int sl, int s, int idx) __attribute__((always_inline)) -> uint32_t {
      if (s < top) return *(lds_u32*)(beta3 + sl * hn + (1 << s) + idx);
      return *reinterpret_cast<const uint32_t*>(gbe + (size_t)sl * (n - hn) + ((1 << s) - hn) + idx);
    };
    auto put_beta4 = [&](int sl, int s, int idx, uint32_t v, int a1) __attribute__((always_inline)) {   // v: one result bit per byte
      const uint32_t keep = a1 ? 0x01010101u : 0x02020202u;
      if (s < top) {
        lds_u32* d = (lds_u32*)(beta3 + sl * hn + (1 << s) + idx);
        *d = (*d & keep) | (v << a1);
      } else {
        uint32_t* d = reinterpret_cast<uint32_t*>(gbe + (size_t)sl * (n - hn) + ((1 << s) - hn) + idx);
        *d = (*d & keep) | (v << a1);
      }
    };

    for (int i = lane0; i < L * wstride; i += 64) bits[i] = 0u;
    for (int i = lane0; i < L * 48; i += 64) tab[i] = (unsigned char)(i / 48);
    // LLRs of the register stages: separate scalars selected by the wave-uniform stage (an indexed array, or
    // references captured by a lambda, would be placed in scratch memory)
    float A0 = 0.f, A1 = 0.f, A2 = 0.f, A3 = 0.f, A4 = 0.f;
    // (round 6: BRANCHES on the wave-uniform stage.  Written as selections - `A0 = s == 0 ? r : A0` - each became s_cmp +
    // s_cselect_b64 vcc + v_cndmask_b32_e32, and a v_cndmask_b32_e32 that reads a vcc written by a SCALAR instruction takes ~24
    // cycles of the SIMD's vector pipe on gfx950 against ~2.5 behind a vector comparison (profiles/r06w_valu_rate2.txt); the empty
    // asm statements keep the optimiser from converting the branches back)
#define SCL_GETA(s_)                                                                     \
  ({                                                                                     \
    const int gs_ = (s_);                                                                \
    float gx_;                                                                           \
    if (R >= 4 && gs_ == 4) { asm volatile(""); gx_ = A4; }                              \
    else if (R >= 3 && gs_ == 3) { asm volatile(""); gx_ = A3; }                         \
    else if (R >= 2 && gs_ == 2) { asm volatile(""); gx_ = A2; }                         \
    else if (gs_ == 1) { asm volatile(""); gx_ = A1; }                                   \
    else { asm volatile(""); gx_ = A0; }                                                 \
    gx_;                                                                                 \
  })
#define SCL_SETA(s_, r_)                                                                 \
  do {                                                                                   \
    const int ss_ = (s_);                                                                \
    const float sr_ = (r_);                                                              \
    if (ss_ == 0) { asm volatile(""); A0 = sr_; }                                        \
    else if (ss_ == 1) { asm volatile(""); A1 = sr_; }                                   \
    else if (R >= 2 && ss_ == 2) { asm volatile(""); A2 = sr_; }                         \
    else if (R >= 3 && ss_ == 3) { asm volatile(""); A3 = sr_; }                         \
    else if (R >= 4 && ss_ == 4) { asm volatile(""); A4 = sr_; }                         \
  } while (0)
    uint32_t bb = 0u;                                   // bit 2s / 2s+1: left / right child result at stage s, position j
    float pm = lane0 / W == 0 ? 0.f : kPolarLlrMax;     // decoding.py:1029-1033 (first lane of the slot)
    int pos = lane0 / W;
    __syncthreads();

    int next_rec = p.ops[0];
    for (int ip = 0;; ++ip) {
      const int rec = next_rec;
      next_rec = (ip + 1 < p.num_ops) ? p.ops[ip + 1] : (int)OP_END;
      int op = rec & 7, a2 = ((rec >> 8) & 0xFFF) - 2048;
      const int s = (rec >> 3) & 15, a1 = (rec >> 7) & 1;
      if (op == OP_END) break;
      uint32_t fm = 0u;                                     // frozen pattern of a SUBTREE record (bit i = leaf i frozen)
      if (op == OP_SUBTREE) {
        const uint32_t word = fzb[a2 >> 5] >> (a2 & 31);
        fm = (s >= 5) ? word : (word & ((1u << (1 << s)) - 1u));
        if (s == R + 1 && ((rec >> 20) & 1)) {              // a node above the register stages that the fast-SCL rules
          const uint32_t all = (R + 1 >= 5) ? 0xFFFFFFFFu : ((1u << (1 << (R + 1))) - 1u);   // replace as a whole
          if (fm == all) op = OP_RATE0;
          else if (!SC && fm == (all >> 1)) { op = OP_REP; a2 += (1 << (R + 1)) - 1; }
        }
      }
      // the lane index is made opaque per operation: everything derived from it (slot, position, addresses, lane
      // predicates) is recomputed with a few VALU operations instead of being hoisted out of the schedule loop into
      // ~30 registers that live for the whole decode and cost occupancy
      unsigned lane = lane0;
      asm volatile("" : "+v"(lane));
      const unsigned slot = lane / W, j = lane % W;
      const bool head = j == 0;
      // ---- fork of every path at an information bit: candidate (u, slot) has metric pm + m_u and sort index
      // u L + position (stable sort, :1345-1390); returns the slots (bit = first lane) whose decided bit is 1
      auto fork = [&](float m0, float m1, int bit_index) __attribute__((always_inline)) -> unsigned long long {
        // The 2L sort keys are exchanged through LDS as 64-bit integers (metric bits : sort index) - metrics are
        // sums of non-negative terms, so their bit patterns order like the numbers - and one unsigned 64-bit compare
        // is the "smaller metric, ties by index" rule.
        const float c0 = pm + m0, c1 = pm + m1;
        if (head) {
          ck[slot] = make_uint2((uint32_t)pos, __float_as_uint(c0));
          ck[L + slot] = make_uint2((uint32_t)(L + pos), __float_as_uint(c1));
        }
        __syncthreads();
        constexpr int G2 = W / 2;                           // lanes per candidate
        constexpr int PER = (2 * L >= G2) ? (2 * L) / G2 : 1;     // comparisons per lane (lists of 1, 2: lanes q < 2L only)
        const unsigned u = j & 1u, q = j >> 1;
        const uint2 me2 = ck[u * L + slot];
        const unsigned long long me = ((unsigned long long)me2.y << 32) | me2.x;
        int rank = 0;
        if constexpr (PER == 4) {
          const uint4 a = reinterpret_cast<const uint4*>(ck)[2 * q], c = reinterpret_cast<const uint4*>(ck)[2 * q + 1];
          rank += ((((unsigned long long)a.y << 32) | a.x) < me) ? 1 : 0;
          rank += ((((unsigned long long)a.w << 32) | a.z) < me) ? 1 : 0;
          rank += ((((unsigned long long)c.y << 32) | c.x) < me) ? 1 : 0;
          rank += ((((unsigned long long)c.w << 32) | c.z) < me) ? 1 : 0;
        } else {
#pragma unroll 8
          for (int t = 0; t < PER; ++t) {
            const unsigned d = q * PER + t;
            const uint2 d2 = ck[d < 2u * L ? d : 0u];
            rank += (d < 2u * L && (((unsigned long long)d2.y << 32) | d2.x) < me) ? 1 : 0;
          }
        }
        if constexpr (W >= 16) rank += dpp_up_i<8>(rank);
        if constexpr (W >= 8) rank += dpp_up_i<4>(rank);
        if constexpr (W >= 4) rank += dpp_up_i<2>(rank);
        const int r0 = rank, r1 = dpp_up_i<1>(rank);        // first lane of the slot: ranks of (0, slot), (1, slot)
        const bool stay0 = r0 < L, stay1 = r1 < L;
        unsigned long long md = __builtin_amdgcn_ballot_w64(head && !stay0 && !stay1);     // slots without survivor
        unsigned long long mb = __builtin_amdgcn_ballot_w64(head && stay0 && stay1);       // slots with two survivors
        unsigned long long m1mask = __builtin_amdgcn_ballot_w64(head && !stay0 && stay1);  // slots that continue with u = 1
        float npm = stay0 ? c0 : c1;
        int npos = stay0 ? r0 : r1;
        int srcl = lane;
        const bool any = md != 0ull;
        while (md != 0ull && mb != 0ull) {                  // the i-th dead slot takes the second child of the i-th
          const int d = __builtin_ctzll(md), sp = __builtin_ctzll(mb);   // slot with two survivors
          md &= md - 1ull;
          mb &= mb - 1ull;
          if ((lane & ~(W - 1)) == d) srcl = sp + j;
          m1mask |= 1ull << d;
        }
        if (any) {
          const int addr = srcl << 2;
          const bool moved = srcl != (int)lane;
          const unsigned ssl = (unsigned)srcl / W;
          // the lazy copy of everything above stage R (decided bits, pointer tables of the upper stages) is requested
          // first, the register pulls next: one wait covers both
          uint4* bw = reinterpret_cast<uint4*>(bits);
          uint4* tw = reinterpret_cast<uint4*>(tab);
          const bool cw = moved && (int)j < wq, ct = moved && j < 3u;
          uint4 wv = make_uint4(0u, 0u, 0u, 0u), tv = wv;
          if (cw) wv = bw[ssl * wq + j];
          if (ct) tv = tw[ssl * 3 + j];
          auto pull = [&](float v) __attribute__((always_inline)) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); };
          A1 = pull(A1);
          if (R >= 2) A2 = pull(A2);
          if (R >= 3) A3 = pull(A3);
          if (R >= 4) A4 = pull(A4);
          bb = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)bb);
          const float pc1 = pull(c1);
          const int pr1 = __builtin_amdgcn_ds_bpermute(addr, r1);
          if (cw) bw[slot * wq + j] = wv;
          if (ct) tw[slot * 3 + j] = tv;
          if (moved) {
            npm = pc1;
            npos = pr1;
            for (int e = j + W; e < wq; e += W) bw[slot * wq + e] = bw[ssl * wq + e];      // W < 8 lanes per slot
            for (int e = j + W; e < 3; e += W) tw[slot * 3 + e] = tw[ssl * 3 + e];
          }
          // LDS executes a wave's operations in order: later readers of these words need no wait, only program order
          asm volatile("" ::: "memory");
        }
        pm = npm;
        pos = npos;
        if (head && ((m1mask >> lane) & 1ull)) bits[slot * wstride + (bit_index >> 5)] |= 1u << (bit_index & 31);
        return m1mask;
      };
      // ---- a complete subtree of register stage S: the recursion of decoding.py:919-1005 (rate-0 / repetition
      // shortcuts when `fast`, else f, left child, g, right child, combine) unrolled at compile time - straight-line
      // code with wave-uniform branches on the frozen pattern fm (bit i = leaf i frozen), no schedule dispatch
      auto Aref = [&](auto S_) __attribute__((always_inline)) -> float& {
        constexpr int S = decltype(S_)::value;
        if constexpr (S == 0) return A0;
        else if constexpr (S == 1) return A1;
        else if constexpr (S == 2) return A2;
        else if constexpr (S == 3) return A3;
        else return A4;
      };
      auto subtree = [&](auto self, auto S_, uint32_t fm, int first, int side, bool fast) __attribute__((always_inline)) -> void {
        constexpr int S = decltype(S_)::value, nl = 1 << S;
        constexpr uint32_t all = (1u << nl) - 1u;
        auto setres = [&](unsigned long long ones) __attribute__((always_inline)) {   // the node's result: the all-u codeword
          const int sh = 2 * S + side;
          bb = (bb & ~(1u << sh)) | ((uint32_t)((ones >> (lane & ~(W - 1))) & 1ull) << sh);
        };
        if constexpr (S == 0) {
          if constexpr (SC) {
            // u = 0.5 (1 - sign(l)), an exact zero decides 1 (decoding.py:208-212); frozen: 0
            const unsigned long long ones = (fm & 1u) ? 0ull : __builtin_amdgcn_ballot_w64(head && A0 <= 0.f);
            if (head && (ones & 1ull)) bits[first >> 5] |= 1u << (first & 31);
            setres(ones);
          } else {
            const float l = clampf(A0, -kPolarLlrMax, kPolarLlrMax);
            const float tl = scl_T(fabsf(l));
            const float m0 = fmaxf(-l, 0.f) + tl;
            if (fm & 1u) { pm += m0; setres(0ull); }
            else setres(fork(m0, fmaxf(l, 0.f) + tl, first));
          }
        } else {
          if (SC && fast && fm == all) {
            setres(0ull);                                                               // rate-0, no metric to keep
          } else if (!SC && fast && (fm == all || fm == (all >> 1))) {
            const float l = clampf(Aref(S_), -kPolarLlrMax, kPolarLlrMax);
            const float tl = scl_T(fabsf(l));
            float t0 = fmaxf(-l, 0.f) + tl, t1 = fmaxf(l, 0.f) + tl;
            if (j >= (unsigned)nl) { t0 = 0.f; t1 = 0.f; }
            const float m0 = slot_tree<S>(t0);
            if (fm == all) { pm += m0; setres(0ull); }                                  // rate-0
            else setres(fork(m0, slot_tree<S>(t1), first + nl - 1));                    // repetition: the last bit
          } else {
            constexpr std::integral_constant<int, S - 1> C_{};
            {
              const float x = Aref(S_), y = dpp_up<nl / 2>(x);
              Aref(C_) = cn_op(x, y);
            }
            self(self, C_, fm & (all >> (nl / 2)), first, 0, fast);
            {
              const float x = Aref(S_), y = dpp_up<nl / 2>(x);   // re-read: forks replace the registers of dead slots
              Aref(C_) = (1.f - 2.f * (float)((bb >> (2 * (S - 1))) & 1u)) * x + y;     // vn_op :707-714
            }
            self(self, C_, fm >> (nl / 2), first + nl / 2, 1, fast);
            const uint32_t l = (bb >> (2 * (S - 1))) & 1u, r = (bb >> (2 * (S - 1) + 1)) & 1u;
            const uint32_t hi = (uint32_t)dpp_down_i<nl / 2>((int)r);
            const uint32_t nb = (j < (unsigned)(nl / 2)) ? (l ^ r) : hi;
            const int sh = 2 * S + side;
            bb = (bb & ~(1u << sh)) | (nb << sh);
          }
        }
      };
      if (op == OP_SUBTREE) {
        // a2 = index of the subtree's first bit.  Stage R + 1 (what pack_schedule emits for this engine): f / g from
        // the stage-(R+1) LLRs in memory around the two register-stage subtrees, then the combine into memory; stage R:
        // one register-stage subtree; stage 1: two information leaves.  ONE call site of the unrolled recursion.
        const bool fast = (rec >> 20) & 1;
        if (s >= R) {
          const int halves = (s == R) ? 1 : 2;
#pragma unroll 1
          for (int h = 0; h < halves; ++h) {
            if (s == R + 1) {
              const int si = (int)tab[slot * 48 + s];
              const int jm = (int)(j & (H - 1));      // lanes past the register stages repeat the first H (no stray reads)
              const float x = ld_llr(si, s, jm), y = ld_llr(si, s, jm + H);
              float r;
              if (h == 0) r = cn_op(x, y);
              else r = (1.f - 2.f * (float)((bb >> (2 * R)) & 1u)) * x + y;         // vn_op :707-714
              SCL_SETA(R, r);
            }
            const uint32_t fh = (s == R) ? fm : ((h ? (fm >> H) : fm) & ((1u << H) - 1u));
            subtree(subtree, std::integral_constant<int, R>{}, fh, a2 + h * H, (s == R) ? a1 : h, fast);
          }
          if (s == R + 1) {
            const uint32_t l = (bb >> (2 * R)) & 1u, r = (bb >> (2 * R + 1)) & 1u;
            if (j < (unsigned)H) {
              put_beta(slot, R + 1, j, l ^ r, a1);
              put_beta(slot, R + 1, H + j, r, a1);
            }
            if (head) tab[slot * 48 + (a1 ? 32 : 16) + R + 1] = (unsigned char)slot;
            __syncthreads();
          }
        } else {
          subtree(subtree, std::integral_constant<int, 1>{}, fm & 3u, a2, a1, fast);
        }
      } else if (op == OP_F || op == OP_G) {
        // inputs at stage s (2^s values), outputs at stage so = s - 1; g uses the left results of stage so
        const int so = s - 1;
        if (so <= R) {
          // one instance of the arithmetic for all register stages: only the operand fetch depends on the stage
          float x = 0.f, y = 0.f;
          if (so < R) {
            x = SCL_GETA(s);
            stage_switch<R - 1>(so, [&](auto S_) __attribute__((always_inline)) { y = dpp_up<(1 << decltype(S_)::value)>(x); });
          } else {
            const int si = (int)tab[slot * 48 + s];
            const int jm = (int)(j & (H - 1));
            x = ld_llr(si, s, jm);
            y = ld_llr(si, s, jm + H);
          }
          float r;
          if (op == OP_F) r = cn_op(x, y);
          else r = (1.f - 2.f * (float)((bb >> (2 * so)) & 1u)) * x + y;             // vn_op :707-714
          SCL_SETA(so, r);
        } else {
          // f of the channel LLRs is the same for every path: computed once, all tables point to slot 0
          const int half = 1 << so;
          const bool shared = s == m && op == OP_F;
          for (int w = lane; w < (shared ? 1 : L) * half / 4; w += 64) {     // four outputs per lane
            const int e = w * 4, sl = e >> so, jj = e & (half - 1);
            const int si = (s == m) ? 0 : (int)tab[sl * 48 + s];
            const float4 x = ld_llr4(si, s, jj), y = ld_llr4(si, s, jj + half);
            float4 r;
            if (op == OP_F) {
              r = make_float4(cn_op(x.x, y.x), cn_op(x.y, y.y), cn_op(x.z, y.z), cn_op(x.w, y.w));
            } else {
              const uint32_t lb = ld_beta4(tab[sl * 48 + 16 + so], so, jj);
              r.x = (1.f - 2.f * (float)(lb & 1u)) * x.x + y.x;
              r.y = (1.f - 2.f * (float)((lb >> 8) & 1u)) * x.y + y.y;
              r.z = (1.f - 2.f * (float)((lb >> 16) & 1u)) * x.z + y.z;
              r.w = (1.f - 2.f * (float)((lb >> 24) & 1u)) * x.w + y.w;
            }
            st_llr4(sl, so, jj, r);
          }
          if (lane < L) tab[lane * 48 + so] = (unsigned char)(shared ? 0 : lane);
          __syncthreads();
        }
      } else if (op == OP_COMBINE) {
        // children results at stage s -> this node's result at stage s + 1 on side a1: (l ^ r, r)
        if (s < R) {
          stage_switch<R - 1>(s, [&](auto S_) __attribute__((always_inline)) {
            constexpr int S = decltype(S_)::value, sz = 1 << S;
            const uint32_t l = (bb >> (2 * S)) & 1u, r = (bb >> (2 * S + 1)) & 1u;
            const uint32_t hi = (uint32_t)dpp_down_i<sz>((int)r);
            const uint32_t nb = (j < sz) ? (l ^ r) : hi;
            const int sh = 2 * (S + 1) + a1;
            bb = (bb & ~(1u << sh)) | (nb << sh);
          });
        } else if (s == R) {
          const uint32_t l = (bb >> (2 * R)) & 1u, r = (bb >> (2 * R + 1)) & 1u;
          if (j < (unsigned)H) {
            put_beta(slot, R + 1, j, l ^ r, a1);
            put_beta(slot, R + 1, H + j, r, a1);
          }
          if (head) tab[slot * 48 + (a1 ? 32 : 16) + R + 1] = (unsigned char)slot;
          __syncthreads();
        } else {
          const int sz = 1 << s;
          for (int w = lane; w < L * sz / 4; w += 64) {       // four positions per lane, one bit per byte
            const int e = w * 4, sl = e >> s, jj = e & (sz - 1);
            const uint32_t l = ld_beta4(tab[sl * 48 + 16 + s], s, jj) & 0x01010101u;
            const uint32_t r = (ld_beta4(tab[sl * 48 + 32 + s], s, jj) >> 1) & 0x01010101u;
            put_beta4(sl, s + 1, jj, l ^ r, a1);
            put_beta4(sl, s + 1, sz + jj, r, a1);
          }
          if (lane < L) tab[lane * 48 + (a1 ? 32 : 16) + s + 1] = (unsigned char)lane;
          __syncthreads();
        }
      } else {
        // ---- leaf / rate-0 / repetition node at stage s, side a1; a2 = (last) bit index (frozen leaf: -1-index)
        const bool info = (op == OP_REP) || (op == OP_LEAF && a2 >= 0);
        // block metrics of the slot in its first lane: m0 = sum softplus(-l), m1 = sum softplus(+l)
        float m0 = 0.f, m1 = 0.f;
        if constexpr (SC) {
          // hard decisions only
        } else if (s <= R) {
          const float l = clampf(SCL_GETA(s), -kPolarLlrMax, kPolarLlrMax);
          const float tl = scl_T(fabsf(l));                   // shared by softplus(-l) and softplus(l)
          m0 = fmaxf(-l, 0.f) + tl;
          m1 = fmaxf(l, 0.f) + tl;
          if (j >= (1 << s)) { m0 = 0.f; m1 = 0.f; }
          // halving tree of scl_math.h over the 2^s lanes of the slot (steps over absent lanes would add zeros)
          if (R >= 4 && s >= 4) { m0 += dpp_up<8>(m0); m1 += dpp_up<8>(m1); }
          if (R >= 3 && s >= 3) { m0 += dpp_up<4>(m0); m1 += dpp_up<4>(m1); }
          if (R >= 2 && s >= 2) { m0 += dpp_up<2>(m0); m1 += dpp_up<2>(m1); }
          if (s >= 1) { m0 += dpp_up<1>(m0); m1 += dpp_up<1>(m1); }
        } else {
          const int sz = 1 << s;
          for (int sl = 0; sl < L; ++sl) {
            const int si = (s == m) ? 0 : (int)tab[sl * 48 + s];
            // scl_math.h order: lane l accumulates terms l, l + 64, ... ascending, then the halving tree over 64 lanes
            float a0 = 0.f, a1v = 0.f;
            for (int jj = lane; jj < sz; jj += 64) {
              const float l = clampf(ld_llr(si, s, jj), -kPolarLlrMax, kPolarLlrMax);
              const float tl = scl_T(fabsf(l));
              a0 += fmaxf(-l, 0.f) + tl;
              a1v += fmaxf(l, 0.f) + tl;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a0 += __shfl_xor(a0, o, 64); a1v += __shfl_xor(a1v, o, 64); }
            if (slot == sl) { m0 = a0; m1 = a1v; }
          }
        }
        unsigned long long ones = 0ull;                       // slots (bit = first lane) whose decided bit is 1
        if constexpr (SC) {
          if (info) {                                         // a leaf (the SC schedule has no repetition nodes)
            ones = __builtin_amdgcn_ballot_w64(head && A0 <= 0.f);
            if (head && (ones & 1ull)) bits[a2 >> 5] |= 1u << (a2 & 31);
          }
        } else if (!info) {
          pm += m0;                                           // frozen leaf / rate-0: all-zero block
        } else {
          ones = fork(m0, m1, a2);                            // the node's only information bit is its last
        }
        // the node's result: the all-u codeword
        if (s <= R) {
          const int sh = 2 * s + a1;
          const uint32_t nb = (uint32_t)((ones >> (lane & ~(W - 1))) & 1ull);
          bb = (bb & ~(1u << sh)) | (nb << sh);
        } else {
          const int sz = 1 << s;
          for (int w = lane; w < L * sz / 4; w += 64) {
            const int e = w * 4, sl = e >> s, jj = e & (sz - 1);
            put_beta4(sl, s, jj, (uint32_t)((ones >> (sl * W)) & 1ull) * 0x01010101u, a1);
          }
          if (lane < L) tab[lane * 48 + (a1 ? 32 : 16) + s] = (unsigned char)lane;
        }
        __syncthreads();
      }
    }
    // ---- final selection (decoding.py:1396-1419): CRC over the info bits of every path, penalty, first min
    const int lane = lane0;
    if (lane % W == 0) { pm_s[pos] = pm; order[pos] = lane / W; }
    __syncthreads();
    if (p.crc_len > 0) {
      // parallel CRC of every path: remainder = XOR of the table entries of its set bits (linear over GF(2))
      for (int q = 0; q < L; ++q) {
        const uint32_t* bw = bits + (size_t)order[q] * wstride;
        uint32_t acc = 0u;
        for (int i = lane; i < p.k; i += 64) {
          const int src = p.iil_inv ? p.iil_inv[i] : i;
          const int ps = p.info_pos[src];
          if ((bw[ps >> 5] >> (ps & 31)) & 1u) acc ^= p.crc_tab[i];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc ^= (uint32_t)__shfl_xor((int)acc, o, 64);
        if (lane == 0) {
          blk[q] = acc == 0u ? 1.f : 0.f;
          cv[q] = pm_s[q] + (acc == 0u ? 0.f : kPolarLlrMax * (float)p.k);
        }
      }
    } else if (lane < L) {
      cv[lane] = pm_s[lane];
    }
    __syncthreads();
    // first minimum of the penalised metrics in the order of the final stable sort by path metric (:1391, 1415):
    // among equal penalised metrics the smaller path metric, then the lower position
    int best = 0;
    for (int q = 1; q < L; ++q)
      if (cv[q] < cv[best] || (cv[q] == cv[best] && pm_s[q] < pm_s[best])) best = q;
    const uint32_t* bw = bits + (size_t)order[best] * wstride;
    for (int i = lane; i < p.k; i += 64) {
      const int ps = p.info_pos[i];
      p.u_hat[(size_t)b * p.k + i] = (float)((bw[ps >> 5] >> (ps & 31)) & 1u);
    }
    if (lane == 0 && p.crc_status) p.crc_status[b] = p.crc_len > 0 ? blk[best] : 1.f;
    __syncthreads();
  }
}

int scl_reg_stages(int n, int list_size, int sc_mode) {
  // read once: the host caches a schedule built for the engine this function announced, so the decision must not
  // change during the life of the process
  static CachedOpt scl_generic_opt("SAMD_SCL_GENERIC");
  const bool force_generic = scl_generic_opt.is_set();
  if (force_generic) return -1;
  if (list_size != 1 && list_size != 2 && list_size != 4 && list_size != 8 && list_size != 16 && list_size != 32) return -1;
  if (sc_mode && list_size != 1) return -1;
  int m = 0;
  while ((1 << m) < n) ++m;
  const int w = 64 / list_size, r = w >= 16 ? 4 : w == 8 ? 3 : w == 4 ? 2 : 1;
  return (m >= r + 2 && n <= 1024) ? r : -1;
}

bool scl_reg_supported(int n, int list_size, int sc_mode) { return scl_reg_stages(n, list_size, sc_mode) >= 0; }

size_t scl_reg_lds_bytes(int n, int L) {
  int m = 0;
  while ((1 << m) < n) ++m;
  const size_t hn = (size_t)1 << (m - scl_gstages(n, true));
  return (size_t)L * hn * 5 + (size_t)L * scl_reg_wstride(n) * 4 + (size_t)L * 48 + (size_t)L * 4 * 10 + scl_reg_wstride(n) * 4 + 64;
}

template <int L, bool SC>
static int scl_reg_launch_l(const SclArgs& p, int grid, hipStream_t stream) {
  const size_t lds = scl_reg_lds_bytes(p.n, L);
  SAMD_SET_MAX_LDS((polar_scl_reg_kernel<L, SC>), 160 * 1024);
  hipLaunchKernelGGL((polar_scl_reg_kernel<L, SC>), dim3(grid), dim3(64), lds, stream, p);
  return launch_status();
}

int scl_reg_launch(const SclArgs& p, int grid, hipStream_t stream) {
  switch (p.L) {
    case 1: return p.sc_mode ? scl_reg_launch_l<1, true>(p, grid, stream) : scl_reg_launch_l<1, false>(p, grid, stream);
    case 2: return scl_reg_launch_l<2, false>(p, grid, stream);
    case 4: return scl_reg_launch_l<4, false>(p, grid, stream);
    case 8: return scl_reg_launch_l<8, false>(p, grid, stream);
    case 16: return scl_reg_launch_l<16, false>(p, grid, stream);
    case 32: return scl_reg_launch_l<32, false>(p, grid, stream);
  }
  set_error("list size not supported by the register engine");
  return SAMD_ERR_UNSUPPORTED;
}

#undef SCL_GETA
#undef SCL_SETA

}  // namespace samd
