// CRC, Polar encoder and successive-cancellation (list) Polar decoder (north-star config C5).
//
// Replaces (reference src/sionna/phy/fec/):
//   CRCEncoder.call / CRCDecoder.call     crc.py:175-215, 289-321  (dense GF(2) generator matmul)
//   PolarEncoder.call                     polar/encoding.py:140-209 (scatter + log2(n) gather/XOR stages)
//   PolarSCDecoder                        polar/decoding.py:122-263
//   PolarSCLDecoder (default TF path, use_fast_scl) polar/decoding.py:525-723, 919-1045, 1345-1437
//
// MI355X design of the decoder: the reference unrolls the whole decoding tree into a TF graph whose every leaf step
// re-concatenates a [batch, 2L, log2(n)+1, n] float state (1.44 MB per codeword at n=1024, L=8) and itself recommends
// a NumPy fallback for n > 128.  Here ONE wave owns one codeword, 32 codewords per CU:
//   * the decoding schedule (f / g / leaf / rate-0 / repetition / combine operations, exactly the recursion of
//     polar/decoding.py:919-1005 including the fast-SCL shortcuts) is a flat list built once on the host and
//     interpreted by the kernel - no recursion, no divergence;
//   * only the L live paths are stored: at an information bit each path forks into (u=0, u=1), the 2L candidates are
//     ranked by a stable parallel rank (position order breaks ties - the behaviour of the reference's sort +
//     duplicate), survivors whose parent also survives are cloned into the slots of dead parents; clones are lazy
//     (per-slot pointer tables say which slot holds a stage's data);
//   * CRC-aided selection (penalty llr_max*k on CRC failures, first minimum) runs in the same kernel; the f-operation
//     is the exact boxplus softplus(x+y) - logsumexp(x,y) with the +-30 clip of the reference, in the defined float32
//     arithmetic of scl_math.h (bit-exact against oracle/polar_scl.c).
// Two engines share these definitions (polar_scl.h).  polar_scl_reg.hip - SC and list sizes 1..32 at n >= 64 - keeps
// the low tree stages in registers and is the one the benchmarks run (5.1 M SCL-8 decodes/s at C5).  The kernel in
// THIS file is the generic engine for everything else (short codes, other list sizes): every stage in LDS / L2
// scratch behind the pointer tables, 1.3 M SCL-8 decodes/s at C5 - instruction-issue and scalar-dispatch bound at
// ~2200 dependent operations per codeword with at most 64 independent work items each.
#include "common.h"
#include "bp_math.h"
#include "polar_scl.h"

namespace samd {

// ------------------------------------------------------------------ CRC
// crc_step: polar_scl.h
// out [N, k+len] = [bits, parity]  (or only the validity flag when check != 0: bits [N, k] incl. parity)
// The remainder is linear over GF(2): parity(u) = XOR over the set bits i of T[i], T[i] = remainder of x^(k-1-i+len).
// A workgroup builds T once in LDS (one serial chain of k shift steps, amortised over all the words it handles); a
// wave then owns a word: its lanes read the word's bits coalesced, XOR the table entries of the set bits and reduce.
// (The first version walked one word per thread: k dependent steps on reads 4k bytes apart - 0.4 TB/s at C5.)
__global__ __launch_bounds__(256) void crc_kernel(const float* __restrict__ bits, int64_t n_words, int k, uint32_t poly,
                                                  int len, int check, float* __restrict__ out) {
  extern __shared__ uint32_t crc_tab[];
  if (threadIdx.x == 0) {
    uint32_t reg = crc_step(0u, 1u, poly, len);              // a single one in the last position
    for (int i = k - 1; i >= 0; --i) {
      crc_tab[i] = reg;
      reg = crc_step(reg, 0u, poly, len);
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t w = (int64_t)blockIdx.x * 4 + wave; w < n_words; w += (int64_t)gridDim.x * 4) {
    const float* b = bits + w * k;
    float* o = out + w * (k + len);
    uint32_t acc = 0u;
    for (int i = lane; i < k; i += 64) {
      const float v = b[i];
      if ((int)v & 1) acc ^= crc_tab[i];
      if (!check) o[i] = v;
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) acc ^= (uint32_t)__shfl_xor((int)acc, s, 64);
    if (check) {
      if (lane == 0) out[w] = acc == 0u ? 1.f : 0.f;
    } else if (lane < len) {
      o[k + lane] = (float)((acc >> (len - 1 - lane)) & 1u);
    }
  }
}

// fallback for words longer than the LDS table can hold: one word per thread
__global__ __launch_bounds__(256) void crc_serial_kernel(const float* __restrict__ bits, int64_t n_words, int k,
                                                         uint32_t poly, int len, int check, float* __restrict__ out) {
  const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  const float* b = bits + w * k;
  uint32_t reg = 0;
  for (int i = 0; i < k; ++i) reg = crc_step(reg, (uint32_t)((int)b[i] & 1), poly, len);
  if (check) { out[w] = reg == 0 ? 1.f : 0.f; return; }
  float* o = out + w * (k + len);
  for (int i = 0; i < k; ++i) o[i] = b[i];
  for (int i = 0; i < len; ++i) o[k + i] = (float)((reg >> (len - 1 - i)) & 1u);
}

// T[i] of the parallel CRC above, for kernels that check a CRC themselves (Polar list decoders)
__global__ void crc_table_kernel(uint32_t* __restrict__ tab, int k, uint32_t poly, int len) {
  if (threadIdx.x != 0) return;
  uint32_t reg = crc_step(0u, 1u, poly, len);
  for (int i = k - 1; i >= 0; --i) {
    tab[i] = reg;
    reg = crc_step(reg, 0u, poly, len);
  }
}

// ------------------------------------------------------------------ Polar encoder
__global__ __launch_bounds__(256) void polar_encode_kernel(const float* __restrict__ u, const int32_t* __restrict__ info_pos,
                                                           const int32_t* __restrict__ out_idx, int k, int n,
                                                           int n_out, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char x[];
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < n; i += 256) x[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < k; i += 256) x[info_pos[i]] = (unsigned char)((int)u[(size_t)b * k + i] & 1);
  __syncthreads();
  for (int s = 1; s < n; s <<= 1) {            // x[d] ^= x[d + s] for every d whose bit s is clear
    for (int r = threadIdx.x; r < n / 2; r += 256) {
      const int d = 2 * r - (r & (s - 1));
      x[d] ^= x[d + s];
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n_out; i += 256) out[(size_t)b * n_out + i] = (float)x[out_idx[i]];
}

// ------------------------------------------------------------------ SC / SCL decoder (definitions: polar_scl.h)
// NT = threads per workgroup: 64 (one wave per codeword: the hardware barrier of a single-wave
// workgroup is free, which is what the ~14k dependent steps per codeword want) or 256.
// The decoder is a latency-bound dependent chain (measured ~10 cycles per wave instruction with one
// resident wave per SIMD), so throughput scales with the number of codewords resident on a CU.  Only
// the state that is touched by (almost) every op stays in LDS - the low LLR / partial-sum stages,
// the decided bits and the path bookkeeping (~4 KB at n = 1024, L = 8); the top G stages, touched by
// 2^(G+1) ops per decode, and the channel LLRs live in L2: 32 instead of 2 codewords per CU
// (measured on MI355X, n=1024 k=512 L=8: 137k -> 642k decodes/s).
template <int NT, typename R = float>
__global__ __launch_bounds__(NT) void polar_scl_kernel(SclArgsT<R> p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int n = p.n, L = p.L, tid = threadIdx.x;
  const int words = (n + 31) / 32;
  R* llr = reinterpret_cast<R*>(smem);                                      // [L][n/2^G] stage s < m-G at [2^s, 2^(s+1))
  const int hn = n >> p.gstages;
  auto stage = [&](int slot, int s) -> R* {           // the 2^s LLRs of stage s held by `slot`
    if (s >= p.m - p.gstages) return p.gscratch + ((size_t)blockIdx.x * L + slot) * (n - hn) + ((1 << s) - hn);
    return llr + (size_t)slot * hn + (1 << s);
  };
  // partial sums: ONE byte per (slot, stage position), bit 0 = left child's result, bit 1 = right
  // child's (halves the LDS of two byte banks: 3 instead of 2 codewords per CU at n = 1024, L = 8)
  unsigned char* beta = reinterpret_cast<unsigned char*>(llr + (size_t)L * hn);   // [L][n/2^G] (+ top stages in gbeta)
  auto bstage = [&](int slot, int s) -> unsigned char* {  // the 2^s partial-sum bytes of stage s held by `slot`
    if (s >= p.m - p.gstages) return p.gbeta + ((size_t)blockIdx.x * L + slot) * (n - hn) + ((1 << s) - hn);
    return beta + (size_t)slot * hn + (1 << s);
  };
  uint32_t* bits = reinterpret_cast<uint32_t*>(beta + (size_t)L * hn);            // [L][words] decided u bits
  R* pm = reinterpret_cast<R*>((reinterpret_cast<uintptr_t>(bits + (size_t)L * words) + sizeof(R) - 1) & ~(uintptr_t)(sizeof(R) - 1));   // [L] by position
  R* cand = pm + L;                                   // [2L] candidate metrics
  R* blk = cand + 2 * L;                              // [2L] block metrics (rate-0 / rep)
  int* order = reinterpret_cast<int*>(blk + 2 * L);       // [L]   position -> slot
  int* new_order = order + L;                             // [L]
  int* clone_src = new_order + L;                         // [L]   for new position: slot to copy from (-1: none)
  int* new_bit = clone_src + L;                           // [L]
  R* new_pm = reinterpret_cast<R*>(new_bit + L);  // [L]
  R* red = new_pm + L;                                // [256] reduction scratch
  // Lazy path copies.  All live paths execute the same op in lockstep and every op rewrites one whole
  // (array, stage) region of every path, so a path always writes into its OWN slot and records that in
  // its pointer table; a clone only inherits the tables (3 x 16 small ints) instead of the parent's n
  // LLRs and 2n partial sums.  No op reads the (array, stage) region it writes, so re-using a dead
  // slot that other tables still point to is safe: those tables are rewritten by the very op that
  // overwrites the region.
  unsigned char* lp = reinterpret_cast<unsigned char*>(red + 256);   // [L][16] slot of the stage-s LLRs
  unsigned char* bl = lp + (size_t)L * 16;                           // [L][16] slot of the stage-s left sums
  unsigned char* br = bl + (size_t)L * 16;                           // [L][16] slot of the stage-s right sums
  // the schedule is read with scalar loads one op ahead (wave-uniform index): no LDS copy

  for (int b = blockIdx.x; b < p.batch; b += gridDim.x) {
    const R* llr_ch = p.llr_in + (size_t)b * n;       // logits: negated where they are read (LLR = -logit)
    for (int i = tid; i < L * words; i += NT) bits[i] = 0u;
    for (int i = tid; i < L * 16; i += NT) { lp[i] = (unsigned char)(i >> 4); bl[i] = lp[i]; br[i] = lp[i]; }
    if (tid < L) { pm[tid] = tid == 0 ? (R)0 : (R)kPolarLlrMax; order[tid] = tid; }         // decoding.py:1029-1033
    __syncthreads();

    int next_rec = p.ops[0];
    int fused = 0, micro = 0;                               // a fused stage-1 node (OP_SUBTREE) and its pending parts
    for (int ip = 0;;) {
      int rec;
      if (micro == 0) {
        rec = next_rec;
        ++ip;
        next_rec = (ip < p.num_ops) ? p.ops[ip] : (int)OP_END;
        if ((rec & 7) == OP_SUBTREE) {
          // this engine expands stage-1 records only; a schedule built for the register engine (stage R / R + 1
          // records) must never be decoded here with silently wrong bits: fail the launch
          if (((rec >> 3) & 15) != 1) __builtin_trap();
          fused = rec; micro = 5;
        }
      }
      if (micro > 0) {
        // f, leaf (bit a2), g, leaf (bit a2 + 1), combine - the five operations pack_schedule fused
        const int part = 5 - micro, fa1 = (fused >> 7) & 1, fa2 = (fused >> 8) - 2048;
        --micro;
        rec = part == 0   ? (OP_F | (1 << 3))
              : part == 1 ? (OP_LEAF | ((fa2 + 2048) << 8))
              : part == 2 ? (OP_G | (1 << 3))
              : part == 3 ? (OP_LEAF | (1 << 7) | ((fa2 + 1 + 2048) << 8))
                          : (OP_COMBINE | (fa1 << 7));
      }
      const int op = rec & 7, a0 = (rec >> 3) & 15, a1 = (rec >> 7) & 1, a2 = (rec >> 8) - 2048;
      if (op == OP_END) break;
      if (op == OP_F || op == OP_G) {
        // a0 = stage s of the inputs (block of 2^s), outputs go to stage s-1; OP_G uses betaL[s-1]
        const int s = a0, half = 1 << (s - 1);
        for (int w = tid; w < L * half; w += NT) {
          const int pos = w >> (s - 1), j = w & (half - 1);     // half = 2^(s-1): no integer division
          const int slot = order[pos];
          const R* in = (s == p.m) ? llr_ch : stage(lp[slot * 16 + s], s);
          const R sg = (s == p.m) ? (R)-1 : (R)1;
          const R x = sg * in[j], y = sg * in[j + half];
          R r;
          if (op == OP_F) r = cn_op(x, y);
          else r = ((R)1 - (R)2 * (R)(bstage(bl[slot * 16 + s - 1], s - 1)[j] & 1)) * x + y;  // vn_op :707-714
          stage(slot, s - 1)[j] = r;
        }
        if (tid < L) lp[order[tid] * 16 + s - 1] = (unsigned char)order[tid];
        __syncthreads();
      } else if (op == OP_COMBINE) {
        // children results at stage s (a0) -> this node's result at stage s+1 on side a1
        const int s = a0, sz = 1 << s;
        for (int w = tid; w < L * sz; w += NT) {
          const int pos = w >> s, j = w & (sz - 1);
          const int slot = order[pos];
          const unsigned char l = bstage(bl[slot * 16 + s], s)[j] & 1;
          const unsigned char r = (bstage(br[slot * 16 + s], s)[j] >> 1) & 1;
          unsigned char* dst = bstage(slot, s + 1);
          const unsigned char keep = a1 ? 1 : 2;                    // the other side's bit stays
          dst[j] = (dst[j] & keep) | (unsigned char)((l ^ r) << a1);
          dst[sz + j] = (dst[sz + j] & keep) | (unsigned char)(r << a1);
        }
        if (tid < L) (a1 ? br : bl)[order[tid] * 16 + s + 1] = (unsigned char)order[tid];
        __syncthreads();
      } else {
        // ---- leaf / rate-0 / repetition node: a0 = stage s of the node, a1 = side, a2 = (last) bit index
        // (frozen leaf: a2 = -1-index)
        const int s = a0, sz = 1 << s;
        const bool info = (op == OP_REP) || (op == OP_LEAF && a2 >= 0);
        // block metrics of every live path: m0 = sum softplus(-l), m1 = sum softplus(+l)
        if (sz == 1) {
          if (tid < L) {                                      // one lane per path
            const R* in = (s == p.m) ? llr_ch : stage(lp[order[tid] * 16 + s], s);
            const R l = clamp_llr(((s == p.m) ? (R)-1 : (R)1) * in[0]);
            R s0, s1;
            softplus_pair(l, s0, s1);
            blk[tid] = s0;
            if (info) blk[L + tid] = s1;
          }
        } else {
          for (int pos = 0; pos < L; ++pos) {
            const R* in = (s == p.m) ? llr_ch : stage(lp[order[pos] * 16 + s], s);
            // defined summation order (oracle/polar_scl.c block_sum_f32): lane l accumulates terms l, l+64, l+128, ...
            // in ascending order, then the halving tree over the 64 lanes (xor butterfly: both operands of every
            // addition are the same two numbers on both lanes, so all lanes hold lane 0's tree)
            R m0 = (R)0, m1 = (R)0;
            for (int j = tid; j < sz; j += NT) {
              const R l = clamp_llr(((s == p.m) ? (R)-1 : (R)1) * in[j]);
              R s0, s1;
              softplus_pair(l, s0, s1);
              m0 += s0;
              m1 += s1;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { m0 += __shfl_xor(m0, o, 64); m1 += __shfl_xor(m1, o, 64); }
            if constexpr (NT > 64) {                          // fixed-order combination of the waves
              if ((tid & 63) == 0) { red[tid >> 6] = m0; red[8 + (tid >> 6)] = m1; }
              __syncthreads();
              m0 = (R)0; m1 = (R)0;
              for (int w = 0; w < NT / 64; ++w) { m0 += red[w]; m1 += red[8 + w]; }
              __syncthreads();
            }
            if (tid == 0) { blk[pos] = m0; blk[L + pos] = m1; }
          }
        }
        __syncthreads();
        if (!info) {
          // frozen leaf / rate-0: metric of the all-zero block, result zeros
          if (tid < L) pm[tid] += blk[tid];
          for (int w = tid; w < L * sz; w += NT) {
            const int pos = w >> s, j = w & (sz - 1);
            unsigned char* d = bstage(order[pos], s) + j;
            *d = *d & (a1 ? 1 : 2);
          }
          if (tid < L) (a1 ? br : bl)[order[tid] * 16 + s] = (unsigned char)order[tid];
          __syncthreads();
          continue;
        }
        // ---- fork: candidates c = u*L + pos
        if (p.sc_mode) {
          // PolarSCDecoder leaf: u = 0.5 (1 - sign(l)), exact zero -> 1 (decoding.py:208-212)
          if (tid == 0) {
            const R l = ((s == p.m) ? (R)-1 : (R)1) * ((s == p.m) ? llr_ch : stage(lp[order[0] * 16 + s], s))[0];
            new_order[0] = order[0]; clone_src[0] = -1; new_bit[0] = (l <= (R)0) ? 1 : 0; new_pm[0] = (R)0;
          }
        } else {
          int* rnk = reinterpret_cast<int*>(red);                                  // [2L] candidate ranks
          if (tid < 2 * L) cand[tid] = pm[tid % L] + blk[tid];
          __syncthreads();
          // stable rank of the 2L candidates (ties: lower candidate index first).  With one wave per codeword
          // every candidate gets g = 64 / 2L lanes, each counting a slice of the comparisons (xor-shuffle sum)
          if (NT == 64 && 2 * L <= 64 && (64 % (2 * L)) == 0) {
            const int g = 64 / (2 * L), c = tid / g, q = tid - c * g, per = (2 * L + g - 1) / g;
            const R me = cand[c];
            int rank = 0;
            for (int d = q * per; d < min((q + 1) * per, 2 * L); ++d) rank += (cand[d] < me || (cand[d] == me && d < c)) ? 1 : 0;
            for (int o = 1; o < g; o <<= 1) rank += __shfl_xor(rank, o, 64);
            if (q == 0) rnk[c] = rank;
          } else if (tid < 2 * L) {
            int rank = 0;
            const R me = cand[tid];
            for (int d = 0; d < 2 * L; ++d) rank += (cand[d] < me || (cand[d] == me && d < tid)) ? 1 : 0;
            rnk[tid] = rank;
          }
          __syncthreads();
          // survivors (rank < L) take position = rank; a parent's first survivor keeps its slot, the
          // j-th parent with two survivors clones its second child into the slot of the j-th parent
          // without survivors (prefix counts over LDS flags - no private arrays, no scratch)
          int* dead_slot = rnk + 2 * L;                                            // [L]
          if (NT == 64 && L <= 32) {
            // prefix counts by ballot: the j-th dead parent / the j-th parent with two survivors
            const bool act = tid < L;
            const int r0 = act ? rnk[tid] : 2 * L, r1 = act ? rnk[L + tid] : 2 * L;
            const bool dead = act && r0 >= L && r1 >= L, both = act && r0 < L && r1 < L;
            const unsigned long long md = __ballot(dead), mb = __ballot(both);
            const unsigned long long below = (1ull << tid) - 1ull;
            if (dead) dead_slot[__popcll(md & below)] = order[tid];
            __syncthreads();
            if (act) {
              const int q = tid, slot = order[q];
              if (r0 < L) { new_order[r0] = slot; clone_src[r0] = -1; new_bit[r0] = 0; new_pm[r0] = cand[q]; }
              if (r1 < L) {
                new_bit[r1] = 1; new_pm[r1] = cand[L + q];
                if (r0 < L) { new_order[r1] = dead_slot[__popcll(mb & below)]; clone_src[r1] = slot; }
                else { new_order[r1] = slot; clone_src[r1] = -1; }
              }
            }
          } else {
          if (tid < L) {
            const bool dead = rnk[tid] >= L && rnk[L + tid] >= L;
            if (dead) {
              int j = 0;
              for (int d = 0; d < tid; ++d) j += (rnk[d] >= L && rnk[L + d] >= L) ? 1 : 0;
              dead_slot[j] = order[tid];
            }
          }
          __syncthreads();
          if (tid < L) {
            const int q = tid, slot = order[q], r0 = rnk[q], r1 = rnk[L + q];
            if (r0 < L) { new_order[r0] = slot; clone_src[r0] = -1; new_bit[r0] = 0; new_pm[r0] = cand[q]; }
            if (r1 < L) {
              new_bit[r1] = 1; new_pm[r1] = cand[L + q];
              if (r0 < L) {
                int j = 0;
                for (int d = 0; d < q; ++d) j += (rnk[d] < L && rnk[L + d] < L) ? 1 : 0;
                new_order[r1] = dead_slot[j]; clone_src[r1] = slot;
              } else { new_order[r1] = slot; clone_src[r1] = -1; }
            }
          }
          }
        }
        __syncthreads();
        // clones: inherit the parent's pointer tables and decided bits (lazy copy, see above)
        // (sources are live parents, destinations dead slots: disjoint, so all clones copy concurrently)
        if (NT % L == 0) {
          const int per = NT / L, r = tid / per, i = tid - r * per;
          const int src = clone_src[r];
          if (src >= 0) {
            const int dst = new_order[r];
            for (int e = i; e < 16; e += per) {
              lp[dst * 16 + e] = lp[src * 16 + e];
              bl[dst * 16 + e] = bl[src * 16 + e];
              br[dst * 16 + e] = br[src * 16 + e];
            }
            for (int w2 = i; w2 < words; w2 += per) bits[(size_t)dst * words + w2] = bits[(size_t)src * words + w2];
          }
        } else {
          for (int r = 0; r < L; ++r) {
            const int src = clone_src[r];
            if (src < 0) continue;
            const int dst = new_order[r];
            if (tid < 16) {
              lp[dst * 16 + tid] = lp[src * 16 + tid];
              bl[dst * 16 + tid] = bl[src * 16 + tid];
              br[dst * 16 + tid] = br[src * 16 + tid];
            }
            for (int i = tid; i < words; i += NT) bits[(size_t)dst * words + i] = bits[(size_t)src * words + i];
          }
        }
        __syncthreads();
        // commit: order, metrics, decided bit (the node's only information bit is its last one), result
        const int bit_index = a2;
        if (tid < L) {
          order[tid] = new_order[tid];
          pm[tid] = new_pm[tid];
          if (new_bit[tid]) bits[(size_t)new_order[tid] * words + (bit_index >> 5)] |= 1u << (bit_index & 31);
        }
        __syncthreads();
        for (int w = tid; w < L * sz; w += NT) {
          const int pos = w >> s, j = w & (sz - 1);
          unsigned char* d = bstage(order[pos], s) + j;
          *d = (*d & (a1 ? 1 : 2)) | (unsigned char)(new_bit[pos] << a1);                          // all-u codeword
        }
        if (tid < L) (a1 ? br : bl)[order[tid] * 16 + s] = (unsigned char)order[tid];
        __syncthreads();
      }
    }
    // ---- final selection (decoding.py:1396-1419): CRC over the info bits of every path, penalty, first min
    if (tid < L) {
      const uint32_t* bw = bits + (size_t)order[tid] * words;
      R pen = (R)0;
      if (p.crc_len > 0) {
        uint32_t reg = 0;
        for (int i = 0; i < p.k; ++i) {
          const int src = p.iil_inv ? p.iil_inv[i] : i;
          const int pos = p.info_pos[src];
          reg = crc_step(reg, (bw[pos >> 5] >> (pos & 31)) & 1u, p.crc_poly, p.crc_len);
        }
        blk[tid] = reg == 0 ? (R)1 : (R)0;
        pen = reg == 0 ? (R)0 : (R)kPolarLlrMax * (R)p.k;
      }
      cand[tid] = pm[tid] + pen;
    }
    __syncthreads();
    // first minimum of the penalised metrics in the order of the final stable sort by path metric (:1391, 1415):
    // among equal penalised metrics (30 k + pm rounds to a grid of ~1e-3) the smaller path metric, then the
    // lower position
    int best = 0;
    for (int q = 1; q < L; ++q)
      if (cand[q] < cand[best] || (cand[q] == cand[best] && pm[q] < pm[best])) best = q;
    const uint32_t* bw = bits + (size_t)order[best] * words;
    for (int i = tid; i < p.k; i += NT) {
      const int pos = p.info_pos[i];
      p.u_hat[(size_t)b * p.k + i] = (R)((bw[pos >> 5] >> (pos & 31)) & 1u);
    }
    if (tid == 0 && p.crc_status) p.crc_status[b] = p.crc_len > 0 ? blk[best] : (R)1;
    __syncthreads();
  }
}

static size_t scl_lds_bytes(int n, int L, int num_ops, size_t real = sizeof(float)) {
  const size_t words = (n + 31) / 32;
  (void)num_ops;
  return (size_t)L * (n >> scl_gstages(n)) * (real + 1) + (size_t)L * words * 4 + (size_t)L * real * 5 +
         (size_t)L * (4 * 4 + real) + 256 * real + 3 * (size_t)L * 16 + 64;
}

// resident workgroups (= codewords in flight): as many as the list state of the engine that runs lets a CU hold
static int scl_grid(int batch, int n, int L, bool reg_engine) {
  const size_t lds = reg_engine ? scl_reg_lds_bytes(n, L) : scl_lds_bytes(n, L, 0);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  static CachedOpt per_cu_opt("SAMD_SCL_PER_CU");
  const size_t cap = (size_t)std::max<long>(1, per_cu_opt.get(32));   // one wave per workgroup: 8 per SIMD
  const size_t per_cu = std::min<size_t>(cap, std::max<size_t>(1, (160 * 1024) / lds));
  return (int)std::min<size_t>((size_t)batch, (size_t)cus * per_cu);
}

}  // namespace samd

using namespace samd;

extern "C" int samd_crc_f32(const float* bits, int64_t n_words, int k, uint32_t poly, int crc_len, int check,
                            float* out, void* stream) {
  SAMD_REQUIRE(bits && out && n_words >= 0 && k > 0 && crc_len > 0 && crc_len <= 32, "bad argument");
  if (n_words == 0) return SAMD_OK;
  if ((size_t)k * sizeof(uint32_t) <= 128 * 1024) {
    SAMD_SET_MAX_LDS(crc_kernel, 160 * 1024);
    const unsigned grid = (unsigned)std::min<int64_t>((n_words + 3) / 4, 256 * 8);
    hipLaunchKernelGGL(crc_kernel, dim3(grid), dim3(256), (size_t)k * sizeof(uint32_t), (hipStream_t)stream, bits, n_words, k,
                       poly, crc_len, check, out);
  } else {
    hipLaunchKernelGGL(crc_serial_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bits,
                       n_words, k, poly, crc_len, check, out);
  }
  return launch_status();
}

extern "C" int samd_polar_encode_f32(const float* u, const int32_t* info_pos, const int32_t* out_idx, int batch, int k,
                                     int n, int n_out, float* out, void* stream) {
  SAMD_REQUIRE(u && info_pos && out_idx && out && batch > 0 && k > 0 && n >= 2 && (n & (n - 1)) == 0 && n <= 65536,
               "bad argument");
  hipLaunchKernelGGL(polar_encode_kernel, dim3(batch), dim3(256), n, (hipStream_t)stream, u, info_pos, out_idx, k, n,
                     n_out, out);
  return launch_status();
}

extern "C" int samd_polar_scl_register_stages(int n, int list_size, int sc_mode) {
  if (n < 8 || (n & (n - 1)) != 0 || list_size < 1) return -1;
  return scl_reg_stages(n, list_size, sc_mode);
}

extern "C" size_t samd_polar_scl_workspace_bytes(int batch, int n, int list_size) {
  if (batch <= 0 || n < 8 || list_size < 1) return 0;
  // float + byte scratch of the top stages: n - n/2^G entries per slot, rounded up to n
  // (the engine is chosen at decode time - sc_mode is not known here: room for either)
  const int grid = std::max(scl_grid(batch, n, list_size, false),
                            scl_reg_supported(n, list_size, 0) ? scl_grid(batch, n, list_size, true) : 0);
  return (size_t)grid * list_size * (size_t)n * (sizeof(float) + 1) + 512 + (size_t)n * sizeof(uint32_t) + 256;   // + CRC table
}

static int polar_scl_decode_f32(const float* llr, const int32_t* src_a, const int32_t* src_b, int n_in, float rm_fill, const int32_t* ops,
                                int num_ops, const int32_t* info_pos, const int32_t* iil_inv, int batch, int n, int k, int list_size,
                                int sc_mode, uint32_t crc_poly, int crc_len, float* u_hat, float* crc_status, void* workspace,
                                size_t workspace_bytes, void* stream);

extern "C" int samd_polar_scl_decode_f32(const float* llr, const int32_t* ops, int num_ops, const int32_t* info_pos,
                                         const int32_t* iil_inv, int batch, int n, int k, int list_size, int sc_mode,
                                         uint32_t crc_poly, int crc_len, float* u_hat, float* crc_status,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  return polar_scl_decode_f32(llr, nullptr, nullptr, 0, 0.f, ops, num_ops, info_pos, iil_inv, batch, n, k, list_size, sc_mode, crc_poly,
                              crc_len, u_hat, crc_status, workspace, workspace_bytes, stream);
}

// Polar5GDecoder.call (polar/decoding.py:1999-2086) with its rate recovery (:2018-2052) INSIDE the decoder's channel-LLR load:
// llr [batch, n_in] as received; src_a / src_b DEVICE int32[n] (src_b nullable): position i of the mother code of length n reads
// llr[src_a[i]] (-1: 0, punctured; -2: -rm_fill, shortened) + llr[src_b[i]] (repetition; -1: nothing).  Everything else as
// samd_polar_scl_decode_f32.  SAMD_ERR_UNSUPPORTED when the (n, list_size, sc_mode) runs the generic engine (the host then
// gathers and calls samd_polar_scl_decode_f32).
extern "C" int samd_polar5g_scl_decode_f32(const float* llr, int n_in, const int32_t* src_a, const int32_t* src_b, float rm_fill,
                                           const int32_t* ops, int num_ops, const int32_t* info_pos, const int32_t* iil_inv, int batch,
                                           int n, int k, int list_size, int sc_mode, uint32_t crc_poly, int crc_len, float* u_hat,
                                           float* crc_status, void* workspace, size_t workspace_bytes, void* stream) {
  SAMD_REQUIRE(src_a && n_in > 0, "bad argument");
  if (!scl_reg_supported(n, list_size, sc_mode)) {
    set_error("rate recovery inside the decoder: register engine only");
    return SAMD_ERR_UNSUPPORTED;
  }
  return polar_scl_decode_f32(llr, src_a, src_b, n_in, rm_fill, ops, num_ops, info_pos, iil_inv, batch, n, k, list_size, sc_mode, crc_poly,
                              crc_len, u_hat, crc_status, workspace, workspace_bytes, stream);
}

static int polar_scl_decode_f32(const float* llr, const int32_t* src_a, const int32_t* src_b, int n_in, float rm_fill, const int32_t* ops,
                                int num_ops, const int32_t* info_pos, const int32_t* iil_inv, int batch, int n, int k, int list_size,
                                int sc_mode, uint32_t crc_poly, int crc_len, float* u_hat, float* crc_status, void* workspace,
                                size_t workspace_bytes, void* stream) {
  SAMD_REQUIRE(llr && ops && info_pos && u_hat && batch > 0, "bad argument");
  if (!workspace || workspace_bytes < samd_polar_scl_workspace_bytes(batch, n, list_size)) {
    set_error("workspace too small");
    return SAMD_ERR_WORKSPACE;
  }
  SAMD_REQUIRE(n >= 8 && (n & (n - 1)) == 0 && k >= 0 && k <= n, "n must be a power of two >= 8, 0 <= k <= n");
  SAMD_REQUIRE(list_size >= 1 && list_size <= 32 && (list_size & (list_size - 1)) == 0, "list_size must be a power of two <= 32");
  SAMD_REQUIRE(num_ops > 0 && n <= 1024, "schedule missing or n > 1024");
  const size_t lds = scl_lds_bytes(n, list_size, num_ops);
  if (lds > 160 * 1024) {
    set_error("list state does not fit in LDS (reduce list_size or n)");
    return SAMD_ERR_UNSUPPORTED;
  }
  SAMD_SET_MAX_LDS(polar_scl_kernel<64>, 160 * 1024);
  int m = 0;
  while ((1 << m) < n) ++m;
  const bool reg_engine = scl_reg_supported(n, list_size, sc_mode);
  const int grid = scl_grid(batch, n, list_size, reg_engine);
  float* gs = reinterpret_cast<float*>(align_up((size_t)workspace, 256));
  unsigned char* gb = reinterpret_cast<unsigned char*>(gs + (size_t)grid * list_size * n);
  // table of the CRC-aided selection (k <= n entries behind the scratch): T[i] = remainder of a single one at position i
  uint32_t* crc_tab = reinterpret_cast<uint32_t*>(align_up((size_t)(gb + (size_t)grid * list_size * n), 256));
  if (reg_engine && crc_len > 0 && k > 0)
    hipLaunchKernelGGL(crc_table_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, crc_tab, k, crc_poly, crc_len);
  SclArgs p{llr, u_hat, crc_status, ops, num_ops, info_pos, iil_inv, gs, gb, (reg_engine && crc_len > 0) ? crc_tab : nullptr,
            scl_gstages(n, reg_engine), batch, n, m, k, list_size,
            sc_mode, crc_len, crc_poly};
  p.src_a = src_a; p.src_b = src_b; p.n_in = n_in; p.rm_fill = rm_fill;
  // SC and list decoding with 1..32 paths of codes with n >= 64: the engine whose low stages live in registers (polar_scl_reg.hip)
  // (samd_polar_scl_register_stages() tells the host which engine runs, i.e. which subtree stage its schedule may use)
  if (reg_engine) return scl_reg_launch(p, grid, (hipStream_t)stream);
  // one wave per codeword: the block sums of rate-0 / repetition nodes are defined on 64 lanes (scl_math.h)
  hipLaunchKernelGGL(polar_scl_kernel<64>, dim3(grid), dim3(64), lds, (hipStream_t)stream, p);
  return launch_status();
}

// ---- precision = "double" (reference block.py:25-52): the generic engine on float64 (definitions: polar_scl.h)
extern "C" size_t samd_polar_scl_workspace_bytes_f64(int batch, int n, int list_size) {
  if (batch <= 0 || n < 8 || list_size < 1) return 0;
  const int grid = scl_grid(batch, n, list_size, false);
  return (size_t)grid * list_size * (size_t)n * (sizeof(double) + 1) + 512;
}

extern "C" int samd_polar_scl_decode_f64(const double* llr, const int32_t* ops, int num_ops, const int32_t* info_pos,
                                         const int32_t* iil_inv, int batch, int n, int k, int list_size, int sc_mode,
                                         uint32_t crc_poly, int crc_len, double* u_hat, double* crc_status, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  SAMD_REQUIRE(llr && ops && info_pos && u_hat && batch > 0, "bad argument");
  if (!workspace || workspace_bytes < samd_polar_scl_workspace_bytes_f64(batch, n, list_size)) {
    set_error("workspace too small");
    return SAMD_ERR_WORKSPACE;
  }
  SAMD_REQUIRE(n >= 8 && (n & (n - 1)) == 0 && k >= 0 && k <= n, "n must be a power of two >= 8, 0 <= k <= n");
  SAMD_REQUIRE(list_size >= 1 && list_size <= 32 && (list_size & (list_size - 1)) == 0, "list_size must be a power of two <= 32");
  SAMD_REQUIRE(num_ops > 0 && n <= 1024, "schedule missing or n > 1024");
  const size_t lds = scl_lds_bytes(n, list_size, num_ops, sizeof(double));
  if (lds > 160 * 1024) {
    set_error("list state does not fit in LDS (reduce list_size or n)");
    return SAMD_ERR_UNSUPPORTED;
  }
  auto kern = polar_scl_kernel<64, double>;
  SAMD_SET_MAX_LDS(kern, 160 * 1024);
  int m = 0;
  while ((1 << m) < n) ++m;
  // resident workgroups: the float32 engine's count (its LDS footprint is the smaller one; the hardware places as many as fit)
  const int grid = scl_grid(batch, n, list_size, false);
  double* gs = reinterpret_cast<double*>(align_up((size_t)workspace, 256));
  unsigned char* gb = reinterpret_cast<unsigned char*>(gs + (size_t)grid * list_size * n);
  SclArgsT<double> p{llr, u_hat, crc_status, ops, num_ops, info_pos, iil_inv, gs, gb, nullptr, scl_gstages(n, false), batch, n, m, k,
                     list_size, sc_mode, crc_len, crc_poly};
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64), lds, (hipStream_t)stream, p);
  return launch_status();
}
