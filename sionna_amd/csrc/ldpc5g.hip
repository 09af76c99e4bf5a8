// 5G-NR quasi-cyclic LDPC: code handle, encoder + rate matching, rate recovery, and the
// on-chip (LDS-resident) flooding min-sum decoder.
//
// Replaces (reference src/sionna/phy/fec/ldpc/):
//   LDPC5GEncoder.call / _encode_fast / _matmul_gather     encoding.py:559-668
//   LDPC5GDecoder.call (rate recovery, output mapping)      decoding.py:1427-1536
//   LDPCBPDecoder._bp_iter + cn_update_(offset_)minsum + vn_update_sum for 5G graphs
//                                                           decoding.py:416-524, 681-953
//
// MI355X design: everything here exploits the QC structure instead of the reference's
// flat gather lists.  A lifted block with shift s maps check (r, z) to variable
// (c, (z+s) mod Z); with one lane per lifted copy z, a base-graph edge is a ROTATED
// CONTIGUOUS access - conflict-free in LDS and describable by two scalars (c, s) that
// live in SGPRs.  One workgroup owns one codeword for its whole life:
//   * encoder: codeword bits in LDS, four rotated-XOR passes (RU method in closed form).
//   * decoder: all num_iter flooding iterations run inside ONE kernel; HBM traffic is the
//     compulsory 4n bytes in + 4k bytes out per codeword ("B_io"), not the
//     num_iter*(16E+4N) bytes of the HBM-resident formulation.  The message state fits
//     in 160 KiB LDS because min-sum check-node outputs are kept COMPRESSED: per check
//     node two magnitudes, the position of the unique minimum and the sign bits (12 B)
//     instead of one float per edge; v2c messages are never stored - they are recomputed
//     exactly as clip(x_tot[v] - c2v_e) from the VN totals (decoding.py:724-731).
//     The arithmetic and its order are those of oracle/ldpc_bp.py, so results are
//     bit-identical to the HBM-resident generic decoder.
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "ldpc5g.h"
#include "ldpc5g_jit.h"

namespace samd {

// ------------------------------------------------------------------ encoder
__global__ __launch_bounds__(256) void ldpc5g_encode_kernel(
    const float* __restrict__ bits, float* __restrict__ out, RateMatch p, int mb, int k_b, int bg, int s_a,
    int s_b, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ row_ent) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cw[];  // [n_ldpc] + lambda[4Z]
  const int z = p.z;
  const int n_ldpc = (mb + k_b) * z;
  unsigned char* lam = cw + n_ldpc;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* u = bits + (size_t)b * p.k;
  for (int i = tid; i < p.k_ldpc; i += 256)
    cw[i] = (i < p.k) ? (unsigned char)((int)u[i] & 1) : (unsigned char)0;   // filler = 0 (encoding.py:637)
  __syncthreads();
  // lambda_r = sum_j P(a_rj) s_j for the 4 core rows (columns < k_b only)
  for (int i = tid; i < 4 * z; i += 256) {
    const int r = i / z, zz = i - r * z;
    unsigned acc = 0;
    for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
      const int c = row_ent[e] & 0xFFFF, s = row_ent[e] >> 16;
      if (c >= k_b) break;
      int zi = zz + s; zi = zi >= z ? zi - z : zi;
      acc ^= cw[c * z + zi];
    }
    lam[i] = (unsigned char)acc;
  }
  __syncthreads();
  // p0 = P_B^-1 (lambda0+lambda1+lambda2+lambda3)   (first row of encoding.py:491-495)
  unsigned char* pa = cw + k_b * z;
  for (int zz = tid; zz < z; zz += 256) {
    int zi = zz - s_b; zi = zi < 0 ? zi + z : zi;
    pa[zz] = lam[zi] ^ lam[z + zi] ^ lam[2 * z + zi] ^ lam[3 * z + zi];
  }
  __syncthreads();
  for (int zz = tid; zz < z; zz += 256) {
    int zi = zz + s_a; zi = zi >= z ? zi - z : zi;
    const unsigned char ap0 = pa[zi];                       // (P_A p0)[z]
    const unsigned char p1 = lam[zz] ^ ap0;                 // core row 0:  P_A p0 + p1 = lambda0
    const unsigned char p3 = lam[3 * z + zz] ^ ap0;         // core row 3:  P_A p0 + p3 = lambda3
    const unsigned char p2 = (bg == 1) ? (lam[2 * z + zz] ^ p3)    // bg1 row 2: p2 + p3 = lambda2
                                       : (lam[z + zz] ^ p1);       // bg2 row 1: p1 + p2 = lambda1
    pa[z + zz] = p1; pa[2 * z + zz] = p2; pa[3 * z + zz] = p3;
  }
  __syncthreads();
  // extension rows: p_b = C1 s + C2 p_a  (encoding.py:579-581)
  for (int i = tid; i < (mb - 4) * z; i += 256) {
    const int r = 4 + i / z, zz = i % z;
    unsigned acc = 0;
    for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
      const int c = row_ent[e] & 0xFFFF, s = row_ent[e] >> 16;
      if (c >= k_b + 4) break;
      int zi = zz + s; zi = zi >= z ? zi - z : zi;
      acc ^= cw[c * z + zi];
    }
    cw[(k_b + 4) * z + i] = (unsigned char)acc;
  }
  __syncthreads();
  float* o = out + (size_t)b * p.n;
  for (int i = tid; i < p.n; i += 256) o[i] = (float)cw[short_to_full(p, out_to_short(p, i))];
}

// Bit-packed encoder for lifting sizes that are multiples of 32 (round 3; C2: Z = 128).  One WAVE per codeword, four
// codewords per workgroup, no workgroup barrier.  The codeword lives in LDS as n_ldpc / 32 words (bit j of word w of
// block c = variable node c Z + 32 w + j), 1 KB instead of 8.5 KB: a lifted block rotated by shift s is, per output word,
// two word reads and one v_alignbit (the bit offset s mod 32 is the same for every word of the block, the word offset
// is (w + s / 32) mod (Z / 32)) - the byte-per-bit kernel above spent 43 % of its LDS cycles on bank conflicts of byte
// stores.  Information bits are packed with one ballot per 64 floats; the rate-matched, interleaved output is one table
// look-up per element (out_idx[i] = position in the full codeword of output i: short_to_full(out_to_short(i)), built
// once per code on the host - the kernel above divides by the run-time modulation order for every output element).
// Same GF(2) arithmetic as above (RU method in closed form, encoding.py:559-668), hence the same bits; HBM traffic is
// the compulsory 4 (k + n) bytes per codeword.
__device__ __forceinline__ uint32_t enc_rot_word(const uint32_t* __restrict__ blk, int wq, int w, int s) {
  int a = w + (s >> 5);
  a = a >= wq ? a - wq : a;
  int a2 = a + 1;
  a2 = a2 >= wq ? 0 : a2;
  return __builtin_amdgcn_alignbit(blk[a2], blk[a], (uint32_t)(s & 31));
}

__global__ __launch_bounds__(256) void ldpc5g_encode_packed_kernel(
    const float* __restrict__ bits, float* __restrict__ out, RateMatch p, int batch, int mb, int k_b, int bg, int s_a,
    int s_b, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ row_ent,
    const uint16_t* __restrict__ out_idx, int dbg) {
  extern __shared__ __attribute__((aligned(16))) uint32_t cww[];      // 4 x ([n_ldpc / 32] codeword + [4 wq] lambda), tables
  const int z = p.z, wq = z >> 5;
  const int nw = (mb + k_b) * wq;                                    // words of the full codeword
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t* cw = cww + (size_t)wv * (nw + 4 * wq);
  uint32_t* lam = cw + nw;
  // the base graph (row pointers, entries) once per workgroup in LDS: the row loops below read an entry per edge
  int32_t* rp = reinterpret_cast<int32_t*>(cww + 4 * (size_t)(nw + 4 * wq));
  int32_t* re = rp + mb + 1;
  // ... and the output table: the output loop must not hold a global LOAD - loads and stores share the in-order vmcnt
  // counter, so waiting for a batch's table entries meant waiting for the previous batch's stores to reach memory
  // (measured at C2: input + rows 0.29 ms, output 0.42 ms, but 1.08 ms together)
  const int nnz_l = row_ptr[mb];
  uint16_t* oi = reinterpret_cast<uint16_t*>(re + nnz_l);
  for (int i = threadIdx.x; i <= mb; i += 256) rp[i] = row_ptr[i];
  for (int i = threadIdx.x; i < nnz_l; i += 256) re[i] = row_ent[i];
  for (int i = threadIdx.x; 2 * i < p.n; i += 256)
    reinterpret_cast<uint32_t*>(oi)[i] = reinterpret_cast<const uint32_t*>(out_idx)[i];   // table padded to an even length
  __syncthreads();
  row_ptr = rp;
  row_ent = re;
  const auto wave_sync = [] { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
  for (int b = blockIdx.x * 4 + wv; b < batch; b += gridDim.x * 4) {
    const float* u = bits + (size_t)b * p.k;
    // information bits -> words (filler bits = 0, encoding.py:637): one ballot per 64 floats
    // (16 loads in flight per lane before the first ballot: a wave is the only owner of its codeword's latency)
    for (int i0 = 0; i0 < p.k_ldpc && !(dbg & 1); i0 += 1024) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int i = i0 + 64 * j + lane;
        v[j] = i < p.k ? u[i] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(((int)v[j]) & 1);
        if (lane < 2 && i0 + 64 * j < p.k_ldpc) cw[((i0 + 64 * j) >> 5) + lane] = (uint32_t)(m >> (32 * lane));
      }
    }
    wave_sync();
    // lambda_r = sum_j P(a_rj) s_j for the 4 core rows (columns < k_b only): lane = (row, word)
    if (!(dbg & 2))
    for (int i = lane; i < 4 * wq; i += 64) {
      const int r = i / wq, w = i - r * wq;
      uint32_t acc = 0u;
      for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
        const int c = row_ent[e] & 0xFFFF, s = row_ent[e] >> 16;
        if (c >= k_b) break;
        acc ^= enc_rot_word(cw + c * wq, wq, w, s);
      }
      lam[i] = acc;
    }
    wave_sync();
    uint32_t* pa = cw + k_b * wq;
    // p0 = P_B^-1 (lambda0 + lambda1 + lambda2 + lambda3): rotation by -s_b
    if (lane < wq) {
      const int sb = s_b ? z - s_b : 0;
      pa[lane] = enc_rot_word(lam, wq, lane, sb) ^ enc_rot_word(lam + wq, wq, lane, sb) ^ enc_rot_word(lam + 2 * wq, wq, lane, sb) ^
                 enc_rot_word(lam + 3 * wq, wq, lane, sb);
    }
    wave_sync();
    if (lane < wq) {
      const uint32_t ap0 = enc_rot_word(pa, wq, lane, s_a);           // (P_A p0)
      const uint32_t p1 = lam[lane] ^ ap0;                            // core row 0:  P_A p0 + p1 = lambda0
      const uint32_t p3 = lam[3 * wq + lane] ^ ap0;                   // core row 3:  P_A p0 + p3 = lambda3
      const uint32_t p2 = (bg == 1) ? (lam[2 * wq + lane] ^ p3) : (lam[wq + lane] ^ p1);
      pa[wq + lane] = p1; pa[2 * wq + lane] = p2; pa[3 * wq + lane] = p3;
    }
    wave_sync();
    // extension rows: p_b = C1 s + C2 p_a  (encoding.py:579-581): lane = (row, word)
    if (!(dbg & 2))
    for (int i = lane; i < (mb - 4) * wq; i += 64) {
      const int r = 4 + i / wq, w = i % wq;
      uint32_t acc = 0u;
      for (int e = row_ptr[r]; e < row_ptr[r + 1]; ++e) {
        const int c = row_ent[e] & 0xFFFF, s = row_ent[e] >> 16;
        if (c >= k_b + 4) break;
        acc ^= enc_rot_word(cw + c * wq, wq, w, s);
      }
      cw[(k_b + 4) * wq + i] = acc;
    }
    wave_sync();
    float* o = out + (size_t)b * p.n;
    for (int i0 = 0; i0 < p.n && !(dbg & 4); i0 += 512) {
      unsigned t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + 64 * j + lane;
        t[j] = i < p.n ? oi[i] : 0u;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = i0 + 64 * j + lane;
        if (i < p.n) o[i] = (float)((cw[t[j] >> 5] >> (t[j] & 31)) & 1u);
      }
    }
    wave_sync();                                                     // the next codeword overwrites cw
  }
}

__global__ void ldpc5g_rate_recover_kernel(const float* __restrict__ llr, float* __restrict__ out, RateMatch p,
                                           float llr_max) {
  const int b = blockIdx.y;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < p.n_vn) out[(size_t)b * p.n_vn + v] = recover_llr(p, llr + (size_t)b * p.n, v, llr_max);
}

__global__ void ldpc5g_extract_kernel(const float* __restrict__ x_hat, float* __restrict__ out, RateMatch p) {
  const int b = blockIdx.y;
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o < p.n) out[(size_t)b * p.n + o] = x_hat[(size_t)b * p.n_vn + short_to_full(p, out_to_short(p, o))];
}

// ------------------------------------------------------------------ on-chip decoder
// LDS layout (floats): xt[n_vn] | llr[n_vn] | m1[n_cn] | m2[n_cn] | pk[n_cn] | counter
constexpr int kDecThreads = 1024;

template <bool OFFSET>
__global__ __launch_bounds__(kDecThreads) void ldpc5g_decode_kernel(
    const float* __restrict__ llr_in, float* __restrict__ out, RateMatch p, int n_cn, int batch, int num_iter,
    float llr_max, float offset, int hard_out, int return_infobits, const int32_t* __restrict__ row_ptr,
    const int32_t* __restrict__ row_ent, const int32_t* __restrict__ col_ptr, const int32_t* __restrict__ col_ent,
    const int32_t* __restrict__ cn_items, int n_cn_items, const int32_t* __restrict__ vn_items, int n_vn_items) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int z = p.z, n_vn = p.n_vn;
  float* xt = smem;
  float* llr = xt + n_vn;
  float* m1 = llr + n_vn;
  float* m2 = m1 + n_cn;
  unsigned* pk = reinterpret_cast<unsigned*>(m2 + n_cn);
  unsigned* counter = pk + n_cn;
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int kWaves = kDecThreads / 64;

  for (int b = blockIdx.x; b < batch; b += gridDim.x) {
    const float* row = llr_in + (size_t)b * p.n;
    for (int v = tid; v < n_vn; v += kDecThreads) {
      // decoding.py:552-565: clip, then logits -> LLR
      const float l = -1.f * clampf(recover_llr(p, row, v, llr_max), -llr_max, llr_max);
      llr[v] = l;
      xt[v] = l;
    }
    for (int c = tid; c < n_cn; c += kDecThreads) { m1[c] = 0.f; m2[c] = 0.f; pk[c] = 0u; }
    if (tid == 0) *counter = 0u;
    __syncthreads();

    unsigned base = 0;
    for (int it = 0; it < num_iter; ++it) {
      // ---------------- CN phase: one wave per (base row, 64 lifted copies)
      for (;;) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(counter, 1u);
        const int item = (int)(__builtin_amdgcn_readfirstlane(t) - base);
        if (item >= n_cn_items) break;
        const int desc = cn_items[item];
        const int r = desc & 0xFFFF;
        const int zz = (desc >> 16) * 64 + lane;
        const int cn = r * z + zz;
        const int e0 = row_ptr[r], d = row_ptr[r + 1] - e0;
        if (zz < z && cn < n_cn) {
          const float om1 = m1[cn], om2 = m2[cn];
          const unsigned opk = pk[cn];
          const int oidx = (int)(opk & 31u);
          float min1 = INFINITY, min2 = INFINITY;
          int idx = 0, cnt = 0;
          unsigned neg = 0;
          for (int i = 0; i < d; ++i) {
            const int ent = row_ent[e0 + i];
            const int c = ent & 0xFFFF, s = ent >> 16;
            int zi = zz + s; zi = zi >= z ? zi - z : zi;
            // old c2v of this edge, then v2c = clip(x_tot - c2v)   (decoding.py:724-731)
            float c2v = (i == oidx) ? om2 : om1;
            c2v = ((opk >> (5 + i)) & 1u) ? -c2v : c2v;
            const float v2c = clampf(-1.f * c2v + xt[c * z + zi], -llr_max, llr_max);
            neg |= (v2c < 0.f ? 1u : 0u) << i;                      // sign(0) := +1
            const float a = fabsf(v2c);
            if (a < min1) { min2 = min1; min1 = a; idx = i; cnt = 1; }
            else if (a == min1) { cnt++; }
            else if (a < min2) { min2 = a; }
          }
          // decoding.py:842-876: extrinsic magnitude at the minimum position(s)
          const float min_e = (cnt == 1) ? ((min2 - min1) + min1) : min1;
          float a1 = min1, a2 = min_e;
          if constexpr (OFFSET) { a1 = fmaxf(a1 - offset, 0.f); a2 = fmaxf(a2 - offset, 0.f); }
          else { a1 = fmaxf(a1, 0.f); a2 = fmaxf(a2, 0.f); }
          a1 = fminf(a1, llr_max); a2 = fminf(a2, llr_max);        // output clip (:906-908)
          const unsigned all = (d >= 32) ? 0xFFFFFFFFu : ((1u << d) - 1u);
          const unsigned sgn = (__popc(neg) & 1) ? (~neg & all) : neg;   // own sign x node sign
          m1[cn] = a1; m2[cn] = a2;
          pk[cn] = (unsigned)idx | (sgn << 5);
        }
      }
      base += (unsigned)(n_cn_items + kWaves);
      __syncthreads();
      // ---------------- VN phase: one wave per (base column, 64 lifted copies)
      for (;;) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(counter, 1u);
        const int item = (int)(__builtin_amdgcn_readfirstlane(t) - base);
        if (item >= n_vn_items) break;
        const int desc = vn_items[item];
        const int c = desc & 0xFFFF;
        const int zz = (desc >> 16) * 64 + lane;
        const int vn = c * z + zz;
        const int e0 = col_ptr[c], d = col_ptr[c + 1] - e0;
        if (zz < z && vn < n_vn) {
          float x = 0.f;
          for (int i = 0; i < d; ++i) {
            const int ent = col_ent[e0 + i];
            const int r = ent & 0xFF, s = (ent >> 8) & 0xFFF, pos = ent >> 20;
            int zi = zz - s; zi = zi < 0 ? zi + z : zi;
            const int cn = r * z + zi;
            if (cn < n_cn) {                                        // pruned rows carry no edge
              const unsigned q = pk[cn];
              float c2v = (pos == (int)(q & 31u)) ? m2[cn] : m1[cn];
              c2v = ((q >> (5 + pos)) & 1u) ? -c2v : c2v;
              x += c2v;                                             // ascending CN = edge order
            }
          }
          xt[vn] = x + llr[vn];                                     // unclipped x_tot (:716)
        }
      }
      base += (unsigned)(n_vn_items + kWaves);
      __syncthreads();
    }
    // ---------------- output (decoding.py:620-626, 1486-1531)
    if (return_infobits) {
      float* o = out + (size_t)b * p.k;
      for (int v = tid; v < p.k; v += kDecThreads) {
        const float x = clampf(xt[v], -llr_max, llr_max);
        o[v] = hard_out ? ((0.f >= x) ? 1.f : 0.f) : -1.f * x;
      }
    } else {
      float* o = out + (size_t)b * p.n;
      for (int i = tid; i < p.n; i += kDecThreads) {
        const float x = clampf(xt[short_to_full(p, out_to_short(p, i))], -llr_max, llr_max);
        o[i] = hard_out ? ((0.f >= x) ? 1.f : 0.f) : -1.f * x;
      }
    }
    __syncthreads();
  }
}

static RateMatch make_rm(const samd_ldpc5g* h) {
  return make_rate_match(h);
}

static size_t decode_lds_bytes(const samd_ldpc5g* h) {
  return ((size_t)2 * h->n_vn + (size_t)3 * h->n_cn) * 4 + 16;
}

}  // namespace samd

using namespace samd;

extern "C" int samd_ldpc5g_create(int bg, int z, const int16_t* rows, const int16_t* cols, const int16_t* shifts,
                                  int num_entries, int k, int n, int num_bits_per_symbol, int nb_pruned,
                                  samd_ldpc5g_t** out) {
  SAMD_REQUIRE(out && rows && cols && shifts, "null argument");
  SAMD_REQUIRE(bg == 1 || bg == 2, "bg must be 1 or 2");
  SAMD_REQUIRE(z >= 2 && z <= 384 && num_entries > 0, "bad lifting size");
  auto* h = new samd_ldpc5g();
  h->opt.capture();                                        // development switches: read once, here (options.h)
  h->host_only = host_only() ? 1 : 0;
  h->bg = bg; h->z = z; h->k = k; h->n = n; h->m_int = num_bits_per_symbol; h->nb_pruned = nb_pruned;
  h->mb = bg == 1 ? 46 : 42; h->nb = bg == 1 ? 68 : 52; h->k_b = bg == 1 ? 22 : 10;
  h->k_ldpc = h->k_b * z; h->n_ldpc = h->nb * z;
  h->n_vn = h->n_ldpc - nb_pruned; h->n_cn = h->mb * z - nb_pruned;
  h->nnz = num_entries;
  const bool ok = k > 0 && k <= h->k_ldpc && n > 0 && nb_pruned >= 0 && h->n_cn > 0 &&
                  2 * z + n <= h->n_vn - (h->k_ldpc - k) && (num_bits_per_symbol <= 0 || n % num_bits_per_symbol == 0);
  if (!ok) { delete h; set_error("inconsistent 5G LDPC parameters"); return SAMD_ERR_INVALID; }
  std::vector<std::vector<std::pair<int, int>>> by_row(h->mb), by_col(h->nb);
  for (int e = 0; e < num_entries; ++e) {
    const int r = rows[e], c = cols[e];
    if (r < 0 || r >= h->mb || c < 0 || c >= h->nb || shifts[e] < 0) { delete h; set_error("bad base-graph entry"); return SAMD_ERR_INVALID; }
    by_row[r].push_back({c, shifts[e] % z});
  }
  std::vector<int32_t> row_ptr(h->mb + 1, 0), row_ent, col_ptr(h->nb + 1, 0), col_ent;
  for (int r = 0; r < h->mb; ++r) {
    std::sort(by_row[r].begin(), by_row[r].end());
    for (size_t i = 0; i < by_row[r].size(); ++i) {
      row_ent.push_back(by_row[r][i].first | (by_row[r][i].second << 16));
      by_col[by_row[r][i].first].push_back({r, (int)i});      // rows visited ascending
    }
    row_ptr[r + 1] = (int32_t)row_ent.size();
    h->max_dc = std::max(h->max_dc, (int)by_row[r].size());
  }
  for (int c = 0; c < h->nb; ++c) {
    for (auto& rp : by_col[c]) {
      const int r = rp.first, pos = rp.second;
      const int s = by_row[r][pos].second;
      col_ent.push_back(r | (s << 8) | (pos << 20));
    }
    col_ptr[c + 1] = (int32_t)col_ent.size();
    h->max_dv = std::max(h->max_dv, (int)by_col[c].size());
  }
  auto find = [&](int r, int c) { for (auto& e : by_row[r]) if (e.first == c) return e.second; return -1; };
  h->s_a = find(0, h->k_b);
  h->s_b = find(bg == 1 ? 1 : 2, h->k_b);
  if (h->s_a < 0 || h->s_b < 0) { delete h; set_error("unexpected base-graph core structure"); return SAMD_ERR_INVALID; }
  // decoder work items, longest first (LPT order for the dynamic wave scheduler)
  const int chunks = (z + 63) / 64;
  std::vector<std::pair<int, int32_t>> ci, vi;
  for (int r = 0; r < h->mb; ++r)
    for (int q = 0; q < chunks; ++q)
      if (r * z + q * 64 < h->n_cn) ci.push_back({-(int)by_row[r].size(), r | (q << 16)});
  for (int c = 0; c < h->nb; ++c)
    for (int q = 0; q < chunks; ++q)
      if (c * z + q * 64 < h->n_vn) vi.push_back({-(int)by_col[c].size(), c | (q << 16)});
  std::stable_sort(ci.begin(), ci.end(), [](auto& a, auto& b) { return a.first < b.first; });
  std::stable_sort(vi.begin(), vi.end(), [](auto& a, auto& b) { return a.first < b.first; });
  std::vector<int32_t> cn_items, vn_items;
  for (auto& x : ci) cn_items.push_back(x.second);
  for (auto& x : vi) vn_items.push_back(x.second);
  h->n_cn_items = (int)cn_items.size(); h->n_vn_items = (int)vn_items.size();
  int rc = upload(&h->row_ptr, row_ptr.data(), row_ptr.size());
  if (rc == SAMD_OK) rc = upload(&h->row_ent, row_ent.data(), row_ent.size());
  if (rc == SAMD_OK) rc = upload(&h->col_ptr, col_ptr.data(), col_ptr.size());
  if (rc == SAMD_OK) rc = upload(&h->col_ent, col_ent.data(), col_ent.size());
  if (rc == SAMD_OK) rc = upload(&h->cn_items, cn_items.data(), cn_items.size());
  if (rc == SAMD_OK) rc = upload(&h->vn_items, vn_items.data(), vn_items.size());
  if (rc == SAMD_OK && z % 32 == 0 && h->n_ldpc < 65536) {
    // output position -> position in the full codeword (output interleaver and puncturing / filler removal folded)
    std::vector<uint16_t> oi(n + 2, 0);                    // (+ padding: the kernel copies the table in 32-bit words)
    for (int i = 0; i < n; ++i) {
      int t = i;
      if (h->m_int > 0) t = (i % h->m_int) * (n / h->m_int) + i / h->m_int;   // out_to_short
      const int uu = t + 2 * z;                                                // short_to_full
      oi[i] = (uint16_t)(uu < k ? uu : uu + (h->k_ldpc - k));
    }
    rc = upload(&h->enc_out_idx, oi.data(), oi.size());
  }
  if (rc == SAMD_OK) rc = build_onchip_tables(h, by_row);
  if (rc == SAMD_OK) rc = build_onchip_bp_tables(h, by_row);
  if (rc == SAMD_OK) rc = build_onchip_mss_tables(h, by_row);
  if (rc == SAMD_OK) rc = build_onchip_ly_tables(h, by_row);
  if (rc != SAMD_OK) { samd_ldpc5g_destroy(h); return rc; }
  build_jit_plan_general(h, by_row);                       // graph data for the generated kernel of any other code
  if (h->jit_plan) h->jit_state = new_jit_state();
  *out = h;
  return SAMD_OK;
}

extern "C" void samd_ldpc5g_destroy(samd_ldpc5g_t* h) {
  if (!h) return;
  (void)hipFree(h->row_ptr); (void)hipFree(h->row_ent); (void)hipFree(h->col_ptr); (void)hipFree(h->col_ent);
  (void)hipFree(h->cn_items); (void)hipFree(h->vn_items); (void)hipFree(h->enc_out_idx);
  free_onchip_tables(h);
  free_onchip_bp_tables(h);
  free_onchip_mss_tables(h);
  free_onchip_ly_tables(h);
  free_jit(h);
  delete h;
}

extern "C" int samd_ldpc5g_encode_f32(const samd_ldpc5g_t* h, const float* bits, float* out, int batch, void* stream) {
  SAMD_REQUIRE(h && bits && out && batch > 0, "bad argument");
  if (h->enc_out_idx && h->z % 32 == 0 && !h->opt.enc_bytes) {
    // lifting sizes that are multiples of 32: the bit-packed kernel, one wave per codeword
    const int wq = h->z / 32;
    const size_t lds_p = (4 * (size_t)((h->mb + h->k_b) * wq + 4 * wq) + (size_t)h->mb + 1 + (size_t)h->nnz) * sizeof(uint32_t) +
                         (size_t)(h->n + 2) * sizeof(uint16_t);
    if (lds_p > 64 * 1024) SAMD_SET_MAX_LDS(ldpc5g_encode_packed_kernel, 160 * 1024);
    // one codeword per wave and launch slot: a wave that went on to a second codeword would wait for its own output
    // stores before the next input arrives (loads and stores share the in-order vmcnt counter) - measured 1.12 ms
    // against 0.28 ms (input + rows) + 0.42 ms (output) for the separate phases at C2
    const int grid = h->opt.enc_persist ? std::min((batch + 3) / 4, 256 * 8 * 4) : (batch + 3) / 4;
    hipLaunchKernelGGL(ldpc5g_encode_packed_kernel, dim3(grid), dim3(256), lds_p, (hipStream_t)stream, bits, out, make_rm(h),
                       batch, h->mb, h->k_b, h->bg, h->s_a, h->s_b, h->row_ptr, h->row_ent, h->enc_out_idx,
                       h->opt.enc_dbg);
    return launch_status();
  }
  const size_t lds = (size_t)h->n_ldpc + 4 * (size_t)h->z;
  SAMD_SET_MAX_LDS(ldpc5g_encode_kernel, 64 * 1024);
  hipLaunchKernelGGL(ldpc5g_encode_kernel, dim3(batch), dim3(256), lds, (hipStream_t)stream, bits, out, make_rm(h),
                     h->mb, h->k_b, h->bg, h->s_a, h->s_b, h->row_ptr, h->row_ent);
  return launch_status();
}

extern "C" int samd_ldpc5g_rate_recover_f32(const samd_ldpc5g_t* h, const float* llr, float* out, int batch,
                                            float llr_max, void* stream) {
  SAMD_REQUIRE(h && llr && out && batch > 0, "bad argument");   // any batch: the loop below walks 65535-row chunks
  for (int b0 = 0; b0 < batch; b0 += 65535) {
    const int nb = std::min(65535, batch - b0);
    hipLaunchKernelGGL(ldpc5g_rate_recover_kernel, dim3((h->n_vn + 255) / 256, nb), dim3(256), 0, (hipStream_t)stream,
                       llr + (size_t)b0 * h->n, out + (size_t)b0 * h->n_vn, make_rm(h), llr_max);
  }
  return launch_status();
}

extern "C" int samd_ldpc5g_extract_codeword_f32(const samd_ldpc5g_t* h, const float* x_hat, float* out, int batch,
                                                void* stream) {
  SAMD_REQUIRE(h && x_hat && out && batch > 0, "bad argument");
  for (int b0 = 0; b0 < batch; b0 += 65535) {
    const int nb = std::min(65535, batch - b0);
    hipLaunchKernelGGL(ldpc5g_extract_kernel, dim3((h->n + 255) / 256, nb), dim3(256), 0, (hipStream_t)stream,
                       x_hat + (size_t)b0 * h->n_vn, out + (size_t)b0 * h->n, make_rm(h));
  }
  return launch_status();
}

// min-sum family: explicit messages (ldpc5g_onchip_ms.hip) when they fit in LDS, else the compressed
// check-node state (ldpc5g_onchip.hip, every 5G code).  SAMD_ONCHIP_COMPRESSED=1 forces the latter.
static bool use_explicit_minsum(const samd_ldpc5g* h) { return h->bp_ok && h->ms_cn_list && h->ms_vn_list && !h->opt.onchip_compressed; }
// ... or explicit messages with the last base rows' blocks in the L2 workspace row (ldpc5g_onchip_mss.hip)
// measured (tools/sweep_ldpc.py): up to about a quarter of the edges in L2 this beats the compressed state engine
// (+16 % at 4 %, +9 % at 26 %, even at 28 %); beyond that the L2 round trips of the VN phase dominate
static bool use_spill_minsum(const samd_ldpc5g* h) {
  return !h->bp_ok && h->sp_ok && (h->sp_spill_pct <= 27 || h->opt.force_spill) &&
         !h->opt.onchip_compressed && !h->opt.no_spill;
}
// boxplus rules on codes whose messages exceed LDS: the alternative is the HBM-resident engine, and the phi / tanh
// arithmetic (VALU bound) hides the L2 round trips - any spill share
static bool use_spill_boxplus(const samd_ldpc5g* h) { return !h->bp_ok && h->sp_ok && !h->opt.no_spill; }

extern "C" size_t samd_ldpc5g_decode_workspace_bytes(const samd_ldpc5g_t* h, int batch, int cn_mode) {
  // 0 when the whole state fits in LDS; larger codes keep part of it in this (L2-resident) scratch
  if (!h) return 0;
  const bool boxplus = cn_mode == SAMD_CN_BOXPLUS || cn_mode == SAMD_CN_BOXPLUS_PHI || cn_mode == SAMD_CN_BOXPLUS_PHI_FAST;
  if (boxplus && use_spill_boxplus(h)) return onchip_mss_workspace_bytes(h, batch);
  if (boxplus || use_explicit_minsum(h)) return onchip_bp_workspace_bytes(h, batch);
  // (the kernel generated for a code beyond LDS keeps its last base rows' messages in a workspace row per workgroup; the
  // generic engine behind it - the fall-back - has its own need: the larger of the two)
  const size_t jw = jit_workspace_bytes(h, batch, cn_mode);
  if (use_spill_minsum(h)) return std::max(jw, onchip_mss_workspace_bytes(h, batch));
  return std::max(jw, onchip_workspace_bytes(h, batch));
}

// ---- layered schedule (one sub-iteration per base row) on chip: ldpc5g_onchip_ly.hip
extern "C" int samd_ldpc5g_decode_layered_supported(const samd_ldpc5g_t* h, int cn_mode) {
  const bool rule = cn_mode == SAMD_CN_MINSUM || cn_mode == SAMD_CN_OFFSET_MINSUM || cn_mode == SAMD_CN_BOXPLUS_PHI ||
                    cn_mode == SAMD_CN_BOXPLUS_PHI_FAST;
  return (h && h->ly_ok && rule && !h->opt.no_onchip_layered) ? 1 : 0;
}

extern "C" size_t samd_ldpc5g_decode_layered_workspace_bytes(const samd_ldpc5g_t* h, int batch) {
  return h ? onchip_ly_workspace_bytes(h, batch) : 0;
}

extern "C" int samd_ldpc5g_decode_layered_f32(const samd_ldpc5g_t* h, const float* llr, float* out, int batch, int num_iter,
                                              int cn_mode, float llr_max, float offset, int hard_out, int return_infobits,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  SAMD_REQUIRE(h && llr && out && batch > 0 && num_iter >= 0, "bad argument");
  return launch_onchip_ly(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out, return_infobits, workspace,
                          workspace_bytes, (hipStream_t)stream);
}

extern "C" int samd_ldpc5g_decode_engine(const samd_ldpc5g_t* h, int cn_mode) {
  if (!h) return 0;
  if (cn_mode == SAMD_CN_BOXPLUS || cn_mode == SAMD_CN_BOXPLUS_PHI || cn_mode == SAMD_CN_BOXPLUS_PHI_FAST)
    return h->bp_ok ? 2 : (use_spill_boxplus(h) ? 3 : 0);
  if (cn_mode != SAMD_CN_MINSUM && cn_mode != SAMD_CN_OFFSET_MINSUM) return 0;
  if (use_explicit_minsum(h)) return 2;
  if (use_spill_minsum(h)) return 3;
  return (h->v2_ok || decode_lds_bytes(h) <= 160 * 1024) ? 1 : 0;
}

extern "C" int samd_ldpc5g_decode_f32(const samd_ldpc5g_t* h, const float* llr, float* out, int batch, int num_iter,
                                      int cn_mode, float llr_max, float offset, int hard_out, int return_infobits,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  SAMD_REQUIRE(h && llr && out && batch > 0 && num_iter >= 0, "bad argument");
  SAMD_REQUIRE(!h->host_only, "handle was built without a device (SAMD_HOST_ONLY)");
  if (cn_mode == SAMD_CN_BOXPLUS || cn_mode == SAMD_CN_BOXPLUS_PHI || cn_mode == SAMD_CN_BOXPLUS_PHI_FAST) {   // one float per edge in LDS
    SAMD_REQUIRE(llr_max >= 0.f, "bad argument");
    if (use_spill_boxplus(h))                                             // ... the last rows' messages in L2
      return launch_onchip_mss(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out, return_infobits,
                               workspace, workspace_bytes, (hipStream_t)stream);
    // the explicit-message engine with the boxplus node update (pair items, fused degree-1 columns, prefetched
    // descriptors - ldpc5g_onchip_ms.hip); SAMD_BP_ENGINE=1 keeps the first boxplus kernel (ldpc5g_onchip_bp.hip)
    if (use_explicit_minsum(h) && !h->opt.bp_engine) {
      const int rc = launch_onchip_ms(h, llr, out, batch, num_iter, cn_mode, llr_max, 0.f, hard_out, return_infobits,
                                      workspace, workspace_bytes, (hipStream_t)stream);
      if (rc != SAMD_ERR_UNSUPPORTED) return rc;
    }
    // (the first boxplus kernel has the defined phi only: "fast" permits the hardware transcendentals, it does not demand them)
    return launch_onchip_bp(h, llr, out, batch, num_iter, cn_mode == SAMD_CN_BOXPLUS_PHI_FAST ? SAMD_CN_BOXPLUS_PHI : cn_mode,
                            llr_max, hard_out, return_infobits, workspace, workspace_bytes, (hipStream_t)stream);
  }
  if (cn_mode != SAMD_CN_MINSUM && cn_mode != SAMD_CN_OFFSET_MINSUM) {
    set_error("unknown cn_mode");
    return SAMD_ERR_UNSUPPORTED;
  }
  const size_t lds = decode_lds_bytes(h);
  // the count-based duplicate-minimum test equals the reference's 1e5-sentinel sum test only
  // while node_degree * 2 * llr_max stays below the sentinel (decoding.py:865-872)
  if (h->max_dc > 27 || !(llr_max >= 0.f) || (double)h->max_dc * 2.0 * (double)llr_max >= 99999.0) {
    set_error("code / llr_max outside the on-chip decoder's envelope");
    return SAMD_ERR_UNSUPPORTED;
  }
  if (use_explicit_minsum(h)) {
    const int rc = launch_onchip_ms(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out, return_infobits,
                                    workspace, workspace_bytes, (hipStream_t)stream);
    if (rc != SAMD_ERR_UNSUPPORTED) return rc;
  } else {
    // the kernel generated for this code (ldpc5g_jit.cpp: any even lifting size whose messages fit LDS)
    const int rc = launch_onchip_jit(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out, return_infobits, workspace, workspace_bytes, stream);
    if (rc != SAMD_ERR_UNSUPPORTED) return rc;
  }
  if (use_spill_minsum(h)) {
    const int rc = launch_onchip_mss(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out, return_infobits,
                                     workspace, workspace_bytes, (hipStream_t)stream);
    if (rc != SAMD_ERR_UNSUPPORTED) return rc;
  }
  if (h->v2_ok && !h->opt.onchip_v1) {   // statically scheduled, unrolled engine
    const int rc = launch_onchip_v2(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out,
                                    return_infobits, workspace, workspace_bytes, (hipStream_t)stream);
    if (rc != SAMD_ERR_UNSUPPORTED) return rc;
  }
  if (lds > 160 * 1024) {
    set_error("code does not fit in LDS");
    return SAMD_ERR_UNSUPPORTED;
  }
  const bool off = (cn_mode == SAMD_CN_OFFSET_MINSUM);
  const void* fn = off ? (const void*)ldpc5g_decode_kernel<true> : (const void*)ldpc5g_decode_kernel<false>;
  SAMD_SET_MAX_LDS(fn, 160 * 1024);
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const size_t per_cu = std::max<size_t>(1, (160 * 1024) / lds);
  const int grid = (int)std::min<size_t>((size_t)batch, (size_t)cus * std::min<size_t>(per_cu, 2));
  const float off_v = off ? offset : 0.f;
#define SAMD_DEC_ARGS llr, out, make_rm(h), h->n_cn, batch, num_iter, llr_max, off_v, hard_out, return_infobits, \
                      h->row_ptr, h->row_ent, h->col_ptr, h->col_ent, h->cn_items, h->n_cn_items, h->vn_items, h->n_vn_items
  if (off) hipLaunchKernelGGL(ldpc5g_decode_kernel<true>, dim3(grid), dim3(kDecThreads), lds, (hipStream_t)stream, SAMD_DEC_ARGS);
  else hipLaunchKernelGGL(ldpc5g_decode_kernel<false>, dim3(grid), dim3(kDecThreads), lds, (hipStream_t)stream, SAMD_DEC_ARGS);
#undef SAMD_DEC_ARGS
  return launch_status();
}
