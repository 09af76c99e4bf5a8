// Generic flooding belief-propagation LDPC decoder for an arbitrary Tanner graph.
//
// Replaces the TensorFlow op chain of LDPCBPDecoder.call / _bp_iter
// (reference src/sionna/phy/fec/ldpc/decoding.py:544-637, 416-524) and the node updates
// vn_update_sum (:681-732), cn_update_offset_minsum/minsum (:755-953), cn_update_tanh
// (:955-1043), cn_update_phi (:1045-1166).
//
// MI355X design (HBM-bound formulation, SURVEY.md section 8d "B_msg"):
//  * messages live in HBM batch-LAST, ONE buffer msg[E][Bs] shared by v2c and c2v: an
//    edge belongs to exactly one CN and one VN, so the CN pass overwrites the v2c rows of
//    its edges with c2v in place and the VN pass does the reverse.  Per iteration the
//    traffic is exactly read E + write E (CN pass) + read E + write E + read N_vn (VN
//    pass) = 16 E + 4 N_vn bytes per codeword - the algorithmic figure - while the state
//    is half the reference's (one [E,B] tensor instead of two).
//  * one 64-lane wave per node, each lane owns 4 consecutive batch columns (float4): a
//    wave moves 1 KiB per message row, every load/store is a fully coalesced dwordx4 and
//    the node's edge list is wave-uniform (scalar loads, no divergence).
//  * a node's messages are held in registers between the reduce and the extrinsic pass
//    (degree <= MAXD template bound), so every message is read once and written once.
//  * floating-point order is DEFINED and equals oracle/ldpc_bp.py: sequential over the
//    node's edges in edge order (VN-major edge numbering, ascending CN inside a VN /
//    ascending VN inside a CN); the library is compiled with -ffp-contract=off.
#include "common.h"
#include "accurate_math.h"
#include "bp_math.h"
#include "ldpc_graph.h"

#include <algorithm>
#include <vector>

namespace samd {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- CN pass.  Where the incoming v2c messages come from:
//  SRC_V2C      the shared message buffer holds v2c (flooding, after a VN pass)
//  SRC_LLR      v2c of iteration 0 = channel LLR of the edge's VN (decoding.py:571); aux = llr_t
//  SRC_DERIVED  the buffer holds c2v and aux = x_tot [N_vn][Bs]: v2c_e = clip(x_tot[v] - c2v_e)
//               (vn_update_sum, decoding.py:724-731) - scheduled decoding keeps c2v resident.
// node_list (nullable): the check nodes to update, num_cn = its length.
enum { SRC_V2C = 0, SRC_LLR = 1, SRC_DERIVED = 2 };

template <int MODE, int MAXD, int SRC>
__global__ __launch_bounds__(256) void cn_pass_kernel(
    float* __restrict__ msg, const float* __restrict__ aux, const int32_t* __restrict__ cn_ptr,
    const int32_t* __restrict__ cn_edge, const int32_t* __restrict__ cn_vn,
    const int32_t* __restrict__ node_list, int num_cn, int bs4, size_t stride, float llr_max, float offset) {
  const int slot = blockIdx.y * 4 + threadIdx.y;
  const int b4 = blockIdx.x * kWave + threadIdx.x;
  if (slot >= num_cn || b4 >= bs4) return;
  const int cn = node_list ? __builtin_amdgcn_readfirstlane(node_list[slot]) : slot;
  const int e0 = __builtin_amdgcn_readfirstlane(cn_ptr[cn]);
  const int d = __builtin_amdgcn_readfirstlane(cn_ptr[cn + 1]) - e0;
  float v0[MAXD], v1[MAXD], v2[MAXD], v3[MAXD];
#pragma unroll
  for (int i = 0; i < MAXD; ++i)
    if (i < d) {
      float4 x;
      if constexpr (SRC == SRC_LLR) x = ld4(aux + (size_t)cn_vn[e0 + i] * stride + 4 * (size_t)b4);
      else x = ld4(msg + (size_t)cn_edge[e0 + i] * stride + 4 * (size_t)b4);
      if constexpr (SRC == SRC_DERIVED) {
        const float4 t = ld4(aux + (size_t)cn_vn[e0 + i] * stride + 4 * (size_t)b4);
        x.x = clampf(-1.f * x.x + t.x, -llr_max, llr_max); x.y = clampf(-1.f * x.y + t.y, -llr_max, llr_max);
        x.z = clampf(-1.f * x.z + t.z, -llr_max, llr_max); x.w = clampf(-1.f * x.w + t.w, -llr_max, llr_max);
      }
      v0[i] = x.x; v1[i] = x.y; v2[i] = x.z; v3[i] = x.w;
    }
  cn_update_col<MODE, MAXD>(v0, d, llr_max, offset);
  cn_update_col<MODE, MAXD>(v1, d, llr_max, offset);
  cn_update_col<MODE, MAXD>(v2, d, llr_max, offset);
  cn_update_col<MODE, MAXD>(v3, d, llr_max, offset);
#pragma unroll
  for (int i = 0; i < MAXD; ++i)
    if (i < d)
      st4(msg + (size_t)cn_edge[e0 + i] * stride + 4 * (size_t)b4,
          make_float4(v0[i], v1[i], v2[i], v3[i]));
}

// High-degree fallback: one column per lane-slot, values re-read from memory (two passes).
template <int MODE, int SRC>
__global__ __launch_bounds__(256) void cn_pass_bigdeg_kernel(
    float* __restrict__ msg, const float* __restrict__ aux, const int32_t* __restrict__ cn_ptr,
    const int32_t* __restrict__ cn_edge, const int32_t* __restrict__ cn_vn,
    const int32_t* __restrict__ node_list, int num_cn, int bs, size_t stride, float llr_max, float offset) {
  const int slot = blockIdx.y * 4 + threadIdx.y;
  const int b = blockIdx.x * kWave + threadIdx.x;
  if (slot >= num_cn || b >= bs) return;
  const int cn = node_list ? node_list[slot] : slot;
  const int e0 = cn_ptr[cn], d = cn_ptr[cn + 1] - e0;
  auto load = [&](int i) -> float {
    if constexpr (SRC == SRC_LLR) return aux[(size_t)cn_vn[e0 + i] * stride + b];
    else if constexpr (SRC == SRC_DERIVED)
      return clampf(-1.f * msg[(size_t)cn_edge[e0 + i] * stride + b] + aux[(size_t)cn_vn[e0 + i] * stride + b],
                    -llr_max, llr_max);
    else return msg[(size_t)cn_edge[e0 + i] * stride + b];
  };
  // chunks of 32 values through the register kernel are not possible for a reduction over
  // the whole node, so the reductions are recomputed from memory.
  if constexpr (MODE == SAMD_CN_MINSUM || MODE == SAMD_CN_OFFSET_MINSUM) {
    float node_sign = 1.f, min1 = INFINITY;
    for (int i = 0; i < d; ++i) {
      const float x = clampf(load(i), -kLargeVal, kLargeVal);
      node_sign *= sign_nz(x);
      min1 = fminf(min1, fabsf(x));
    }
    float min2 = INFINITY, node_sum = 0.f;
    for (int i = 0; i < d; ++i) {
      const float t = fabsf(clampf(load(i), -kLargeVal, kLargeVal)) - min1;
      const float r = (t == 0.f) ? kLargeVal : t;
      min2 = fminf(min2, r);
      node_sum += r;
    }
    min2 = min2 + min1;
    node_sum = node_sum - (2.f * kLargeVal - 1.f);
    const float dm = 0.5f * (1.f - sgn3(node_sum));
    const float min_e = (1.f - dm) * min1 + dm * min2;
    for (int i = 0; i < d; ++i) {
      const float x = clampf(load(i), -kLargeVal, kLargeVal);
      const float t = fabsf(x) - min1;
      float m = (t == 0.f) ? min_e : min1;
      m = fmaxf(m - offset, 0.f);
      msg[(size_t)cn_edge[e0 + i] * stride + b] = clampf((sign_nz(x) * node_sign) * m, -llr_max, llr_max);
    }
  } else if constexpr (MODE == SAMD_CN_BOXPLUS_PHI || MODE == SAMD_CN_BOXPLUS_PHI_FAST) {
    float node_sign = 1.f, sum = 0.f;
    for (int i = 0; i < d; ++i) {
      const float x = load(i);
      node_sign *= sign_nz(x);
      sum += phi1_f32<MODE>(fabsf(x));
    }
    for (int i = 0; i < d; ++i) {
      const float x = load(i);
      const float e = -1.f * phi1_f32<MODE>(fabsf(x)) + sum;
      msg[(size_t)cn_edge[e0 + i] * stride + b] = clampf((sign_nz(x) * node_sign) * phi1_f32<MODE>(e), -llr_max, llr_max);
    }
  } else {
    float prod = 1.f;
    for (int i = 0; i < d; ++i) {
      float t = tanhf(load(i) / 2.f);
      prod *= (t == 0.f) ? 1e-12f : t;
    }
    const float ac = 1.f - 1e-7f;
    for (int i = 0; i < d; ++i) {
      float t = tanhf(load(i) / 2.f);
      t = (t == 0.f) ? 1e-12f : t;
      float e = (1.f / t) * prod;
      e = (fabsf(e) < 1e-7f) ? 0.f : e;
      e = clampf(e, -ac, ac);
      msg[(size_t)cn_edge[e0 + i] * stride + b] = clampf(2.f * atanhf(e), -llr_max, llr_max);
    }
  }
}

// ---- VN pass (vn_update_sum).  LAST: also write the marginals of VNs < out_rows.
template <int MAXD, bool LAST>
__global__ __launch_bounds__(256) void vn_pass_kernel(
    float* __restrict__ msg, const float* __restrict__ llr_t, float* __restrict__ xhat_t,
    const int32_t* __restrict__ vn_ptr, int num_vn, int out_rows, int bs4, size_t stride, float llr_max) {
  const int vn = blockIdx.y * 4 + threadIdx.y;
  const int b4 = blockIdx.x * kWave + threadIdx.x;
  if (vn >= num_vn || b4 >= bs4) return;
  const int e0 = __builtin_amdgcn_readfirstlane(vn_ptr[vn]);
  const int d = __builtin_amdgcn_readfirstlane(vn_ptr[vn + 1]) - e0;
  float4 c[MAXD];
  float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < MAXD; ++i)
    if (i < d) {
      c[i] = ld4(msg + (size_t)(e0 + i) * stride + 4 * (size_t)b4);
      x.x += c[i].x; x.y += c[i].y; x.z += c[i].z; x.w += c[i].w;
    }
  const float4 l = ld4(llr_t + (size_t)vn * stride + 4 * (size_t)b4);
  x.x += l.x; x.y += l.y; x.z += l.z; x.w += l.w;
#pragma unroll
  for (int i = 0; i < MAXD; ++i)
    if (i < d) {
      float4 o;
      o.x = clampf(-1.f * c[i].x + x.x, -llr_max, llr_max);
      o.y = clampf(-1.f * c[i].y + x.y, -llr_max, llr_max);
      o.z = clampf(-1.f * c[i].z + x.z, -llr_max, llr_max);
      o.w = clampf(-1.f * c[i].w + x.w, -llr_max, llr_max);
      st4(msg + (size_t)(e0 + i) * stride + 4 * (size_t)b4, o);
    }
  if constexpr (LAST) {
    if (vn < out_rows) {
      x.x = clampf(x.x, -llr_max, llr_max); x.y = clampf(x.y, -llr_max, llr_max);
      x.z = clampf(x.z, -llr_max, llr_max); x.w = clampf(x.w, -llr_max, llr_max);
      st4(xhat_t + (size_t)vn * stride + 4 * (size_t)b4, x);
    }
  }
}

template <bool LAST>
__global__ __launch_bounds__(256) void vn_pass_bigdeg_kernel(
    float* __restrict__ msg, const float* __restrict__ llr_t, float* __restrict__ xhat_t,
    const int32_t* __restrict__ vn_ptr, int num_vn, int out_rows, int bs, size_t stride, float llr_max) {
  const int vn = blockIdx.y * 4 + threadIdx.y;
  const int b = blockIdx.x * kWave + threadIdx.x;
  if (vn >= num_vn || b >= bs) return;
  const int e0 = vn_ptr[vn], d = vn_ptr[vn + 1] - e0;
  float x = 0.f;
  for (int i = 0; i < d; ++i) x += msg[(size_t)(e0 + i) * stride + b];
  x += llr_t[(size_t)vn * stride + b];
  for (int i = 0; i < d; ++i) {
    const size_t a = (size_t)(e0 + i) * stride + b;
    msg[a] = clampf(-1.f * msg[a] + x, -llr_max, llr_max);
  }
  if (LAST && vn < out_rows) xhat_t[(size_t)vn * stride + b] = clampf(x, -llr_max, llr_max);
}

// ---- [B, cols] batch-first logits  ->  [cols, Bs] clipped internal LLRs (= -logit).
__global__ __launch_bounds__(256) void prep_kernel(const float* __restrict__ in, float* __restrict__ llr_t,
                                                   int batch, int cols, size_t stride, float llr_max) {
  __shared__ float tile[64][65];
  const int c0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int r = ty; r < 64; r += 4) {
    const int b = b0 + r, c = c0 + tx;
    float v = 0.f;
    if (b < batch && c < cols) v = -1.f * clampf(in[(size_t)b * cols + c], -llr_max, llr_max);
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int c = c0 + r, b = b0 + tx;
    if (c < cols) llr_t[(size_t)c * stride + b] = tile[tx][r];   // b < Bs always (Bs % 64 == 0)
  }
}

// ---- x_hat [rows, Bs] -> out [B, rows]: hard decision (0 >= x) or logits (-x).
// clip: the scheduled path keeps the unclipped x_tot and clips here (INFINITY = identity).
__global__ __launch_bounds__(256) void finish_kernel(const float* __restrict__ xhat_t, float* __restrict__ out,
                                                     int batch, int rows, size_t stride, int hard, float clip) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int row = r0 + r;
    tile[r][tx] = (row < rows) ? xhat_t[(size_t)row * stride + b0 + tx] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int b = b0 + r, row = r0 + tx;
    if (b < batch && row < rows) {
      const float x = clampf(tile[tx][r], -clip, clip);
      out[(size_t)b * rows + row] = hard ? ((0.f >= x) ? 1.f : 0.f) : -1.f * x;
    }
  }
}

// grid.y of the row-indexed helper kernels below: the rows are walked with a gridDim.y stride, because large
// 5G codes (n_ldpc = 26112: ~121 k edges) and user PCMs exceed the 65535 limit of grid.y
static inline unsigned rows_grid(int rows) { return (unsigned)std::max(1, std::min(rows, 65535)); }

// state [E,B] (logit sign) <-> msg [E,Bs] (internal sign)
__global__ void state_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int batch,
                                  size_t src_stride, size_t dst_stride, int rows) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  for (int e = blockIdx.y; e < rows; e += gridDim.y)      // grid.y is capped at 65535 (rows_grid)
    dst[(size_t)e * dst_stride + b] = -1.f * src[(size_t)e * src_stride + b];
}

// v2c init when no iteration runs but the state is requested: msg[e] = llr_t[vn(e)]
__global__ void init_v2c_kernel(float* __restrict__ msg, const float* __restrict__ llr_t,
                                const int32_t* __restrict__ vn_ptr, int num_vn, int bs, size_t stride) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= bs) return;
  for (int vn = blockIdx.y; vn < num_vn; vn += gridDim.y) {
    const float l = llr_t[(size_t)vn * stride + b];
    for (int e = vn_ptr[vn]; e < vn_ptr[vn + 1]; ++e) msg[(size_t)e * stride + b] = l;
  }
}

// ---- scheduled decoding: x_tot[v] = (sum_e c2v_e) + llr[v] for the listed VNs (vn_update_sum
// with the c2v-resident buffer; v2c is derived on the fly by the CN pass).
template <int MAXD>
__global__ __launch_bounds__(256) void vn_total_kernel(
    const float* __restrict__ msg, const float* __restrict__ llr_t, float* __restrict__ xtot,
    const int32_t* __restrict__ vn_ptr, const int32_t* __restrict__ node_list, int n_nodes, int bs4, size_t stride) {
  const int slot = blockIdx.y * 4 + threadIdx.y;
  const int b4 = blockIdx.x * kWave + threadIdx.x;
  if (slot >= n_nodes || b4 >= bs4) return;
  const int vn = __builtin_amdgcn_readfirstlane(node_list[slot]);
  const int e0 = __builtin_amdgcn_readfirstlane(vn_ptr[vn]);
  const int d = __builtin_amdgcn_readfirstlane(vn_ptr[vn + 1]) - e0;
  float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (MAXD > 0) {
#pragma unroll
    for (int i = 0; i < MAXD; ++i)
      if (i < d) {
        const float4 c = ld4(msg + (size_t)(e0 + i) * stride + 4 * (size_t)b4);
        x.x += c.x; x.y += c.y; x.z += c.z; x.w += c.w;
      }
  } else {
    for (int i = 0; i < d; ++i) {
      const float4 c = ld4(msg + (size_t)(e0 + i) * stride + 4 * (size_t)b4);
      x.x += c.x; x.y += c.y; x.z += c.z; x.w += c.w;
    }
  }
  const float4 l = ld4(llr_t + (size_t)vn * stride + 4 * (size_t)b4);
  x.x += l.x; x.y += l.y; x.z += l.z; x.w += l.w;
  st4(xtot + (size_t)vn * stride + 4 * (size_t)b4, x);
}

// c2v = 0 for every check node that is not active in sub-iteration 0 (state_in start).
__global__ void zero_inactive_kernel(float* __restrict__ msg, const int32_t* __restrict__ cn_ptr,
                                     const int32_t* __restrict__ cn_edge, const int32_t* __restrict__ active,
                                     int num_cn, int bs, size_t stride) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= bs) return;
  for (int cn = blockIdx.y; cn < num_cn; cn += gridDim.y) {
    if (active[cn]) continue;
    for (int i = cn_ptr[cn]; i < cn_ptr[cn + 1]; ++i) msg[(size_t)cn_edge[i] * stride + b] = 0.f;
  }
}

// state_out of the scheduled path: msg_v2c[e] = clip(x_tot[v] - c2v_e), logit sign, [E,B].
__global__ void v2c_state_kernel(const float* __restrict__ msg, const float* __restrict__ xtot,
                                 const int32_t* __restrict__ vn_ptr, float* __restrict__ state, int num_vn, int batch,
                                 size_t stride, float llr_max) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  for (int vn = blockIdx.y; vn < num_vn; vn += gridDim.y) {
    const float x = xtot[(size_t)vn * stride + b];
    for (int e = vn_ptr[vn]; e < vn_ptr[vn + 1]; ++e)
      state[(size_t)e * batch + b] = -1.f * clampf(-1.f * msg[(size_t)e * stride + b] + x, -llr_max, llr_max);
  }
}

static void launch_vn_total(const samd_ldpc_graph* g, const float* msg, const float* llr_t, float* xtot,
                            const int32_t* node_list, int n_nodes, int bs, hipStream_t st) {
  const dim3 blk(kWave, 4);
  const int bs4 = bs / 4;
  const dim3 grid((bs4 + kWave - 1) / kWave, (n_nodes + 3) / 4);
#define SAMD_VT_LAUNCH(MAXD)                                                                          \
  hipLaunchKernelGGL((vn_total_kernel<MAXD>), grid, blk, 0, st, msg, llr_t, xtot, g->vn_ptr, node_list, \
                     n_nodes, bs4, (size_t)bs)
  if (g->max_dv <= 8) SAMD_VT_LAUNCH(8);
  else if (g->max_dv <= 32) SAMD_VT_LAUNCH(32);
  else SAMD_VT_LAUNCH(0);
#undef SAMD_VT_LAUNCH
}

template <int MODE, int SRC>
static void launch_cn(const samd_ldpc_graph* g, float* msg, const float* aux, const int32_t* node_list, int n_nodes,
                      int bs, float llr_max, float offset, hipStream_t st) {
  const dim3 blk(kWave, 4);
  const int bs4 = bs / 4;
  const dim3 grid((bs4 + kWave - 1) / kWave, (n_nodes + 3) / 4);
#define SAMD_CN_LAUNCH(MAXD)                                                                    \
  hipLaunchKernelGGL((cn_pass_kernel<MODE, MAXD, SRC>), grid, blk, 0, st, msg, aux, g->cn_ptr,   \
                     g->cn_edge, g->cn_vn, node_list, n_nodes, bs4, (size_t)bs, llr_max, offset)
  if (g->max_dc <= 8) SAMD_CN_LAUNCH(8);
  else if (g->max_dc <= 12) SAMD_CN_LAUNCH(12);
  else if (g->max_dc <= 20) SAMD_CN_LAUNCH(20);
  else if (g->max_dc <= 32) SAMD_CN_LAUNCH(32);
  else {
    const dim3 grid1((bs + kWave - 1) / kWave, (n_nodes + 3) / 4);
    hipLaunchKernelGGL((cn_pass_bigdeg_kernel<MODE, SRC>), grid1, blk, 0, st, msg, aux, g->cn_ptr, g->cn_edge,
                       g->cn_vn, node_list, n_nodes, bs, (size_t)bs, llr_max, offset);
  }
#undef SAMD_CN_LAUNCH
}

// node_list == nullptr: all check nodes.
template <int SRC>
static int launch_cn_mode(const samd_ldpc_graph* g, int mode, float* msg, const float* aux, int bs, float llr_max,
                          float offset, hipStream_t st, const int32_t* node_list = nullptr, int n_nodes = -1) {
  if (n_nodes < 0) n_nodes = g->num_cn;
  switch (mode) {
    case SAMD_CN_BOXPLUS: launch_cn<SAMD_CN_BOXPLUS, SRC>(g, msg, aux, node_list, n_nodes, bs, llr_max, offset, st); break;
    case SAMD_CN_BOXPLUS_PHI: launch_cn<SAMD_CN_BOXPLUS_PHI, SRC>(g, msg, aux, node_list, n_nodes, bs, llr_max, offset, st); break;
    case SAMD_CN_BOXPLUS_PHI_FAST: launch_cn<SAMD_CN_BOXPLUS_PHI_FAST, SRC>(g, msg, aux, node_list, n_nodes, bs, llr_max, offset, st); break;
    case SAMD_CN_MINSUM: launch_cn<SAMD_CN_MINSUM, SRC>(g, msg, aux, node_list, n_nodes, bs, llr_max, 0.f, st); break;
    case SAMD_CN_OFFSET_MINSUM: launch_cn<SAMD_CN_OFFSET_MINSUM, SRC>(g, msg, aux, node_list, n_nodes, bs, llr_max, offset, st); break;
    default: set_error("unknown cn_mode"); return SAMD_ERR_INVALID;
  }
  return SAMD_OK;
}

template <bool LAST>
static void launch_vn(const samd_ldpc_graph* g, float* msg, const float* llr_t, float* xhat_t, int out_rows,
                      int bs, float llr_max, hipStream_t st) {
  const dim3 blk(kWave, 4);
  const int bs4 = bs / 4;
  const dim3 grid((bs4 + kWave - 1) / kWave, (g->num_vn + 3) / 4);
#define SAMD_VN_LAUNCH(MAXD)                                                                          \
  hipLaunchKernelGGL((vn_pass_kernel<MAXD, LAST>), grid, blk, 0, st, msg, llr_t, xhat_t, g->vn_ptr, \
                     g->num_vn, out_rows, bs4, (size_t)bs, llr_max)
  if (g->max_dv <= 8) SAMD_VN_LAUNCH(8);
  else if (g->max_dv <= 16) SAMD_VN_LAUNCH(16);
  else if (g->max_dv <= 32) SAMD_VN_LAUNCH(32);
  else {
    const dim3 grid1((bs + kWave - 1) / kWave, (g->num_vn + 3) / 4);
    hipLaunchKernelGGL((vn_pass_bigdeg_kernel<LAST>), grid1, blk, 0, st, msg, llr_t, xhat_t, g->vn_ptr,
                       g->num_vn, out_rows, bs, (size_t)bs, llr_max);
  }
#undef SAMD_VN_LAUNCH
}

}  // namespace samd

using namespace samd;

extern "C" int samd_ldpc_graph_create(const int32_t* cn_idx, const int32_t* vn_idx, int num_edges, int num_cn,
                                      int num_vn, samd_ldpc_graph_t** out) {
  SAMD_REQUIRE(out != nullptr && cn_idx != nullptr && vn_idx != nullptr, "null argument");
  SAMD_REQUIRE(num_edges > 0 && num_cn > 0 && num_vn > 0, "empty graph");
  std::vector<int32_t> vn_ptr(num_vn + 1, 0), cn_ptr(num_cn + 1, 0);
  for (int e = 0; e < num_edges; ++e) {
    SAMD_REQUIRE(cn_idx[e] >= 0 && cn_idx[e] < num_cn && vn_idx[e] >= 0 && vn_idx[e] < num_vn, "edge index out of range");
    if (e > 0) {
      const bool ordered = vn_idx[e] > vn_idx[e - 1] || (vn_idx[e] == vn_idx[e - 1] && cn_idx[e] > cn_idx[e - 1]);
      SAMD_REQUIRE(ordered, "edges must be VN-major, ascending CN inside a VN, without duplicates");
    }
    vn_ptr[vn_idx[e] + 1]++;
    cn_ptr[cn_idx[e] + 1]++;
  }
  int max_dv = 0, max_dc = 0;
  for (int v = 0; v < num_vn; ++v) { max_dv = std::max(max_dv, vn_ptr[v + 1]); vn_ptr[v + 1] += vn_ptr[v]; }
  for (int c = 0; c < num_cn; ++c) { max_dc = std::max(max_dc, cn_ptr[c + 1]); cn_ptr[c + 1] += cn_ptr[c]; }
  // CN view: stable counting sort by CN keeps ascending VN inside a CN (= argsort(cn_idx, stable))
  std::vector<int32_t> fill(cn_ptr.begin(), cn_ptr.end() - 1), cn_edge(num_edges), cn_vn(num_edges);
  for (int e = 0; e < num_edges; ++e) {
    const int p = fill[cn_idx[e]]++;
    cn_edge[p] = e;
    cn_vn[p] = vn_idx[e];
  }
  auto* g = new samd_ldpc_graph();
  g->num_edges = num_edges; g->num_cn = num_cn; g->num_vn = num_vn;
  g->max_dc = max_dc; g->max_dv = max_dv;
  g->h_cn_ptr = cn_ptr; g->h_cn_vn = cn_vn;
  int rc = upload(&g->cn_ptr, cn_ptr.data(), cn_ptr.size());
  if (rc == SAMD_OK) rc = upload(&g->cn_edge, cn_edge.data(), cn_edge.size());
  if (rc == SAMD_OK) rc = upload(&g->cn_vn, cn_vn.data(), cn_vn.size());
  if (rc == SAMD_OK) rc = upload(&g->vn_ptr, vn_ptr.data(), vn_ptr.size());
  if (rc != SAMD_OK) { samd_ldpc_graph_destroy(g); return rc; }
  *out = g;
  return SAMD_OK;
}

extern "C" void samd_ldpc_graph_destroy(samd_ldpc_graph_t* g) {
  if (!g) return;
  (void)hipFree(g->cn_ptr); (void)hipFree(g->cn_edge); (void)hipFree(g->cn_vn); (void)hipFree(g->vn_ptr);
  delete g;
}

static inline size_t padded_batch(int batch) { return align_up((size_t)batch, 64); }

extern "C" size_t samd_ldpc_bp_workspace_bytes(const samd_ldpc_graph_t* g, int batch) {
  if (!g || batch <= 0) return 0;
  const size_t bs = padded_batch(batch);
  return ((size_t)g->num_edges + 2 * (size_t)g->num_vn) * bs * sizeof(float) + 256;
}

extern "C" int samd_ldpc_bp_decode_f32(const samd_ldpc_graph_t* g, const float* llr_in, float* out, int out_cols,
                                       float* state, int state_in, int state_out, int batch, int num_iter,
                                       int cn_mode, float llr_max, float offset, int hard_out, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  SAMD_REQUIRE(g && llr_in && out, "null argument");
  SAMD_REQUIRE(batch > 0 && num_iter >= 0, "bad batch / num_iter");
  SAMD_REQUIRE(out_cols > 0 && out_cols <= g->num_vn, "bad out_cols");
  SAMD_REQUIRE(!(state_in || state_out) || state, "state pointer missing");
  SAMD_REQUIRE(cn_mode >= 0 && cn_mode <= 4, "unknown cn_mode");
  if (workspace_bytes < samd_ldpc_bp_workspace_bytes(g, batch) || !workspace) {
    set_error("workspace too small");
    return SAMD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t bs = padded_batch(batch);
  float* base = reinterpret_cast<float*>(align_up((size_t)workspace, 256));
  float* msg = base;
  float* llr_t = msg + (size_t)g->num_edges * bs;
  float* xhat_t = llr_t + (size_t)g->num_vn * bs;

  {
    const dim3 grid((g->num_vn + 63) / 64, bs / 64);
    hipLaunchKernelGGL(prep_kernel, grid, dim3(256), 0, st, llr_in, llr_t, batch, g->num_vn, bs, llr_max);
  }
  if (state_in) {
    // padded columns stay whatever they are: they never reach an output
    const dim3 grid((batch + 255) / 256, rows_grid(g->num_edges));
    hipLaunchKernelGGL(state_copy_kernel, grid, dim3(256), 0, st, state, msg, batch, (size_t)batch, bs, g->num_edges);
  }
  const float* xsrc = xhat_t;
  if (num_iter == 0) {
    xsrc = llr_t;  // decoding.py:603-608: x_hat = clipped input
    if (state_out && !state_in) {
      const dim3 grid((bs + 255) / 256, rows_grid(g->num_vn));
      hipLaunchKernelGGL(init_v2c_kernel, grid, dim3(256), 0, st, msg, llr_t, g->vn_ptr, g->num_vn, (int)bs, bs);
    }
  }
  for (int it = 0; it < num_iter; ++it) {
    int rc;
    if (it == 0 && !state_in) rc = launch_cn_mode<SRC_LLR>(g, cn_mode, msg, llr_t, (int)bs, llr_max, offset, st);
    else rc = launch_cn_mode<SRC_V2C>(g, cn_mode, msg, llr_t, (int)bs, llr_max, offset, st);
    if (rc != SAMD_OK) return rc;
    if (it == num_iter - 1) launch_vn<true>(g, msg, llr_t, xhat_t, out_cols, (int)bs, llr_max, st);
    else launch_vn<false>(g, msg, llr_t, xhat_t, out_cols, (int)bs, llr_max, st);
  }
  {
    const dim3 grid((out_cols + 63) / 64, bs / 64);
    hipLaunchKernelGGL(finish_kernel, grid, dim3(256), 0, st, xsrc, out, batch, out_cols, bs, hard_out, INFINITY);
  }
  if (state_out) {
    const dim3 grid((batch + 255) / 256, rows_grid(g->num_edges));
    hipLaunchKernelGGL(state_copy_kernel, grid, dim3(256), 0, st, msg, state, batch, bs, (size_t)batch, g->num_edges);
  }
  return launch_status();
}


// ---------------------------------------------------------------- scheduled (layered) decoding
extern "C" int samd_ldpc_schedule_create(const samd_ldpc_graph_t* g, const int32_t* cn_schedule, int num_sub,
                                         int width, samd_ldpc_schedule_t** out) {
  SAMD_REQUIRE(g && cn_schedule && out, "null argument");
  SAMD_REQUIRE(num_sub > 0 && width > 0, "empty schedule");
  std::vector<int32_t> vn_list, vn_off(1, 0), mask(g->num_cn, 0);
  std::vector<char> seen(g->num_vn);
  for (int j = 0; j < num_sub; ++j) {
    std::fill(seen.begin(), seen.end(), 0);
    std::vector<char> cn_seen(g->num_cn, 0);
    for (int i = 0; i < width; ++i) {
      const int cn = cn_schedule[(size_t)j * width + i];
      SAMD_REQUIRE(cn >= 0 && cn < g->num_cn, "cn_schedule entry out of range");
      SAMD_REQUIRE(!cn_seen[cn], "cn_schedule row holds a check node twice");
      cn_seen[cn] = 1;
      if (j == 0) mask[cn] = 1;
      for (int e = g->h_cn_ptr[cn]; e < g->h_cn_ptr[cn + 1]; ++e) seen[g->h_cn_vn[e]] = 1;
    }
    for (int v = 0; v < g->num_vn; ++v)
      if (seen[v]) vn_list.push_back(v);
    vn_off.push_back((int32_t)vn_list.size());
  }
  auto* s = new samd_ldpc_schedule();
  s->num_sub = num_sub; s->width = width; s->num_cn = g->num_cn; s->vn_off = vn_off;
  int rc = upload(&s->cn_list, cn_schedule, (size_t)num_sub * width);
  if (rc == SAMD_OK) rc = upload(&s->vn_list, vn_list.data(), vn_list.size());
  if (rc == SAMD_OK) rc = upload(&s->first_mask, mask.data(), mask.size());
  if (rc != SAMD_OK) { samd_ldpc_schedule_destroy(s); return rc; }
  *out = s;
  return SAMD_OK;
}

extern "C" void samd_ldpc_schedule_destroy(samd_ldpc_schedule_t* s) {
  if (!s) return;
  (void)hipFree(s->cn_list); (void)hipFree(s->vn_list); (void)hipFree(s->first_mask);
  delete s;
}

extern "C" int samd_ldpc_bp_decode_scheduled_f32(const samd_ldpc_graph_t* g, const samd_ldpc_schedule_t* sched,
                                                 const float* llr_in, float* out, int out_cols, float* state,
                                                 int state_in, int state_out, int batch, int num_iter, int cn_mode,
                                                 float llr_max, float offset, int hard_out, void* workspace,
                                                 size_t workspace_bytes, void* stream) {
  SAMD_REQUIRE(g && sched && llr_in && out, "null argument");
  SAMD_REQUIRE(sched->num_cn == g->num_cn, "schedule was built for another graph");
  SAMD_REQUIRE(batch > 0 && num_iter >= 0, "bad batch / num_iter");
  SAMD_REQUIRE(out_cols > 0 && out_cols <= g->num_vn, "bad out_cols");
  SAMD_REQUIRE(!(state_in || state_out) || state, "state pointer missing");
  SAMD_REQUIRE(cn_mode >= 0 && cn_mode <= 4, "unknown cn_mode");
  if (workspace_bytes < samd_ldpc_bp_workspace_bytes(g, batch) || !workspace) {
    set_error("workspace too small");
    return SAMD_ERR_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t bs = padded_batch(batch);
  float* base = reinterpret_cast<float*>(align_up((size_t)workspace, 256));
  float* msg = base;                                  // c2v, resident for all edges
  float* llr_t = msg + (size_t)g->num_edges * bs;
  float* xtot = llr_t + (size_t)g->num_vn * bs;       // unclipped llr + sum c2v
  {
    const dim3 grid((g->num_vn + 63) / 64, bs / 64);
    hipLaunchKernelGGL(prep_kernel, grid, dim3(256), 0, st, llr_in, llr_t, batch, g->num_vn, bs, llr_max);
  }
  SAMD_HIP_CHECK(hipMemcpyAsync(xtot, llr_t, (size_t)g->num_vn * bs * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (state_in) {
    const dim3 grid((batch + 255) / 256, rows_grid(g->num_edges));
    hipLaunchKernelGGL(state_copy_kernel, grid, dim3(256), 0, st, state, msg, batch, (size_t)batch, bs, g->num_edges);
  } else {
    SAMD_HIP_CHECK(hipMemsetAsync(msg, 0, (size_t)g->num_edges * bs * sizeof(float), st));  // msg_c2v = 0 (:581)
  }
  bool v2c_from_state = state_in != 0;  // only the very first sub-iteration sees the given v2c
  for (int it = 0; it < num_iter; ++it)
    for (int j = 0; j < sched->num_sub; ++j) {
      const int32_t* cns = sched->cn_list + (size_t)j * sched->width;
      int rc;
      if (v2c_from_state) {
        rc = launch_cn_mode<SRC_V2C>(g, cn_mode, msg, xtot, (int)bs, llr_max, offset, st, cns, sched->width);
        const dim3 grid(((int)bs + 255) / 256, rows_grid(g->num_cn));
        hipLaunchKernelGGL(zero_inactive_kernel, grid, dim3(256), 0, st, msg, g->cn_ptr, g->cn_edge, sched->first_mask,
                           g->num_cn, (int)bs, bs);
        v2c_from_state = false;
      } else {
        rc = launch_cn_mode<SRC_DERIVED>(g, cn_mode, msg, xtot, (int)bs, llr_max, offset, st, cns, sched->width);
      }
      if (rc != SAMD_OK) return rc;
      launch_vn_total(g, msg, llr_t, xtot, sched->vn_list + sched->vn_off[j], sched->vn_off[j + 1] - sched->vn_off[j],
                      (int)bs, st);
    }
  {
    const dim3 grid((out_cols + 63) / 64, bs / 64);
    hipLaunchKernelGGL(finish_kernel, grid, dim3(256), 0, st, xtot, out, batch, out_cols, bs, hard_out, llr_max);
  }
  if (state_out) {
    if (num_iter == 0 && state_in) {
      // nothing ran: the state is returned as given
    } else {
      const dim3 grid((batch + 255) / 256, rows_grid(g->num_vn));
      hipLaunchKernelGGL(v2c_state_kernel, grid, dim3(256), 0, st, msg, xtot, g->vn_ptr, state, g->num_vn, batch, bs, llr_max);
    }
  }
  return launch_status();
}
