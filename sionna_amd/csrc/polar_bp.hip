// Belief-propagation decoder for Polar codes.
//
// Replaces (reference src/sionna/phy/fec/polar/decoding.py):
//   PolarBPDecoder.call / _decode_bp / _boxplus_tf     :1587-1771
//   Polar5GDecoder(dec_type="BP")                      :1896-1912 (the decoder it instantiates)
//
// The reference unrolls num_iter x 2 log2(n) stage updates into a TensorFlow graph; every stage update gathers two
// [batch, n/2] halves out of a [batch, n] message tensor, evaluates the boxplus in ~15 element-wise ops with as many
// temporaries, concatenates and gathers back, and keeps ALL num_iter x (log2 n + 1) message tensors alive in two
// TensorArrays (20 iterations at n = 1024: 2 x 220 tensors of batch x 4 KB).  Only the newest tensor of each column is
// ever read again.
//
// MI355X design: the factor graph of a codeword is (log2 n + 1) columns of n left-going (L) and n right-going (R)
// messages; the two columns that change - 76 KB in float32 at n = 1024 - live in the LDS of the workgroup that decodes
// the codeword; HBM / L2 see the channel values once per iteration (4 n bytes, the only stage that reads them) and the k
// decisions once (compulsory 4 n + 4 k bytes per codeword).
//   * one butterfly (two boxplus evaluations) per lane and stage; both evaluations go through the packed-fp32 pipe
//     together (v_pk_fma_f32 / v_pk_mul_f32: 5 packed exp / log calls per butterfly instead of 10 scalar ones);
//   * short codes pack several codewords into a workgroup (256 lanes / (n/2) butterflies) so that every lane has work
//     and the barriers between stages are shared;
//   * the two stage updates whose results nothing reads - R of the last column, and L of column 0 in every iteration but
//     the last - are skipped (same outputs, 2 of 2 log2(n) updates per iteration less);
//   * n > 1024: the same kernel with its message columns in a caller-owned L2 / HBM workspace (no LDS for 172 KB).
// Arithmetic: the literal form of _boxplus_tf (:1587-1603), log(1 + exp(x + y)) - log(exp(x) + exp(y)) on inputs clipped
// to +-19.3, float32, on the DEFINED exp / log of bp_math.h (= oracle/ldpc_bp.c spec_expf / spec_logf): the CPU oracle
// (oracle/polar_bp.py, math="spec") evaluates the same operations in the same order, so outputs are compared with
// array_equal; the oracle itself is pinned bit for bit (math="numpy") to the reference's source executed under the
// NumPy stand-in (tests/test_oracle_ref_exec_polar_bp.py).
#include "common.h"
#include "bp_math.h"

namespace samd {

constexpr float kPolarBpLlrMax = 19.3f;           // decoding.py:1527
constexpr int kPolarBpLdsMax = 160 * 1024;

// {boxplus(xa, ya), boxplus(xb, yb)}
__device__ __forceinline__ f32x2 polar_boxplus2(float xa, float ya, float xb, float yb) {
  const f32x2 x = {clampf(xa, -kPolarBpLlrMax, kPolarBpLlrMax), clampf(xb, -kPolarBpLlrMax, kPolarBpLlrMax)};
  const f32x2 y = {clampf(ya, -kPolarBpLlrMax, kPolarBpLlrMax), clampf(yb, -kPolarBpLlrMax, kPolarBpLlrMax)};
  const f32x2 one = {1.f, 1.f};
  const f32x2 num = spec_log2_f32(one + spec_exp2_f32(x + y));
  return num - spec_log2_f32(spec_exp2_f32(x) + spec_exp2_f32(y));
}

// precision = "double" (reference block.py:25-52): the same literal form in float64 on libm's exp / log
struct f64x2p { double x, y; };
__device__ __forceinline__ double polar_boxplus_f64(double x, double y) {
  x = fmax(fmin(x, (double)kPolarBpLlrMax), -(double)kPolarBpLlrMax);
  y = fmax(fmin(y, (double)kPolarBpLlrMax), -(double)kPolarBpLlrMax);
  return log(1.0 + exp(x + y)) - log(exp(x) + exp(y));
}
__device__ __forceinline__ f64x2p polar_boxplus2(double xa, double ya, double xb, double yb) {
  return f64x2p{polar_boxplus_f64(xa, ya), polar_boxplus_f64(xb, yb)};
}

// One stage update for the butterflies t, t + tpc, ... of a codeword.  Which of the two constant columns a stage reads is
// a template parameter, not a selected pointer: the LDS columns keep their ds_read instructions (a pointer chosen at run
// time between LDS and global memory turns every access into a flat load).
//   R_PRIOR: the right-going inputs are the priors (stage 0);  L_CH: the left-going inputs are the channel (stage S-1,
//   logits -> LLRs by lsgn = -1, decoding.py:1752)
template <bool R_PRIOR, typename R>
__device__ __forceinline__ void polar_bp_stage_lr(const R* Ls, const R* Rs, const R* __restrict__ prior, R* Ro,
                                                  int s, int half, int t, int tpc) {
  const int mask = (1 << s) - 1;
  for (int r = t; r < half; r += tpc) {
    const int i1 = 2 * r - (r & mask), i2 = i1 + (1 << s);
    const R l1 = Ls[i1], l2 = Ls[i2];
    const R r1 = R_PRIOR ? prior[i1] : Rs[i1], r2 = R_PRIOR ? prior[i2] : Rs[i2];
    const auto bp = polar_boxplus2(r1, l2 + r2, r1, l1);                // :1675-1676
    Ro[i1] = bp.x;
    Ro[i2] = bp.y + r2;
  }
}
template <bool R_PRIOR, bool L_CH, typename R>
__device__ __forceinline__ void polar_bp_stage_rl(const R* Ls, const R* __restrict__ ch, R lsgn, const R* Rs,
                                                  const R* __restrict__ prior, R* Lo, int s, int half, int t, int tpc) {
  const int mask = (1 << s) - 1;
  for (int r = t; r < half; r += tpc) {
    const int i1 = 2 * r - (r & mask), i2 = i1 + (1 << s);
    const R l1 = L_CH ? lsgn * ch[i1] : Ls[i1], l2 = L_CH ? lsgn * ch[i2] : Ls[i2];
    const R r1 = R_PRIOR ? prior[i1] : Rs[i1], r2 = R_PRIOR ? prior[i2] : Rs[i2];
    const auto bp = polar_boxplus2(l1, l2 + r2, r1, l1);                // :1706-1707
    Lo[i1] = bp.x;
    Lo[i2] = bp.y + l2;
  }
}

// GLOBAL = false: message columns of the workgroup's codewords in dynamic LDS; true: in `ws` (one slab per codeword).
// Column layout of one codeword (floats): L[0..S-1] (S columns of n), then R[1..S-1] (S - 1 columns).  The two columns
// that never change are NOT stored: L column S is the channel (-llr, read from global memory by the one stage per
// iteration that needs it) and R column 0 the priors (llr_max at frozen positions, 0 elsewhere: `prior`, shared by all
// codewords) - 76 KB instead of 84 KB at n = 1024, so TWO workgroups fit the 160 KB of a CU (4 waves per SIMD instead
// of 2: the barrier of one workgroup is hidden by the other; measured 1.14 M -> see DESIGN section 4).
template <bool GLOBAL, typename R = float>
__global__ __launch_bounds__(512) void polar_bp_kernel(const R* __restrict__ llr, const R* __restrict__ prior,
                                                       const int32_t* __restrict__ info_pos, int batch, int n, int S, int k,
                                                       int num_iter, int hard_out, R* __restrict__ out,
                                                       R* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char polar_bp_lds_raw[];
  R* polar_bp_lds = reinterpret_cast<R*>(polar_bp_lds_raw);
  const int half = n >> 1;
  const int tpc = half < (int)blockDim.x ? half : (int)blockDim.x;    // lanes per codeword
  const int W = (int)blockDim.x / tpc;                                  // codewords per workgroup
  const int w = (int)threadIdx.x / tpc, t = (int)threadIdx.x - w * tpc;
  const int64_t b = (int64_t)blockIdx.x * W + w;
  const bool live = b < batch;
  const size_t slab = (size_t)(2 * S - 1) * n;
  R* L = GLOBAL ? ws + (live ? b : 0) * slab : polar_bp_lds + (size_t)w * slab;     // columns 0 .. S-1
  R* Rc = L + (size_t)(S - 1) * n;                                                    // column c (1 .. S-1) at R + c n
  const R* ch = llr + (live ? b : 0) * (int64_t)n;

  for (int i = t; i < n; i += tpc)
    for (int c = 1; c < S; ++c) L[(size_t)c * n + i] = (R)0;            // "previous iteration" of the first sweep (:1655-1657)
  __syncthreads();

  for (int it = 0; it < num_iter; ++it) {
    // left to right (:1641-1683): R column s+1 from R column s and L column s+1; the last column's R is never read
    for (int s = 0; s + 1 < S; ++s) {
      const R* Ls = L + (size_t)(s + 1) * n;
      R* Ro = Rc + (size_t)(s + 1) * n;
      if (s == 0) polar_bp_stage_lr<true, R>(Ls, Ls, prior, Ro, s, half, t, tpc);
      else polar_bp_stage_lr<false, R>(Ls, Rc + (size_t)s * n, prior, Ro, s, half, t, tpc);
      __syncthreads();
    }
    // right to left (:1685-1713): L column s from L column s+1 and R column s; column 0 only feeds the decisions
    const int s_end = (it == num_iter - 1) ? 0 : 1;
    const R lsgn = live ? (R)-1 : (R)0;
    for (int s = S - 1; s >= s_end; --s) {
      const R* Ls = L + (size_t)(s + 1 == S ? 0 : s + 1) * n;     // (column S is the channel: Ls unused then)
      const R* Rs = Rc + (size_t)s * n;                              // (column 0 are the priors: Rs unused then)
      R* Lo = L + (size_t)s * n;
      if (s + 1 == S) {
        if (s == 0) polar_bp_stage_rl<true, true, R>(Ls, ch, lsgn, Rs, prior, Lo, s, half, t, tpc);
        else polar_bp_stage_rl<false, true, R>(Ls, ch, lsgn, Rs, prior, Lo, s, half, t, tpc);
      } else {
        if (s == 0) polar_bp_stage_rl<true, false, R>(Ls, ch, lsgn, Rs, prior, Lo, s, half, t, tpc);
        else polar_bp_stage_rl<false, false, R>(Ls, ch, lsgn, Rs, prior, Lo, s, half, t, tpc);
      }
      __syncthreads();
    }
  }
  if (!live) return;
  for (int j = t; j < k; j += tpc) {
    const R u = L[info_pos[j]];
    out[b * k + j] = hard_out ? (u > (R)0 ? (R)0 : (R)1) : (R)-1 * u;      // :1719-1723
  }
}

static inline int polar_bp_stages(int n) {
  int S = 0;
  while ((1 << S) < n) ++S;
  return S;
}

}  // namespace samd

using namespace samd;

extern "C" size_t samd_polar_bp_workspace_bytes(int batch, int n) {
  if (batch <= 0 || n < 2 || (n & (n - 1)) != 0) return 0;
  const size_t slab = (size_t)(2 * polar_bp_stages(n) - 1) * n * sizeof(float);
  return slab <= (size_t)kPolarBpLdsMax ? 0 : (size_t)batch * slab + 256;
}

extern "C" int samd_polar_bp_decode_f32(const float* llr, const float* prior, const int32_t* info_pos, int batch, int n,
                                        int k, int num_iter, int hard_out, float* out, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  SAMD_REQUIRE(llr && prior && info_pos && out && batch > 0, "bad argument");
  SAMD_REQUIRE(n >= 2 && (n & (n - 1)) == 0 && n <= (1 << 20) && k >= 0 && k <= n, "n must be a power of two, 0 <= k <= n");
  SAMD_REQUIRE(num_iter >= 1, "num_iter must be positive");
  const int S = polar_bp_stages(n);
  const size_t slab = (size_t)(2 * S - 1) * n * sizeof(float);
  const int half = n / 2;
  if (slab <= (size_t)kPolarBpLdsMax) {
    const int threads = half >= 512 ? 512 : 256;
    const int W = half < threads ? threads / half : 1;
    SAMD_SET_MAX_LDS(polar_bp_kernel<false>, kPolarBpLdsMax);
    hipLaunchKernelGGL(polar_bp_kernel<false>, dim3((batch + W - 1) / W), dim3(threads), W * slab, (hipStream_t)stream, llr,
                       prior, info_pos, batch, n, S, k, num_iter, hard_out, out, (float*)nullptr);
    return launch_status();
  }
  if (!workspace || workspace_bytes < samd_polar_bp_workspace_bytes(batch, n)) {
    set_error("workspace too small");
    return SAMD_ERR_WORKSPACE;
  }
  float* ws = reinterpret_cast<float*>(align_up((size_t)workspace, 256));
  hipLaunchKernelGGL(polar_bp_kernel<true>, dim3(batch), dim3(512), 0, (hipStream_t)stream, llr, prior, info_pos, batch, n, S,
                     k, num_iter, hard_out, out, ws);
  return launch_status();
}

// ---- precision = "double": the same kernel on float64 (message columns of n >= 1024 exceed the LDS and go to the workspace)
extern "C" size_t samd_polar_bp_workspace_bytes_f64(int batch, int n) {
  if (batch <= 0 || n < 2 || (n & (n - 1)) != 0) return 0;
  const size_t slab = (size_t)(2 * polar_bp_stages(n) - 1) * n * sizeof(double);
  const int half = n / 2, threads = half >= 512 ? 512 : 256, W = half < threads ? threads / half : 1;
  return (size_t)W * slab <= (size_t)kPolarBpLdsMax ? 0 : (size_t)batch * slab + 256;
}

extern "C" int samd_polar_bp_decode_f64(const double* llr, const double* prior, const int32_t* info_pos, int batch, int n, int k,
                                        int num_iter, int hard_out, double* out, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  SAMD_REQUIRE(llr && prior && info_pos && out && batch > 0, "bad argument");
  SAMD_REQUIRE(n >= 2 && (n & (n - 1)) == 0 && n <= (1 << 20) && k >= 0 && k <= n, "n must be a power of two, 0 <= k <= n");
  SAMD_REQUIRE(num_iter >= 1, "num_iter must be positive");
  const int S = polar_bp_stages(n);
  const size_t slab = (size_t)(2 * S - 1) * n * sizeof(double);
  const int half = n / 2;
  if (slab <= (size_t)kPolarBpLdsMax) {
    const int threads = half >= 512 ? 512 : 256;
    const int W = half < threads ? threads / half : 1;
    if ((size_t)W * slab <= (size_t)kPolarBpLdsMax) {
      auto kern = polar_bp_kernel<false, double>;
      SAMD_SET_MAX_LDS(kern, kPolarBpLdsMax);
      hipLaunchKernelGGL(kern, dim3((batch + W - 1) / W), dim3(threads), W * slab, (hipStream_t)stream, llr, prior, info_pos, batch, n, S,
                         k, num_iter, hard_out, out, (double*)nullptr);
      return launch_status();
    }
  }
  const size_t need = (size_t)batch * slab + 256;
  if (!workspace || workspace_bytes < need) {
    set_error("workspace too small");
    return SAMD_ERR_WORKSPACE;
  }
  double* ws = reinterpret_cast<double*>(align_up((size_t)workspace, 256));
  hipLaunchKernelGGL((polar_bp_kernel<true, double>), dim3(batch), dim3(512), 0, (hipStream_t)stream, llr, prior, info_pos, batch, n, S,
                     k, num_iter, hard_out, out, ws);
  return launch_status();
}
