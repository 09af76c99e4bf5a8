// Random bit source and AWGN channel on a counter-based Philox4x32-10 stream.
//
// Replaces (reference src/sionna/phy/):
//   BinarySource.call         mapping.py:1350-1352     uniform {0,1} as float32
//   AWGN.call                 channel/awgn.py:63-78    y = x + sqrt(no) * w
//   complex_normal            utils/misc.py:19-54      w ~ CN(0,1), var 1/2 per real dim
//
// The reference draws from tf.random.Generator (stateful Philox); that stream cannot be
// reproduced without TensorFlow, so the build defines its own: element-indexed
// Philox4x32-10 keyed by (seed, call) - specification: oracle/utils.py.  Being counter
// based, a batch is reproducible independently of the launch geometry and shards across
// ranks by giving each rank its own seed.
#include "common.h"

namespace samd {

__global__ __launch_bounds__(256) void binary_source_kernel(uint64_t seed, uint64_t call, int64_t n,
                                                            float* __restrict__ out) {
  const int64_t nblk = (n + 3) / 4;
  for (int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < nblk;
       blk += (int64_t)gridDim.x * blockDim.x) {
    const uint4 r = philox_block(seed, call, (uint64_t)blk);
    const float4 v = make_float4((float)(r.x & 1u), (float)(r.y & 1u), (float)(r.z & 1u), (float)(r.w & 1u));
    const int64_t i = blk * 4;
    if (i + 3 < n && (((uintptr_t)out) & 15) == 0) {
      *reinterpret_cast<float4*>(out + i) = v;
    } else {
      if (i < n) out[i] = v.x;
      if (i + 1 < n) out[i + 1] = v.y;
      if (i + 2 < n) out[i + 2] = v.z;
      if (i + 3 < n) out[i + 3] = v.w;
    }
  }
}

// Box-Muller on two uniforms in (0,1)
__device__ __forceinline__ float2 box_muller(uint32_t a, uint32_t b) {
  const float r = sqrtf(-2.0f * logf(u01(a)));
  const float t = 6.283185307179586f * u01(b);
  float s, c;
  sincosf(t, &s, &c);
  return make_float2(r * c, r * s);
}

__global__ __launch_bounds__(256) void awgn_kernel(const float2* __restrict__ x, const float* __restrict__ no,
                                                   int64_t no_len, uint64_t seed, uint64_t call, int64_t n,
                                                   float2* __restrict__ y) {
  const int64_t nblk = (n + 1) / 2;
  const float sh = sqrtf(1.0f / 2.0f);                       // stddev per real dimension (misc.py:45-46)
  for (int64_t blk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; blk < nblk;
       blk += (int64_t)gridDim.x * blockDim.x) {
    const uint4 r = philox_block(seed, call, (uint64_t)blk);
    const float2 w0 = box_muller(r.x, r.y), w1 = box_muller(r.z, r.w);
    const int64_t i = blk * 2;
    {
      const float s = sqrtf(no_len == 1 ? no[0] : no[i]);
      const float2 xi = x[i];
      y[i] = make_float2(xi.x + (w0.x * sh) * s, xi.y + (w0.y * sh) * s);
    }
    if (i + 1 < n) {
      const float s = sqrtf(no_len == 1 ? no[0] : no[i + 1]);
      const float2 xi = x[i + 1];
      y[i + 1] = make_float2(xi.x + (w1.x * sh) * s, xi.y + (w1.y * sh) * s);
    }
  }
}

}  // namespace samd

using namespace samd;

static inline int grid_for(int64_t n, int block) {
  const int64_t g = (n + block - 1) / block;
  return (int)std::min<int64_t>(std::max<int64_t>(g, 1), 256 * 32);
}

extern "C" int samd_binary_source_f32(uint64_t seed, uint64_t call, int64_t n, float* out, void* stream) {
  SAMD_REQUIRE(out && n >= 0, "bad argument");
  if (n == 0) return SAMD_OK;
  hipLaunchKernelGGL(binary_source_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, seed,
                     call, n, out);
  return launch_status();
}

extern "C" int samd_awgn_c64(const float* x, const float* no, int64_t no_len, uint64_t seed, uint64_t call, int64_t n,
                             float* y, void* stream) {
  SAMD_REQUIRE(x && no && y && n >= 0, "bad argument");
  SAMD_REQUIRE(no_len == 1 || no_len == n, "no must be scalar or per element");
  if (n == 0) return SAMD_OK;
  hipLaunchKernelGGL(awgn_kernel, dim3(grid_for((n + 1) / 2, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float2*)x, no, no_len, seed, call, n, (float2*)y);
  return launch_status();
}
