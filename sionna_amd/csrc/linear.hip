// Generic binary linear block encoder.
//   LinearEncoder.call   /root/reference/src/sionna/phy/fec/linear/encoding.py:122-140
//                        (c = (u G) mod 2 as a float matmul + int_mod_2)
// GF(2) formulation: the information word is packed into 32-bit words once per codeword (ballot)
// and every codeword bit is the parity of popcount(u & g_col) over k/32 words - bit exact, no
// floating point, no dense [k, n] float matrix.  One workgroup per codeword; consecutive lanes
// produce consecutive codeword bits (coalesced stores); the packed columns (n * k/8 bytes) stay in L2.
#include "common.h"

namespace samd {
namespace {

__global__ __launch_bounds__(256) void gf2_encode_kernel(const float* __restrict__ u, const uint32_t* __restrict__ gcol,
                                                         int k, int n, int words, float* __restrict__ out) {
  extern __shared__ uint32_t uw[];                       // [words]
  const int b = blockIdx.x;
  const float* ub = u + (size_t)b * k;
  for (int w0 = 0; w0 < words * 32; w0 += 256) {
    const int i = w0 + threadIdx.x;
    const bool bit = i < k && (((int)ub[i]) & 1);
    const unsigned long long m = __ballot(bit);          // 64 lanes -> two words
    if ((threadIdx.x & 63) == 0) {
      const int w = i >> 5;
      if (w < words) uw[w] = (uint32_t)m;
      if (w + 1 < words) uw[w + 1] = (uint32_t)(m >> 32);
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += 256) {
    const uint32_t* g = gcol + (size_t)j * words;
    uint32_t acc = 0;
    for (int w = 0; w < words; ++w) acc ^= uw[w] & g[w];
    out[(size_t)b * n + j] = (float)(__popc(acc) & 1);
  }
}

}  // namespace
}  // namespace samd

using namespace samd;

extern "C" int samd_gf2_encode_f32(const float* u, const uint32_t* gm_cols, int64_t batch, int k, int n, float* out,
                                   void* stream) {
  SAMD_REQUIRE(u && gm_cols && out && batch >= 0 && k > 0 && n > 0, "bad argument");
  if (batch == 0) return SAMD_OK;
  const int words = (k + 31) / 32;
  SAMD_REQUIRE((size_t)words * 4 <= 64 * 1024, "k too large");
  SAMD_REQUIRE(batch < (1ll << 31), "batch too large");
  gf2_encode_kernel<<<(unsigned)batch, 256, (size_t)words * 4, (hipStream_t)stream>>>(u, gm_cols, k, n, words, out);
  return launch_status();
}
