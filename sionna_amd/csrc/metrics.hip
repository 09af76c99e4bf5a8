// Bit / block error counting.
//
// Replaces count_errors / count_block_errors (reference src/sionna/phy/utils/metrics.py:94-144)
// and the optional hard_decisions of sim_ber (utils/misc.py:713-714, 254-271).  The
// reference gathers the full b / b_hat tensors to one device and reduces there; here each
// GPU reduces its own [blocks, block_len] tensors to two int64 counters (wave shuffle
// reduction, one atomic per workgroup) and only those counters are all-reduced (RCCL).
#include "common.h"

namespace samd {

__global__ __launch_bounds__(256) void count_errors_kernel(const float* __restrict__ b, const float* __restrict__ bh,
                                                           int64_t num_blocks, int64_t block_len, int soft,
                                                           unsigned long long* __restrict__ counters) {
  // one wave per block (codeword); 4 waves per workgroup
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long bit_err = 0, blk_err = 0;
  for (int64_t blk = (int64_t)blockIdx.x * 4 + w; blk < num_blocks; blk += (int64_t)gridDim.x * 4) {
    const float* pb = b + blk * block_len;
    const float* ph = bh + blk * block_len;
    unsigned cnt = 0;
    for (int64_t i = lane; i < block_len; i += 64) {
      float h = ph[i];
      if (soft) h = (h > 0.f) ? 1.f : 0.f;                  // hard_decisions: strict > 0
      cnt += (pb[i] != h) ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    bit_err += cnt;
    blk_err += cnt ? 1u : 0u;
  }
  __shared__ unsigned long long s_bit[4], s_blk[4];
  if (lane == 0) { s_bit[w] = bit_err; s_blk[w] = blk_err; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long tb = s_bit[0] + s_bit[1] + s_bit[2] + s_bit[3];
    const unsigned long long tk = s_blk[0] + s_blk[1] + s_blk[2] + s_blk[3];
    if (tb) atomicAdd(&counters[0], tb);
    if (tk) atomicAdd(&counters[1], tk);
  }
}

}  // namespace samd

using namespace samd;

extern "C" int samd_count_errors_f32(const float* b, const float* b_hat, int64_t num_blocks, int64_t block_len,
                                     int soft, int64_t* counters, void* stream) {
  SAMD_REQUIRE(b && b_hat && counters, "null argument");
  SAMD_REQUIRE(num_blocks >= 0 && block_len >= 0, "negative size");
  if (num_blocks == 0 || block_len == 0) return SAMD_OK;
  const int grid = (int)std::min<int64_t>((num_blocks + 3) / 4, 256 * 16);
  hipLaunchKernelGGL(count_errors_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, b, b_hat, num_blocks,
                     block_len, soft, reinterpret_cast<unsigned long long*>(counters));
  return launch_status();
}
