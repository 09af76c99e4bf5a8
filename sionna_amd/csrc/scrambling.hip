// Scrambler / TB5GScrambler / Descrambler and the 38.211 pseudo-random sequence.
//   Scrambler.call        /root/reference/src/sionna/phy/fec/scrambling.py:186-261
//   TB5GScrambler.call    /root/reference/src/sionna/phy/fec/scrambling.py:412-468
//   generate_prng_seq     /root/reference/src/sionna/phy/nr/utils.py:14-78 (TS 38.211 5.2.1)
// One streaming pass (4 B in + 4 B out per element); the sequence is periodic over the
// leading dimensions (one row per stream / keep_batch_constant) and stays L2 resident.
#include "common.h"

#include <vector>

namespace samd {
namespace {

template <typename R>
__global__ void scramble_kernel(const R* __restrict__ x, const float* __restrict__ seq, long long total,
                                long long period, int binary, R* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const R s = (R)seq[i % period];
  // binary: |x - s| (scrambling.py:252-254); soft values: x * (-2 s + 1) (:255-257)
  const R d = x[i] - s;
  out[i] = binary ? (d < (R)0 ? -d : d) : x[i] * ((R)-2 * s + (R)1);
}

}  // namespace
}  // namespace samd

using namespace samd;

extern "C" int samd_scramble_f32(const float* x, const float* seq, int64_t total, int64_t period, int binary,
                                 float* out, void* stream) {
  SAMD_REQUIRE(x && seq && out, "null argument");
  SAMD_REQUIRE(total >= 0 && period > 0, "bad size");
  if (total == 0) return SAMD_OK;
  SAMD_REQUIRE((total + 255) / 256 < (1ll << 31), "grid too large");
  scramble_kernel<float><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, seq, total, period, binary, out);
  return launch_status();
}

// precision = "double" (reference block.py:25-52): float64 values, the same float32 bit sequence
extern "C" int samd_scramble_f64(const double* x, const float* seq, int64_t total, int64_t period, int binary,
                                 double* out, void* stream) {
  SAMD_REQUIRE(x && seq && out, "null argument");
  SAMD_REQUIRE(total >= 0 && period > 0, "bad size");
  if (total == 0) return SAMD_OK;
  SAMD_REQUIRE((total + 255) / 256 < (1ll << 31), "grid too large");
  scramble_kernel<double><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, seq, total, period, binary, out);
  return launch_status();
}

extern "C" int samd_nr_prng_seq_f32(uint32_t c_init, int64_t length, float* out, void* stream) {
  SAMD_REQUIRE(out && length > 0, "bad argument");
  // Length-31 Gold sequence, N_c = 1600: x1(n+31) = x1(n+3) + x1(n), x2(n+31) = x2(n+3) +
  // x2(n+2) + x2(n+1) + x2(n) mod 2, x1 = 1 0 0 ..., x2 = bits of c_init LSB first;
  // c(n) = x1(n+N_c) + x2(n+N_c).  Init-time (the host class caches the sequence per (c_init, length)): generated
  // on the host, uploaded asynchronously.
  const int64_t nc = 1600, total = length + nc + 31;
  std::vector<uint8_t> x1(total, 0), x2(total, 0);
  x1[0] = 1;
  for (int i = 0; i < 31; ++i) x2[i] = (c_init >> i) & 1u;
  for (int64_t i = 0; i < length + nc; ++i) {
    x1[i + 31] = x1[i + 3] ^ x1[i];
    x2[i + 31] = x2[i + 3] ^ x2[i + 2] ^ x2[i + 1] ^ x2[i];
  }
  // stream-ordered upload without a host synchronisation: the staging vector is released by a host callback that
  // the stream runs after the copy
  auto* c = new std::vector<float>(length);
  for (int64_t i = 0; i < length; ++i) (*c)[i] = (float)(x1[i + nc] ^ x2[i + nc]);
  hipError_t e = hipMemcpyAsync(out, c->data(), length * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream);
  if (e == hipSuccess)
    e = hipLaunchHostFunc((hipStream_t)stream, [](void* p) { delete static_cast<std::vector<float>*>(p); }, c);
  if (e != hipSuccess) {
    (void)hipStreamSynchronize((hipStream_t)stream);
    delete c;
    set_error(hipGetErrorString(e));
    return SAMD_ERR_HIP;
  }
  return SAMD_OK;
}
