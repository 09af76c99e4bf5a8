// Shared host/device helpers for libsionna_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

#include "../../include/sionna_amd.h"

namespace samd {

void set_error(const std::string& msg);

#define SAMD_HIP_CHECK(expr)                                                        \
  do {                                                                              \
    hipError_t _e = (expr);                                                         \
    if (_e != hipSuccess) {                                                         \
      samd::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));           \
      return SAMD_ERR_HIP;                                                          \
    }                                                                               \
  } while (0)

#define SAMD_REQUIRE(cond, msg)                                                     \
  do {                                                                              \
    if (!(cond)) {                                                                  \
      samd::set_error(std::string(msg) + " (" #cond ")");                           \
      return SAMD_ERR_INVALID;                                                      \
    }                                                                               \
  } while (0)

inline int launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(std::string("kernel launch: ") + hipGetErrorString(e));
    return SAMD_ERR_HIP;
  }
  return SAMD_OK;
}

// SAMD_HOST_ONLY (development option, read when a handle is built): tables and schedules are built, nothing is copied
// to a device - for the tools and tests that inspect what the host generates (the specialised LDPC kernels' source) on
// a machine without a GPU.  Such a handle refuses every launch.
bool host_only();

template <typename T>
inline int upload(T** dst, const T* src, size_t n) {
  *dst = nullptr;
  if (n == 0 || host_only()) return SAMD_OK;
  SAMD_HIP_CHECK(hipMalloc((void**)dst, n * sizeof(T)));
  SAMD_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return SAMD_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set it the first time a (kernel,
// device, size) triple is launched from this process instead of on every launch.  Lock-free: a hash collision or a race
// between two host threads only repeats the (idempotent) runtime call.
inline int set_max_dynamic_lds(const void* fn, int bytes) {
  static std::atomic<uint64_t> seen[512];
  int dev = 0;
  SAMD_HIP_CHECK(hipGetDevice(&dev));
  const uint64_t key = ((uint64_t)(uintptr_t)fn << 12) ^ ((uint64_t)bytes << 8) ^ (uint64_t)(dev + 1);
  std::atomic<uint64_t>& slot = seen[(key * 0x9E3779B97F4A7C15ull) >> 55];
  if (slot.load(std::memory_order_acquire) == key) return SAMD_OK;
  SAMD_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  slot.store(key, std::memory_order_release);
  return SAMD_OK;
}
#define SAMD_SET_MAX_LDS(fn, bytes)                                                  \
  do {                                                                              \
    const int _rc = samd::set_max_dynamic_lds((const void*)(fn), (bytes));          \
    if (_rc != SAMD_OK) return _rc;                                                 \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr int kWave = 64;  // CDNA wavefront

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float clampf(float x, float lo, float hi) {
  // tf.clip_by_value = min(max(x, lo), hi)
  return fminf(fmaxf(x, lo), hi);
}

// Philox4x32-10, identical to oracle/utils.py (Random123 constants).
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}

__device__ __forceinline__ uint4 philox_block(uint64_t seed, uint64_t call, uint64_t block) {
  return philox4x32_10(make_uint4((uint32_t)block, (uint32_t)(block >> 32), (uint32_t)call,
                                  (uint32_t)(call >> 32)),
                       make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}

__device__ __forceinline__ float u01(uint32_t x) {
  return (float)(x >> 8) * 5.9604644775390625e-08f + 2.98023223876953125e-08f;  // 2^-24, 2^-25
}

}  // namespace samd
