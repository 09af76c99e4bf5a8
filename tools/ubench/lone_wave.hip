// Micro-benchmark (gfx950): what ONE wave of a workgroup pays when the other waves wait at a barrier - the regime of the
// layered LDPC engine's steps.  Cycles (s_memtime) per: dependent VALU op, independent VALU op, SALU op, taken branch,
// scalar load that hits the scalar cache (dependent chain), LDS read round trip, s_barrier with 16 waves arriving together.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lone_wave.hip -o tools/ubench/lone_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(1024) void k(const int* __restrict__ chain, unsigned long long* out, int n) {
  __shared__ float lds[4096];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  lds[threadIdx.x] = threadIdx.x;
  lds[threadIdx.x + 1024] = 1.f;
  __syncthreads();
  unsigned long long t[10] = {0};
  if (w == 0) {
    float a = lane, b = 1.0001f;
    t[0] = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int q = 0; q < 32; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));      // dependent chain
    }
    t[1] = __builtin_readcyclecounter();
    float r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3;
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r0) : "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r1) : "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r2) : "v"(b));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r3) : "v"(b));
      }
    }
    t[2] = __builtin_readcyclecounter();
    int s = n;
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int q = 0; q < 32; ++q) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");                       // dependent SALU
    }
    t[3] = __builtin_readcyclecounter();
    // taken branches: 32 per iteration (each jumps over one instruction)
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int q = 0; q < 32; ++q) asm volatile("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 1\n1:" : "+s"(s) : : "scc");
    }
    t[4] = __builtin_readcyclecounter();
    // dependent scalar loads (pointer chase in a 1 KB table: scalar-cache hits)
    int idx = 0;
    for (int i = 0; i < n * 8; ++i) {
      idx = __builtin_amdgcn_readfirstlane(chain[idx]);
      asm volatile("" : "+s"(idx));
    }
    t[5] = __builtin_readcyclecounter();
    // LDS round trips (address depends on the value read)
    int ai = lane;
    for (int i = 0; i < n * 8; ++i) {
      const float v = lds[1024 + ai];
      ai = (lane + (int)v) & 1023;
    }
    t[6] = __builtin_readcyclecounter();
    out[16] = (unsigned long long)(a + r0 + r1 + r2 + r3) + s + idx + ai;
  }
  __syncthreads();
  const unsigned long long b0 = __builtin_readcyclecounter();
  for (int i = 0; i < n * 8; ++i) asm volatile("s_barrier" ::: "memory");
  const unsigned long long b1 = __builtin_readcyclecounter();
  if (w == 0 && lane == 0) {
    for (int q = 0; q < 7; ++q) out[q] = t[q];
    out[8] = b0;
    out[9] = b1;
  }
}

int main() {
  const int n = 256;
  std::vector<int> chain(256);
  for (int i = 0; i < 256; ++i) chain[i] = (i * 67 + 13) & 255;
  int* dchain;
  unsigned long long* dout;
  (void)hipMalloc(&dchain, 1024);
  (void)hipMalloc(&dout, 32 * 8);
  (void)hipMemcpy(dchain, chain.data(), 1024, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, dchain, dout, n);
  (void)hipDeviceSynchronize();
  unsigned long long o[32];
  (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  printf("s_memtime cycles per operation, one wave running, 15 waiting at a barrier (gfx950)\n");
  printf("dependent VALU (v_fma)       %.2f\n", (double)(o[1] - o[0]) / (n * 32.0));
  printf("independent VALU x4 (v_fma)  %.2f\n", (double)(o[2] - o[1]) / (n * 32.0));
  printf("dependent SALU (s_add)       %.2f\n", (double)(o[3] - o[2]) / (n * 32.0));
  printf("s_cmp + taken s_cbranch      %.2f\n", (double)(o[4] - o[3]) / (n * 32.0));
  printf("dependent s_load (hit)       %.2f\n", (double)(o[5] - o[4]) / (n * 8.0));
  printf("dependent LDS read           %.2f\n", (double)(o[6] - o[5]) / (n * 8.0));
  printf("s_barrier, 16 waves          %.2f\n", (double)(o[9] - o[8]) / (n * 8.0));
  return 0;
}
