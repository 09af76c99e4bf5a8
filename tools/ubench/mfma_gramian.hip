// Micro-benchmark for the north star's "MFMA used only for the batched small-matrix MIMO Gramian":
// A = H^H H (K x K complex, H is M x K complex per resource element) computed
//   (a) like csrc/mimo.hip does: one lane owns one resource element, lower triangle in registers (VALU), and
//   (b) with v_mfma_f32_4x4x1_16B_f32: the real form Hr = [[Re H, -Im H], [Im H, Re H]] (2M x 2K), G = Hr^T Hr as
//       2M rank-1 updates of 4x4 tiles, 16 resource elements (blocks) per instruction; the operands are loaded from
//       global memory DIRECTLY in the MFMA lane layout (lane 4b+t = element t of the row of block b) - the best case
//       for the matrix pipe: no register transposes in, none out.
// Both read the same [N, M, K] complex64 array and write the lower triangle; prints ns per 1000 REs, GB/s, VGPRs.
// hipcc --offload-arch=gfx950 -O3 -o mfma_gramian mfma_gramian.hip && ./mfma_gramian
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M, int K>
__global__ __launch_bounds__(256) void gram_valu(const float2* __restrict__ h, float2* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float2 hh[M][K];
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int k = 0; k < K; ++k) hh[m][k] = h[(i * M + m) * K + k];
  constexpr int T = K * (K + 1) / 2;
  float2 a[T];
  int t = 0;
#pragma unroll
  for (int r = 0; r < K; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      float re = 0.f, im = 0.f;
#pragma unroll
      for (int m = 0; m < M; ++m) {     // conj(h[m][r]) * h[m][c]
        re += hh[m][r].x * hh[m][c].x + hh[m][r].y * hh[m][c].y;
        im += hh[m][r].x * hh[m][c].y - hh[m][r].y * hh[m][c].x;
      }
      a[t++] = make_float2(re, im);
    }
#pragma unroll
  for (int q = 0; q < T; ++q) out[i * T + q] = a[q];
}

// one wave = 16 resource elements per MFMA group; 4 groups per wave-iteration to match the VALU kernel's 64 REs
template <int M, int K>
__global__ __launch_bounds__(256) void gram_mfma(const float* __restrict__ h, float* __restrict__ out, long n) {
  constexpr int C = 2 * K, NT = C / 4;              // real columns, 4-wide tiles (K = 2 -> 1 tile, K = 4 -> 2 tiles)
  const int lane = threadIdx.x & 63, b = lane >> 2, t = lane & 3;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  for (int g = 0; g < 4; ++g) {
    const long re = (wave * 4 + g) * 16 + b;
    if (re >= n) return;                             // n is a multiple of 64 in this benchmark
    f32x4 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* hp = h + re * (M * K * 2);
#pragma unroll
    for (int m = 0; m < M; ++m) {
      // row m of Hr: [Re h[m][0..K), -Im h[m][0..K)],  row M+m: [Im h[m][..], Re h[m][..]]
      float r1[NT], r2[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int col = 4 * i + t;                   // real column handled by this lane in tile i
        const int k = col % K, part = col / K;       // part 0: first K columns, 1: last K
        const float re_v = hp[(m * K + k) * 2], im_v = hp[(m * K + k) * 2 + 1];
        r1[i] = part == 0 ? re_v : -im_v;
        r2[i] = part == 0 ? im_v : re_v;
      }
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_4x4x1f32(r1[i], r1[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_4x4x1f32(r2[i], r2[j], acc[i][j], 0, 0, 0);
        }
    }
    // lane (b, t) holds column t of every tile: 4 values per tile; write the lower tiles (the consumer would need a
    // 4-lane gather per resource element to continue with a per-lane Cholesky - not charged here)
    float* o = out + re * (C * C);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) o[(4 * i + q) * C + 4 * j + t] = acc[i][j][q];
  }
}

template <int M, int K>
static void run(long n) {
  const size_t hb = (size_t)n * M * K * 8, ob = (size_t)n * 4 * K * K * 4;
  float *h, *o1, *o2;
  hipMalloc(&h, hb); hipMalloc(&o1, ob); hipMalloc(&o2, ob);
  std::vector<float> hh((size_t)n * M * K * 2);
  for (auto& v : hh) v = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(h, hh.data(), hb, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20;
  };
  const float tv = time([&] { hipLaunchKernelGGL((gram_valu<M, K>), dim3((n + 255) / 256), dim3(256), 0, 0, (const float2*)h, (float2*)o1, n); });
  const float tm = time([&] { hipLaunchKernelGGL((gram_mfma<M, K>), dim3((n + 255) / 256), dim3(256), 0, 0, h, o2, n); });
  // check one value: |h[0][.][0]|^2 summed
  std::vector<float> a(K * (K + 1)), g(4 * K * K);
  hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(g.data(), o2, g.size() * 4, hipMemcpyDeviceToHost);
  hipFuncAttributes fa, fm;
  hipFuncGetAttributes(&fa, (const void*)gram_valu<M, K>);
  hipFuncGetAttributes(&fm, (const void*)gram_mfma<M, K>);
  const int T = K * (K + 1) / 2;
  printf("M=%d K=%d N=%ld  VALU per-lane: %.3f ms (%.1f GB/s in+out, %d VGPR)   MFMA 4x4x1: %.3f ms (%d VGPR)   a00 %.6f vs G00 %.6f\n", M, K, n, tv,
         ((double)hb + (double)n * T * 8) / tv / 1e6, fa.numRegs, tm, fm.numRegs, a[0], g[0]);
  hipFree(h); hipFree(o1); hipFree(o2);
}

int main() {
  run<4, 2>(8192L * 768);      // config C4: 6.29 M data resource elements
  run<8, 4>(8192L * 768 / 4);
  return 0;
}
