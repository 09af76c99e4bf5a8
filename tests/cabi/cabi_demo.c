/* Plain-C client of the C-ABI (no Python, no torch): LDPC 5G encode -> BPSK-like LLRs with a few
 * flipped signs -> on-chip min-sum decode -> compare, through include/sionna_amd.h only.
 * Built and run by tests/test_gpu_cabi_c.py with hipcc on the GPU box; prints "CABI_DEMO_OK". */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "sionna_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d at %s\n", (int)e_, #x); return 2; } } while (0)
#define CHECK_SAMD(x) do { int r_ = (x); if (r_ != 0) { printf("samd error %d (%s) at %s\n", r_, samd_last_error(), #x); return 3; } } while (0)

int main(int argc, char** argv) {
  /* base-graph entries (row, col, shift) of the lifted code are supplied by the caller: argv[1] is a
   * text file "bg z k n num_entries" followed by num_entries triples, written by the Python test
   * from the package's BG tables (the C-ABI takes them as plain int16 arrays). */
  if (argc < 2) { printf("usage: cabi_demo <code.txt>\n"); return 1; }
  FILE* f = fopen(argv[1], "r");
  if (!f) { printf("cannot open %s\n", argv[1]); return 1; }
  int bg, z, k, n, ne;
  if (fscanf(f, "%d %d %d %d %d", &bg, &z, &k, &n, &ne) != 5) return 1;
  int16_t* rows = (int16_t*)malloc(sizeof(int16_t) * ne);
  int16_t* cols = (int16_t*)malloc(sizeof(int16_t) * ne);
  int16_t* shifts = (int16_t*)malloc(sizeof(int16_t) * ne);
  for (int i = 0; i < ne; ++i) {
    int r, c, s;
    if (fscanf(f, "%d %d %d", &r, &c, &s) != 3) return 1;
    rows[i] = (int16_t)r; cols[i] = (int16_t)c; shifts[i] = (int16_t)s;
  }
  fclose(f);
  if (samd_device_count() < 1) { printf("no device\n"); return 4; }

  samd_ldpc5g_t* code = NULL;
  CHECK_SAMD(samd_ldpc5g_create(bg, z, rows, cols, shifts, ne, k, n, 0 /*no interleaver*/, 0 /*pruned*/, &code));

  const int batch = 257;
  float* h_u = (float*)malloc(sizeof(float) * batch * k);
  float* h_c = (float*)malloc(sizeof(float) * batch * n);
  float* h_out = (float*)malloc(sizeof(float) * batch * k);
  uint32_t lcg = 12345u;
  for (int i = 0; i < batch * k; ++i) { lcg = lcg * 1664525u + 1013904223u; h_u[i] = (float)((lcg >> 16) & 1u); }

  float *d_u, *d_c, *d_llr, *d_out;
  CHECK_HIP(hipMalloc((void**)&d_u, sizeof(float) * batch * k));
  CHECK_HIP(hipMalloc((void**)&d_c, sizeof(float) * batch * n));
  CHECK_HIP(hipMalloc((void**)&d_llr, sizeof(float) * batch * n));
  CHECK_HIP(hipMalloc((void**)&d_out, sizeof(float) * batch * k));
  CHECK_HIP(hipMemcpy(d_u, h_u, sizeof(float) * batch * k, hipMemcpyHostToDevice));
  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));

  CHECK_SAMD(samd_ldpc5g_encode_f32(code, d_u, d_c, batch, st));
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(h_c, d_c, sizeof(float) * batch * n, hipMemcpyDeviceToHost));
  /* logits: +4 for a one, -4 for a zero, every 37th position received wrongly with low confidence */
  for (int i = 0; i < batch * n; ++i) {
    float l = h_c[i] > 0.5f ? 4.0f : -4.0f;
    if (i % 37 == 0) l = -0.5f * l;
    h_c[i] = l;
  }
  CHECK_HIP(hipMemcpy(d_llr, h_c, sizeof(float) * batch * n, hipMemcpyHostToDevice));
  int rc = samd_ldpc5g_decode_f32(code, d_llr, d_out, batch, 20, SAMD_CN_MINSUM, 20.0f, 0.5f, 1 /*hard*/, 1 /*infobits*/,
                                  NULL, 0, st);
  if (rc != 0) { printf("decode failed: %d (%s)\n", rc, samd_last_error()); return 5; }
  CHECK_HIP(hipStreamSynchronize(st));
  CHECK_HIP(hipMemcpy(h_out, d_out, sizeof(float) * batch * k, hipMemcpyDeviceToHost));
  long errors = 0;
  for (int i = 0; i < batch * k; ++i) errors += (h_out[i] != h_u[i]);
  printf("version %d, %d codewords, bit errors %ld\n", samd_version(), batch, errors);
  samd_ldpc5g_destroy(code);
  if (errors != 0) return 6;
  printf("CABI_DEMO_OK\n");
  return 0;
}
