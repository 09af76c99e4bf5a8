/*
 * oracle/ldpc_bp.c - plain-C restatement of the reference's flooding BP decoder.
 *
 * TEST INFRASTRUCTURE (see oracle/__init__.py): used by tests/ as a fast checker at sizes
 * the NumPy oracle is slow at, and by bench.py as the "port" CPU baseline.  Never linked
 * into or called by the product library.
 *
 * Follows /root/reference/src/sionna/phy/fec/ldpc/decoding.py:
 *   main loop                    :544-637, 416-524   (clip, logits -> LLR, num_iter fixed)
 *   vn_update_sum                :681-732
 *   cn_update_offset_minsum      :755-909  (1e5 sentinel, "double_min" sum test)
 *   cn_update_tanh               :955-1043
 *   cn_update_phi                :1045-1166
 * Arithmetic and order are identical to oracle/ldpc_bp.py (sequential over a node's edges
 * in edge order; compile with -ffp-contract=off): min-sum results are bit-identical to the
 * NumPy oracle, boxplus variants agree to libm rounding.  One codeword at a time,
 * OpenMP over codewords.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LARGE_VAL 100000.0f

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float sign_nz(float x) { return x < 0.f ? -1.f : 1.f; }
static inline float sgn3(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
static inline float phi(float x) {
  x = clampf(x, 8.5e-8f, 16.635532f);
  const float e = expf(x);
  return logf(e + 1.f) - logf(e - 1.f);
}

typedef struct {
  int E, N_cn, N_vn;
  int *cn_ptr, *cn_edge, *vn_ptr;
} graph_t;

static void cn_update(float* v, float* sg, int d, int mode, float llr_max, float offset) {
  int i;
  if (mode == 2 || mode == 3) {
    float node_sign = 1.f, min1 = INFINITY, min2 = INFINITY, node_sum = 0.f;
    if (mode == 2) offset = 0.f;
    for (i = 0; i < d; ++i) {
      const float x = clampf(v[i], -LARGE_VAL, LARGE_VAL);
      sg[i] = sign_nz(x);
      node_sign *= sg[i];
      v[i] = fabsf(x);
      min1 = fminf(min1, v[i]);
    }
    for (i = 0; i < d; ++i) {
      const float t = v[i] - min1;
      v[i] = (t == 0.f) ? LARGE_VAL : t;
      min2 = fminf(min2, v[i]);
      node_sum += v[i];
    }
    min2 = min2 + min1;
    node_sum = node_sum - (2.f * LARGE_VAL - 1.f);
    {
      const float dm = 0.5f * (1.f - sgn3(node_sum));
      const float min_e = (1.f - dm) * min1 + dm * min2;
      for (i = 0; i < d; ++i) {
        float m = (v[i] == LARGE_VAL) ? min_e : min1;
        m = fmaxf(m - offset, 0.f);
        v[i] = clampf((sg[i] * node_sign) * m, -llr_max, llr_max);
      }
    }
  } else if (mode == 1) {
    float node_sign = 1.f, sum = 0.f;
    for (i = 0; i < d; ++i) {
      sg[i] = sign_nz(v[i]);
      node_sign *= sg[i];
      v[i] = phi(fabsf(v[i]));
      sum += v[i];
    }
    for (i = 0; i < d; ++i) {
      const float e = -1.f * v[i] + sum;
      v[i] = clampf((sg[i] * node_sign) * phi(e), -llr_max, llr_max);
    }
  } else {
    float prod = 1.f;
    const float ac = 1.f - 1e-7f;
    for (i = 0; i < d; ++i) {
      float t = tanhf(v[i] / 2.f);
      t = (t == 0.f) ? 1e-12f : t;
      v[i] = t;
      prod *= t;
    }
    for (i = 0; i < d; ++i) {
      float e = (1.f / v[i]) * prod;
      e = (fabsf(e) < 1e-7f) ? 0.f : e;
      e = clampf(e, -ac, ac);
      v[i] = clampf(2.f * atanhf(e), -llr_max, llr_max);
    }
  }
}

/* cn_idx/vn_idx: edge list in VN-major order, ascending CN inside a VN.
 * llr: [B,N_vn] logits; out: [B,N_vn] (hard bits or logits).  Returns 0 on success. */
int oracle_ldpc_bp_decode(int E, int N_cn, int N_vn, const int* cn_idx, const int* vn_idx, const float* llr,
                          float* out, int B, int num_iter, int cn_mode, float llr_max, float offset, int hard_out,
                          int nthreads) {
  graph_t g;
  int e, c, v, max_dc = 0;
  int* fill;
  g.E = E; g.N_cn = N_cn; g.N_vn = N_vn;
  g.cn_ptr = (int*)calloc((size_t)N_cn + 1, sizeof(int));
  g.vn_ptr = (int*)calloc((size_t)N_vn + 1, sizeof(int));
  g.cn_edge = (int*)malloc((size_t)E * sizeof(int));
  fill = (int*)malloc((size_t)N_cn * sizeof(int));
  if (!g.cn_ptr || !g.vn_ptr || !g.cn_edge || !fill) return -1;
  for (e = 0; e < E; ++e) { g.cn_ptr[cn_idx[e] + 1]++; g.vn_ptr[vn_idx[e] + 1]++; }
  for (c = 0; c < N_cn; ++c) { if (g.cn_ptr[c + 1] > max_dc) max_dc = g.cn_ptr[c + 1]; g.cn_ptr[c + 1] += g.cn_ptr[c]; }
  for (v = 0; v < N_vn; ++v) g.vn_ptr[v + 1] += g.vn_ptr[v];
  memcpy(fill, g.cn_ptr, (size_t)N_cn * sizeof(int));
  for (e = 0; e < E; ++e) g.cn_edge[fill[cn_idx[e]]++] = e;   /* stable: ascending VN inside a CN */
  free(fill);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    float* msg = (float*)malloc((size_t)E * sizeof(float));
    float* l = (float*)malloc((size_t)N_vn * sizeof(float));
    float* xh = (float*)malloc((size_t)N_vn * sizeof(float));
    float* tmp = (float*)malloc((size_t)(max_dc > 0 ? max_dc : 1) * sizeof(float));
    float* sg = (float*)malloc((size_t)(max_dc > 0 ? max_dc : 1) * sizeof(float));
    int b;
#pragma omp for schedule(static)
    for (b = 0; b < B; ++b) {
      int it, i, cc, vv;
      for (vv = 0; vv < N_vn; ++vv) {
        l[vv] = -1.f * clampf(llr[(size_t)b * N_vn + vv], -llr_max, llr_max);
        xh[vv] = l[vv];
      }
      for (i = 0; i < E; ++i) msg[i] = l[vn_idx[i]];
      for (it = 0; it < num_iter; ++it) {
        for (cc = 0; cc < N_cn; ++cc) {
          const int e0 = g.cn_ptr[cc], d = g.cn_ptr[cc + 1] - e0;
          for (i = 0; i < d; ++i) tmp[i] = msg[g.cn_edge[e0 + i]];
          cn_update(tmp, sg, d, cn_mode, llr_max, offset);
          for (i = 0; i < d; ++i) msg[g.cn_edge[e0 + i]] = tmp[i];
        }
        for (vv = 0; vv < N_vn; ++vv) {
          const int e0 = g.vn_ptr[vv], e1 = g.vn_ptr[vv + 1];
          float x = 0.f;
          for (i = e0; i < e1; ++i) x += msg[i];
          x += l[vv];
          for (i = e0; i < e1; ++i) msg[i] = clampf(-1.f * msg[i] + x, -llr_max, llr_max);
          xh[vv] = clampf(x, -llr_max, llr_max);
        }
      }
      for (vv = 0; vv < N_vn; ++vv)
        out[(size_t)b * N_vn + vv] = hard_out ? ((0.f >= xh[vv]) ? 1.f : 0.f) : -1.f * xh[vv];
    }
    free(msg); free(l); free(xh); free(tmp); free(sg);
  }
  free(g.cn_ptr); free(g.vn_ptr); free(g.cn_edge);
  return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
