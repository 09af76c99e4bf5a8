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
 * NumPy oracle; boxplus-phi too since round 3 (both evaluate phi with the defined exp / log below), the tanh rule
 * agrees to libm rounding.  One codeword at a time, OpenMP over codewords.
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
/* ---- float32 exp / log of the boxplus-phi rule: a DEFINED arithmetic (round 3) ---------------------------------
 * decoding.py:1120 evaluates log(exp(x)+1) - log(exp(x)-1) with TensorFlow-CPU's float32 exp / log, i.e. Eigen's
 * pexp<Packet8f> / plog<Packet8f> (SURVEY 8(c) "third-party arithmetic").  Eigen is absent from /root/reference and
 * from this image (searched: find / -name GenericPacketMathFunctions.h, -type d -name Eigen; torch/include, /opt/rocm),
 * so its published algorithm is restated here: Moshier's Cephes single-precision expf / logf (range reduction
 * m = floor(x log2(e) + 1/2), r = x - m ln2 with ln2 = 0.693359375 - 2.12194440e-4, degree-5 polynomial, 2^m scaling;
 * frexp to [sqrt(1/2), sqrt(2)), degree-8 polynomial, e ln2 added last) in the evaluation order Eigen gives them
 * (fused multiply-adds, the log polynomial in three interleaved parts).  Coefficients are Cephes' (expf.c / logf.c);
 * whether the Eigen snapshot of a given TensorFlow release refits them cannot be checked offline - "parity unpinned"
 * for the last bits against TensorFlow, but the rule now has ONE bit-level definition that the NumPy oracle, this file
 * and every HIP engine (csrc/bp_math.h) follow, so boxplus-phi parity is array_equal instead of a tolerance.
 * Measured here over all 231,640,148 floats of the clipped domain [8.5e-8, 16.635532]: exp within 1.005 ulp, both logs
 * within 0.90 ulp of long double; phi(16.635532) = 0 and phi(8.5e-8) = 16.6355324 exactly (the reference's
 * "all-erasure -> zeros" test, test_ldpc_decoding.py:279-291) without a special case.
 * Compile with -mfma (fmaf inlined to one instruction) and -ffp-contract=off (nothing else fused). */
static inline float spec_expf(float x) {
  const float m = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
  float r = fmaf(m, -0.693359375f, x);
  float y = 1.9875691500E-4f, z;
  r = fmaf(m, 2.12194440e-4f, r);
  z = r * r;
  y = fmaf(y, r, 1.3981999507E-3f);
  y = fmaf(y, r, 8.3334519073E-3f);
  y = fmaf(y, r, 4.1665795894E-2f);
  y = fmaf(y, r, 1.6666665459E-1f);
  y = fmaf(y, r, 5.0000001201E-1f);
  y = fmaf(y, z, r);
  y = y + 1.0f;
  return ldexpf(y, (int)m);
}
static inline float spec_logf(float x) { /* normal positive x */
  int e;
  float f = frexpf(x, &e), ef = (float)e, x2, x3, y, y1, y2;
  const int lt = f < 0.707106781186547524f;
  const float tmp = lt ? f : 0.f;
  f = f - 1.0f;
  ef = ef - (lt ? 1.0f : 0.f);
  f = f + tmp;
  x2 = f * f;
  x3 = x2 * f;
  y = fmaf(7.0376836292E-2f, f, -1.1514610310E-1f);
  y1 = fmaf(-1.2420140846E-1f, f, 1.4249322787E-1f);
  y2 = fmaf(2.0000714765E-1f, f, -2.4999993993E-1f);
  y = fmaf(y, f, 1.1676998740E-1f);
  y1 = fmaf(y1, f, -1.6668057665E-1f);
  y2 = fmaf(y2, f, 3.3333331174E-1f);
  y = fmaf(y, x3, y1);
  y = fmaf(y, x3, y2);
  y = y * x3;
  y = fmaf(-0.5f, x2, y);
  f = f + y;
  return fmaf(ef, 0.69314718055994530942f, f);
}
/* ---- round 5: a CHEAPER defined exp / log for phi (the two above stay the definition of the Polar BP decoder's boxplus) ----
 * Round 4's verdict: the defined phi costs 1.75 x the hardware-transcendental form and buys agreement with this file, not
 * with TensorFlow - "find the instruction-minimal IEEE sequence that still passes the bars against the executed reference".
 * The literal form log(e^x + 1) - log(e^x - 1) of decoding.py:1120 is kept (its cancellation behaviour IS the rule); what
 * changes is how exp and log are evaluated, still as one fixed sequence of IEEE operations:
 *   exp: the Cephes reduction and polynomial as above, but m = rint(x log2 e) by the magic-number addition (no floor, no
 *        conversion: t = fma(x, log2 e, 1.5 2^23), m = t - 1.5 2^23) and the scaling 2^m as an integer addition to the
 *        exponent field (bits(y) + (bits(t) << 23); 0 <= m <= 24 on the clipped domain);
 *   log: table-driven (Tang): p = mantissa in [1, 2), j = its top six bits, (inv_j, lncm_j) = phi_tab[j] (tools/gen_phi_tables.py:
 *        1 / c_j rounded to float32 and -ln of THAT float32, minus ln 2), r = fma(p, inv_j, -1) (|r| <= 1/128, exact product),
 *        ln p = -ln inv_j + r (1 + r (-1/2 + r/3)) in three fma, ln x = fma(e + 1, ln 2, .) with e + 1 = frexp's exponent.
 *        10 operations instead of 23; absolute error <= 1.1e-7 (0.6 ulp of the results phi takes differences of).
 * phi(16.635532) = 0 - the reference's own "all-erasure -> zeros" test - is a selection now (e + 1 and e - 1 fall into different
 * table intervals there and their logarithms may round one ulp apart).
 * Measured over 6.3 M floats of the clipped domain (tools/phi_agreement.c): identical to the literal form on a correctly rounded
 * libm for 89.9 % of the arguments (Cephes form above: 91.1 %), within 1e-5 relative for 99.39 % (99.48 %). */
static const float phi_tab[64][2] = {
#include "phi_tab.inc"
};
static inline float phi_expf(float x) {
  union { float f; unsigned u; } y, t;
  float m, r, z;
  t.f = fmaf(x, 1.44269504088896341f, 12582912.0f);
  m = t.f - 12582912.0f;
  r = fmaf(m, -0.693359375f, x);
  y.f = 1.9875691500E-4f;
  r = fmaf(m, 2.12194440e-4f, r);
  z = r * r;
  y.f = fmaf(y.f, r, 1.3981999507E-3f);
  y.f = fmaf(y.f, r, 8.3334519073E-3f);
  y.f = fmaf(y.f, r, 4.1665795894E-2f);
  y.f = fmaf(y.f, r, 1.6666665459E-1f);
  y.f = fmaf(y.f, r, 5.0000001201E-1f);
  y.f = fmaf(y.f, z, r);
  y.f = y.f + 1.0f;
  y.u += t.u << 23;
  return y.f;
}
static inline float phi_logf(float x) { /* normal positive x */
  union { float f; unsigned u; } b, p;
  float ef, r, t;
  unsigned j;
  b.f = x;
  ef = (float)((int)((b.u >> 23) & 0xffu) - 126);                 /* frexp's exponent: e + 1 */
  j = (b.u >> 17) & 63u;
  p.u = (b.u & 0x007fffffu) | 0x3f800000u;
  r = fmaf(p.f, phi_tab[j][0], -1.0f);
  t = fmaf(r, 0.333333343f, -0.5f);
  t = fmaf(t, r, 1.0f);
  t = fmaf(t, r, phi_tab[j][1]);
  return fmaf(ef, 0.69314718055994530942f, t);
}
static inline float phi(float x) { /* decoding.py:1110-1120, literal form on the defined exp / log */
  float e, r;
  x = clampf(x, 8.5e-8f, 16.635532f);
  e = phi_expf(x);
  r = phi_logf(e + 1.f) - phi_logf(e - 1.f);
  return (x == 16.635532f) ? 0.f : r;
}
/* element-wise phi / exp / log for the NumPy oracle (oracle/ldpc_bp.py) and the pin tests */
void oracle_phi_f32(const float* x, float* y, long n) { long i; for (i = 0; i < n; ++i) y[i] = phi(x[i]); }
void oracle_spec_exp_f32(const float* x, float* y, long n) { long i; for (i = 0; i < n; ++i) y[i] = spec_expf(x[i]); }
void oracle_spec_log_f32(const float* x, float* y, long n) { long i; for (i = 0; i < n; ++i) y[i] = spec_logf(x[i]); }
void oracle_phi_exp_f32(const float* x, float* y, long n) { long i; for (i = 0; i < n; ++i) y[i] = phi_expf(x[i]); }
void oracle_phi_log_f32(const float* x, float* y, long n) { long i; for (i = 0; i < n; ++i) y[i] = phi_logf(x[i]); }

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

/* ------------------------------------------------------------------------------------------------------------------
 * The same decoder, W codewords at a time (bench.py's CPU baseline since round 3; the scalar function above stays the
 * checker).  Round 2's baseline was the scalar loop - 8 codewords/s per thread - which made the GPU / CPU ratio soft.
 * Here the messages of W = 8 codewords sit side by side (msg[e][W]) and every node update is a loop over the W lanes
 * that gcc vectorises (-O3 -mavx2 -mfma); each lane executes the scalar function's operations in the scalar function's
 * order, so the outputs are bit-identical (tests/test_oracle_pins.py).  exp / log of the boxplus-phi rule are phi_expf /
 * phi_logf above (integer forms throughout; the table look-up is a gather).  min-sum, offset-min-sum and boxplus-phi; the
 * tanh rule stays scalar. */
#define SW 8
/* min / max / clamp as comparisons + selects (vminps / vmaxps / blends): fminf / fmaxf are library calls under
 * -fno-fast-math and keep the lane loops scalar; for the values of this decoder (no NaN; magnitudes are never -0) both
 * forms return the same bits */
static inline float vminf(float a, float b) { return a < b ? a : b; }
static inline float vmaxf(float a, float b) { return a > b ? a : b; }
static inline float vclampf(float x, float lo, float hi) { return vminf(vmaxf(x, lo), hi); }
static inline float phi_v(float x) {                                          /* phi() above, lane form: the same operations */
  float e, r;
  x = vclampf(x, 8.5e-8f, 16.635532f);
  e = phi_expf(x);
  r = phi_logf(e + 1.f) - phi_logf(e - 1.f);
  return (x == 16.635532f) ? 0.f : r;
}

/* node update of one check node for SW codewords: v / sg [d][SW] */
static void cn_update_w(float* v, float* sg, int d, int mode, float llr_max, float offset) {
  int i, j;
  if (mode == 2 || mode == 3) {
    float node_sign[SW], min1[SW], min2[SW], node_sum[SW];
    if (mode == 2) offset = 0.f;
    for (j = 0; j < SW; ++j) { node_sign[j] = 1.f; min1[j] = INFINITY; min2[j] = INFINITY; node_sum[j] = 0.f; }
    for (i = 0; i < d; ++i)
#pragma omp simd
      for (j = 0; j < SW; ++j) {
        const float x = vclampf(v[i * SW + j], -LARGE_VAL, LARGE_VAL);
        const float s = x < 0.f ? -1.f : 1.f;
        sg[i * SW + j] = s;
        node_sign[j] *= s;
        v[i * SW + j] = fabsf(x);
        min1[j] = vminf(min1[j], fabsf(x));
      }
    for (i = 0; i < d; ++i)
#pragma omp simd
      for (j = 0; j < SW; ++j) {
        const float t = v[i * SW + j] - min1[j];
        const float r = (t == 0.f) ? LARGE_VAL : t;
        v[i * SW + j] = r;
        min2[j] = vminf(min2[j], r);
        node_sum[j] += r;
      }
    for (i = 0; i < d; ++i)
#pragma omp simd
      for (j = 0; j < SW; ++j) {
        const float ns = node_sum[j] - (2.f * LARGE_VAL - 1.f);
        const float sg3 = ns > 0.f ? 1.f : (ns < 0.f ? -1.f : 0.f);
        const float dm = 0.5f * (1.f - sg3);
        const float min_e = (1.f - dm) * min1[j] + dm * (min2[j] + min1[j]);
        float m = (v[i * SW + j] == LARGE_VAL) ? min_e : min1[j];
        m = vmaxf(m - offset, 0.f);
        v[i * SW + j] = vclampf((sg[i * SW + j] * node_sign[j]) * m, -llr_max, llr_max);
      }
  } else {
    float node_sign[SW], sum[SW];
    for (j = 0; j < SW; ++j) { node_sign[j] = 1.f; sum[j] = 0.f; }
    for (i = 0; i < d; ++i)
#pragma omp simd
      for (j = 0; j < SW; ++j) {
        const float x = v[i * SW + j];
        const float s = x < 0.f ? -1.f : 1.f;
        const float ph = phi_v(fabsf(x));
        sg[i * SW + j] = s;
        node_sign[j] *= s;
        v[i * SW + j] = ph;
        sum[j] += ph;
      }
    for (i = 0; i < d; ++i)
#pragma omp simd
      for (j = 0; j < SW; ++j) {
        const float e = -1.f * v[i * SW + j] + sum[j];
        v[i * SW + j] = vclampf((sg[i * SW + j] * node_sign[j]) * phi_v(e), -llr_max, llr_max);
      }
  }
}

int oracle_ldpc_bp_decode_simd(int E, int N_cn, int N_vn, const int* cn_idx, const int* vn_idx, const float* llr,
                               float* out, int B, int num_iter, int cn_mode, float llr_max, float offset, int hard_out,
                               int nthreads) {
  graph_t g;
  int e, c, v, max_dc = 0, groups = (B + SW - 1) / SW;
  int* fill;
  if (cn_mode < 1 || cn_mode > 3) return -2;
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
  for (e = 0; e < E; ++e) g.cn_edge[fill[cn_idx[e]]++] = e;
  free(fill);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    float* msg = (float*)aligned_alloc(64, (size_t)E * SW * sizeof(float));
    float* l = (float*)aligned_alloc(64, (size_t)N_vn * SW * sizeof(float));
    float* xh = (float*)aligned_alloc(64, (size_t)N_vn * SW * sizeof(float));
    float* tmp = (float*)aligned_alloc(64, (size_t)(max_dc > 0 ? max_dc : 1) * SW * sizeof(float));
    float* sg = (float*)aligned_alloc(64, (size_t)(max_dc > 0 ? max_dc : 1) * SW * sizeof(float));
    int gi;
#pragma omp for schedule(dynamic, 1)
    for (gi = 0; gi < groups; ++gi) {
      int it, i, j, cc, vv;
      for (vv = 0; vv < N_vn; ++vv)
        for (j = 0; j < SW; ++j) {
          const int b = gi * SW + j;
          const float in = b < B ? llr[(size_t)b * N_vn + vv] : 0.f;
          l[vv * SW + j] = -1.f * vclampf(in, -llr_max, llr_max);
          xh[vv * SW + j] = l[vv * SW + j];
        }
      for (i = 0; i < E; ++i)
        for (j = 0; j < SW; ++j) msg[i * SW + j] = l[vn_idx[i] * SW + j];
      for (it = 0; it < num_iter; ++it) {
        for (cc = 0; cc < N_cn; ++cc) {
          const int e0 = g.cn_ptr[cc], d = g.cn_ptr[cc + 1] - e0;
          for (i = 0; i < d; ++i) memcpy(tmp + i * SW, msg + (size_t)g.cn_edge[e0 + i] * SW, SW * sizeof(float));
          cn_update_w(tmp, sg, d, cn_mode, llr_max, offset);
          for (i = 0; i < d; ++i) memcpy(msg + (size_t)g.cn_edge[e0 + i] * SW, tmp + i * SW, SW * sizeof(float));
        }
        for (vv = 0; vv < N_vn; ++vv) {
          const int e0 = g.vn_ptr[vv], e1 = g.vn_ptr[vv + 1];
          float x[SW];
          for (j = 0; j < SW; ++j) x[j] = 0.f;
          for (i = e0; i < e1; ++i)
#pragma omp simd
            for (j = 0; j < SW; ++j) x[j] += msg[i * SW + j];
#pragma omp simd
          for (j = 0; j < SW; ++j) x[j] += l[vv * SW + j];
          for (i = e0; i < e1; ++i)
#pragma omp simd
            for (j = 0; j < SW; ++j) msg[i * SW + j] = vclampf(-1.f * msg[i * SW + j] + x[j], -llr_max, llr_max);
#pragma omp simd
          for (j = 0; j < SW; ++j) xh[vv * SW + j] = vclampf(x[j], -llr_max, llr_max);
        }
      }
      for (vv = 0; vv < N_vn; ++vv)
        for (j = 0; j < SW; ++j) {
          const int b = gi * SW + j;
          if (b < B) out[(size_t)b * N_vn + vv] = hard_out ? ((0.f >= xh[vv * SW + j]) ? 1.f : 0.f) : -1.f * xh[vv * SW + j];
        }
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
