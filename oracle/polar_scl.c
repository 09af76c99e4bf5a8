/* Oracle (plain C, CPU): successive-cancellation LIST decoding of Polar codes.
 *
 * TEST INFRASTRUCTURE - see oracle/__init__.py.  Restates PolarSCLDecoder of the reference
 * (/root/reference/src/sionna/phy/fec/polar/decoding.py:367-1420; structure cited in polar_scl_body.inc) twice:
 *
 *  precision 1 (float64): the arithmetic of the reference's own NumPy twin (_decode_np_batch, float64 arrays, libm
 *      exp/log, the literal formulas log(1+e^(x+y)) - log(e^x+e^y) and log(1+e^-x)).  Pinned by fixtures generated
 *      from that twin itself (tools/gen_polar_scl_golden.py executes the reference source; tests/golden/
 *      polar_scl_np_golden.npz): identical candidate lists, path metrics to 1e-9.
 *  precision 0 (float32): the same decoder in single precision with a DEFINED arithmetic, so that a float32
 *      implementation can be compared bit for bit (the reference's float32 TensorFlow path uses tf.math.softplus /
 *      reduce_logsumexp / reduce_sum, whose roundings and summation order are not part of its contract):
 *        T(a)        = log(1 + e^-a), a >= 0: t = a * (-log2 e); r = rint(t); f = t - r; e = ldexp(1 + f E(f), r);
 *                      T = e Q(e) with the float32 polynomials of tools/fit_scl_math.py, Horner with fma
 *                      (|error| < 2e-7, tests/test_oracle_polar_scl.py)
 *        softplus(x) = max(x, 0) + T(|x|)                               (tf.math.softplus)
 *        cn_op(x, y) = softplus(x + y) - (max(x, y) + T(|x - y|))       (x, y clipped to +-30; :684-705)
 *        vn_op       = (1 - 2u) x + y, product rounded, then sum         (:707-714)
 *        block sums of rate-0 / repetition nodes over m = 2^s terms: 64 partial sums p[l] = v[l] + v[l+64] +
 *                      v[l+128] + ... (ascending), then the halving tree p[i] += p[i+h], h = 32, 16, ..., 1
 *        sorting     = stable (ties keep the lower position), like tf.argsort / the insertion sort NumPy uses for
 *                      16 elements
 *      csrc/polar.hip follows this definition (tests/test_gpu_polar.py compares hard decisions and CRC status bit
 *      for bit).
 * Both share one body (polar_scl_body.inc), so the fixtures of the float64 twin pin the control flow of both. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LLR_MAX 30.0f

/* ---- float32 specification arithmetic */
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float oracle_scl_T_f32(float a) {
  const float t = a * -1.44269504f;
  const float r = rintf(t);
  const float f = t - r;
  float p = u2f(0x392209c5u);
  p = fmaf(p, f, u2f(0x3aaf8448u));
  p = fmaf(p, f, u2f(0x3c1d952au));
  p = fmaf(p, f, u2f(0x3d6357b6u));
  p = fmaf(p, f, u2f(0x3e75fdf0u));
  p = fmaf(p, f, u2f(0x3f317218u));
  p = fmaf(p, f, 1.0f);
  const float e = ldexpf(p, (int)r);
  float q = u2f(0x3ba7f8dcu);
  q = fmaf(q, e, u2f(0xbcee2cbcu));
  q = fmaf(q, e, u2f(0x3d9ec0c1u));
  q = fmaf(q, e, u2f(0xbe0b497au));
  q = fmaf(q, e, u2f(0x3e4358e6u));
  q = fmaf(q, e, u2f(0xbe7e5082u));
  q = fmaf(q, e, u2f(0x3eaa96bau));
  q = fmaf(q, e, u2f(0xbeffff46u));
  q = fmaf(q, e, u2f(0x3f7fffffu));
  return e * q;
}
static inline float clip_f32(float x) { return fminf(fmaxf(x, -LLR_MAX), LLR_MAX); }
static inline float softplus_f32(float x) { return fmaxf(x, 0.0f) + oracle_scl_T_f32(fabsf(x)); }
static inline float cn_op_f32(float x, float y) {
  x = clip_f32(x);
  y = clip_f32(y);
  const float lse = fmaxf(x, y) + oracle_scl_T_f32(fabsf(x - y));
  return softplus_f32(x + y) - lse;
}
static float block_sum_f32(const float* v, int m) {
  float p[64];
  for (int l = 0; l < 64; ++l) {
    p[l] = l < m ? v[l] : 0.0f;
    for (int j = l + 64; j < m; j += 64) p[l] = p[l] + v[j];
  }
  for (int h = 32; h >= 1; h >>= 1)
    for (int i = 0; i < h; ++i) p[i] = p[i] + p[i + h];
  return p[0];
}
float oracle_scl_softplus_f32(float x) { return softplus_f32(x); }
float oracle_scl_cn_op_f32(float x, float y) { return cn_op_f32(x, y); }
float oracle_scl_block_sum_f32(const float* v, int m) { return block_sum_f32(v, m); }

/* ---- float64 arithmetic of the NumPy twin (literal formulas) */
static inline double clip_f64(double x) { return fmax(fmin(x, (double)LLR_MAX), -(double)LLR_MAX); }
static inline double softplus_f64(double x) { return log(1.0 + exp(x)); }                  /* log(1 + exp(-llr)) */
static inline double cn_op_f64(double x, double y) {
  x = clip_f64(x);
  y = clip_f64(y);
  double o = log(1.0 + exp(x + y));
  o -= log(exp(x) + exp(y));
  return o;
}
static double block_sum_f64(const double* v, int m) {
  double s = 0.0;
  for (int i = 0; i < m; ++i) s += v[i];
  return s;
}

#define REAL float
#define NAME(x) x##_f32
#define CLIP clip_f32
#define SOFTPLUS softplus_f32
#define CN_OP cn_op_f32
#define BLOCK_SUM block_sum_f32
#include "polar_scl_body.inc"
#undef REAL
#undef NAME
#undef CLIP
#undef SOFTPLUS
#undef CN_OP
#undef BLOCK_SUM

#define REAL double
#define NAME(x) x##_f64
#define CLIP clip_f64
#define SOFTPLUS softplus_f64
#define CN_OP cn_op_f64
#define BLOCK_SUM block_sum_f64
#include "polar_scl_body.inc"

/* ---- PolarSCDecoder.call (decoding.py:122-263) in the float32 specification arithmetic: u[lo..lo+m) and the
 * partial sums x[lo..lo+m) of the sub-block whose LLRs are l[0..m) */
static void sc_rec_f32(const int32_t* frozen, int lo, int m, const float* l, uint8_t* u, uint8_t* x) {
  if (m > 1) {
    int nf = 0;
    for (int i = 0; i < m; ++i) nf += frozen[lo + i];
    if (nf == m) {                                                       /* rate-0: all zero (:170-176) */
      memset(u + lo, 0, m);
      memset(x + lo, 0, m);
      return;
    }
    const int h = m / 2;
    float buf[512];
    for (int i = 0; i < h; ++i) buf[i] = cn_op_f32(l[i], l[h + i]);
    sc_rec_f32(frozen, lo, h, buf, u, x);
    for (int i = 0; i < h; ++i) buf[i] = (1.0f - 2.0f * (float)x[lo + i]) * l[i] + l[h + i];
    sc_rec_f32(frozen, lo + h, h, buf, u, x);
    for (int i = 0; i < h; ++i) x[lo + i] ^= x[lo + h + i];
  } else if (frozen[lo]) {
    u[lo] = x[lo] = 0;
  } else {
    u[lo] = x[lo] = l[0] > 0.0f ? 0 : 1;                                 /* 0.5 (1 - sign(l)), an exact zero -> 1 */
  }
}

/* logits [B][n] -> u_hat [B][n] (all positions; frozen ones are 0) */
int oracle_polar_sc_decode(int n, const int32_t* frozen, const float* logits, int batch, uint8_t* u_hat) {
  if (n < 2 || n > 1024 || (n & (n - 1)) || batch < 0) return -1;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b) {
    float l[1024];
    uint8_t x[1024];
    for (int i = 0; i < n; ++i) l[i] = -1.0f * logits[(size_t)b * n + i];
    sc_rec_f32(frozen, 0, n, l, u_hat + (size_t)b * n, x);
  }
  return 0;
}

/* the same decoder in float64 with the literal boxplus of the reference's NumPy twin (precision 1 above): the specification of
 * PolarSCDecoder(precision="double") */
static void sc_rec_f64(const int32_t* frozen, int lo, int m, const double* l, uint8_t* u, uint8_t* x) {
  if (m > 1) {
    int nf = 0;
    for (int i = 0; i < m; ++i) nf += frozen[lo + i];
    if (nf == m) {
      memset(u + lo, 0, m);
      memset(x + lo, 0, m);
      return;
    }
    const int h = m / 2;
    double buf[512];
    for (int i = 0; i < h; ++i) buf[i] = cn_op_f64(l[i], l[h + i]);
    sc_rec_f64(frozen, lo, h, buf, u, x);
    for (int i = 0; i < h; ++i) buf[i] = (1.0 - 2.0 * (double)x[lo + i]) * l[i] + l[h + i];
    sc_rec_f64(frozen, lo + h, h, buf, u, x);
    for (int i = 0; i < h; ++i) x[lo + i] ^= x[lo + h + i];
  } else if (frozen[lo]) {
    u[lo] = x[lo] = 0;
  } else {
    u[lo] = x[lo] = l[0] > 0.0 ? 0 : 1;
  }
}

int oracle_polar_sc_decode_f64(int n, const int32_t* frozen, const double* logits, int batch, uint8_t* u_hat) {
  if (n < 2 || n > 1024 || (n & (n - 1)) || batch < 0) return -1;
#pragma omp parallel for schedule(static)
  for (int b = 0; b < batch; ++b) {
    double l[1024];
    uint8_t x[1024];
    for (int i = 0; i < n; ++i) l[i] = -1.0 * logits[(size_t)b * n + i];
    sc_rec_f64(frozen, 0, n, l, u_hat + (size_t)b * n, x);
  }
  return 0;
}

int oracle_polar_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* logits [B][n] -> uhat_list [B][2L][n], pm [B][2L].  Returns 0, or -1 for bad arguments. */
int oracle_polar_scl_decode(int n, int list_size, const int32_t* frozen, int use_fast_scl, int precision,
                            const float* logits, int batch, uint8_t* uhat_list, double* pm, int nthreads) {
  if (n < 2 || n > 1024 || (n & (n - 1)) || list_size < 1 || list_size > 32 || batch < 0) return -1;
  int stages = 0;
  while ((1 << stages) < n) ++stages;
  const size_t per = 2 * (size_t)list_size * (stages + 1) * n;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  int fail = 0;
#pragma omp parallel
  {
    void* llr_buf = malloc(per * sizeof(double));
    uint8_t* uhat_buf = (uint8_t*)malloc(per);
    if (!llr_buf || !uhat_buf) {
#pragma omp atomic write
      fail = 1;
    } else {
#pragma omp for schedule(dynamic, 1)
      for (int b = 0; b < batch; ++b) {
        uint8_t* ul = uhat_list + (size_t)b * 2 * list_size * n;
        double* pmo = pm + (size_t)b * 2 * list_size;
        if (precision == 0)
          decode_one_f32(n, list_size, frozen, use_fast_scl, logits + (size_t)b * n, ul, pmo, (float*)llr_buf, uhat_buf);
        else
          decode_one_f64(n, list_size, frozen, use_fast_scl, logits + (size_t)b * n, ul, pmo, (double*)llr_buf, uhat_buf);
      }
    }
    free(llr_buf);
    free(uhat_buf);
  }
  return fail ? -2 : 0;
}
