// TEST INFRASTRUCTURE (not part of the product): 64-wide CPU stand-ins for the per-lane operations of the specialised
// 5G LDPC decoder, so that the source libsionna_amd.so GENERATES for a code (samd_ldpc5g_jit_source with with_ops = 0:
// csrc/jit/ldpc5g_jit_templates.h + the per-wave programs) can be compiled with g++ and run against the C oracle on a
// machine without a GPU (tests/test_jit_emu.py).  The gfx950 definitions of the same names are
// sionna_amd/csrc/jit/ldpc5g_jit_ops_gfx950.h.  One "wave" = one host thread, LDS = one shared byte array, the workgroup
// barrier = a pthread barrier; float arithmetic is IEEE single precision (compile with -ffp-contract=off).
#pragma once
#include <pthread.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

struct F32 {
  float v[64];
  F32() {}
  F32(float x) { for (int i = 0; i < 64; ++i) v[i] = x; }
};
struct U32 {
  unsigned v[64];
  U32() {}
  U32(unsigned x) { for (int i = 0; i < 64; ++i) v[i] = x; }
};
#define JIT_EMU_BIN(T, op)                                                       \
  static inline T operator op(const T& a, const T& b) {                          \
    T r;                                                                         \
    for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] op b.v[i];                      \
    return r;                                                                    \
  }
JIT_EMU_BIN(F32, +) JIT_EMU_BIN(F32, -) JIT_EMU_BIN(F32, *)
JIT_EMU_BIN(U32, +) JIT_EMU_BIN(U32, -) JIT_EMU_BIN(U32, *) JIT_EMU_BIN(U32, ^) JIT_EMU_BIN(U32, &) JIT_EMU_BIN(U32, |)
#undef JIT_EMU_BIN

struct JitEmuCtx {
  unsigned char* lds;
  size_t lds_bytes;
  pthread_barrier_t* bar;
  int block, grid;
  size_t ws_bytes;      // bytes of ONE workgroup's row of the workspace
};
static thread_local JitEmuCtx jit_emu_ctx;
// execution mask: JIT_IF(m) ... JIT_END switches the lanes outside m off for every memory access in between (their
// register values are don't-cares, as on the GPU)
static thread_local bool jit_emu_exec[64] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1,
                                             1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

#define JIT_DEV static inline
#define JIT_INF std::numeric_limits<float>::infinity()
#define JIT_BLOCK (jit_emu_ctx.block)
#define JIT_GRID (jit_emu_ctx.grid)

JIT_DEV U32 jit_lane4() {
  U32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = 4u * (unsigned)i;
  return r;
}
JIT_DEV F32 lds_ld(const U32& a, unsigned off) {
  F32 r;
  for (int i = 0; i < 64; ++i) {
    r.v[i] = 0.f;
    if (!jit_emu_exec[i]) continue;
    const size_t p = (size_t)a.v[i] + off;
    if (p + 4 > jit_emu_ctx.lds_bytes || (p & 3)) __builtin_trap();
    memcpy(&r.v[i], jit_emu_ctx.lds + p, 4);
  }
  return r;
}
JIT_DEV F32 lds_ld_single(const U32& a, unsigned off) { return lds_ld(a, off); }
JIT_DEV void lds_st(const U32& a, unsigned off, const F32& x) {
  for (int i = 0; i < 64; ++i) {
    if (!jit_emu_exec[i]) continue;
    const size_t p = (size_t)a.v[i] + off;
    if (p + 4 > jit_emu_ctx.lds_bytes || (p & 3)) __builtin_trap();
    memcpy(jit_emu_ctx.lds + p, &x.v[i], 4);
  }
}
JIT_DEV void lds_ld2(const U32& a, unsigned off, F32& x0, F32& x1) {
  for (int i = 0; i < 64; ++i) {
    x0.v[i] = x1.v[i] = 0.f;
    if (!jit_emu_exec[i]) continue;
    const size_t p = (size_t)a.v[i] + off;
    if (p + 8 > jit_emu_ctx.lds_bytes || (p & 7)) __builtin_trap();
    memcpy(&x0.v[i], jit_emu_ctx.lds + p, 4);
    memcpy(&x1.v[i], jit_emu_ctx.lds + p + 4, 4);
  }
}
JIT_DEV void lds_st2(const U32& a, unsigned off, const F32& x0, const F32& x1) {
  for (int i = 0; i < 64; ++i) {
    if (!jit_emu_exec[i]) continue;
    const size_t p = (size_t)a.v[i] + off;
    if (p + 8 > jit_emu_ctx.lds_bytes || (p & 7)) __builtin_trap();
    memcpy(jit_emu_ctx.lds + p, &x0.v[i], 4);
    memcpy(jit_emu_ctx.lds + p + 4, &x1.v[i], 4);
  }
}
struct M64 { bool v[64]; };
JIT_DEV M64 u_testbit(const U32& x, unsigned mask) {
  M64 r;
  for (int i = 0; i < 64; ++i) r.v[i] = (x.v[i] & mask) != 0u;
  return r;
}
JIT_DEV F32 f_sel(const M64& m, const F32& a, const F32& b) {
  F32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = m.v[i] ? a.v[i] : b.v[i];
  return r;
}
typedef M64 M64S;
struct JitEmuExec {
  bool saved[64];
  explicit JitEmuExec(const M64& m) {
    for (int i = 0; i < 64; ++i) { saved[i] = jit_emu_exec[i]; jit_emu_exec[i] = saved[i] && m.v[i]; }
  }
  ~JitEmuExec() { for (int i = 0; i < 64; ++i) jit_emu_exec[i] = saved[i]; }
};
#define JIT_IF(m) { JitEmuExec jit_emu_scope(m);
#define JIT_END }
JIT_DEV U32 jit_lane() {
  U32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = (unsigned)i;
  return r;
}
#define JIT_EMU_CMP(name, op)                                \
  JIT_DEV M64 name(const U32& a, const U32& b) {             \
    M64 r;                                                   \
    for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] op b.v[i];  \
    return r;                                                \
  }
JIT_EMU_CMP(u_lt, <) JIT_EMU_CMP(u_ge, >=) JIT_EMU_CMP(u_eq, ==)
#undef JIT_EMU_CMP
JIT_DEV M64 m_and(const M64& a, const M64& b) {
  M64 r;
  for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] && b.v[i];
  return r;
}
JIT_DEV M64 m_not(const M64& a) {
  M64 r;
  for (int i = 0; i < 64; ++i) r.v[i] = !a.v[i];
  return r;
}
JIT_DEV M64 m_xor_c(const M64& a, bool c) {
  M64 r;
  for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] != c;
  return r;
}
JIT_DEV U32 u_sel(const M64& m, const U32& a, const U32& b) {
  U32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = m.v[i] ? a.v[i] : b.v[i];
  return r;
}
JIT_DEV U32 u_div(const U32& a, unsigned d) {
  U32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] / d;
  return r;
}
JIT_DEV M64S f_eq_abs(const F32& a, const F32& b) {
  M64 r;
  for (int i = 0; i < 64; ++i) r.v[i] = fabsf(a.v[i]) == b.v[i];
  return r;
}
JIT_DEV F32 f_sel_m(const M64S& m, const F32& a, const F32& b) { return f_sel(m, a, b); }
JIT_DEV U32 u_shl(const U32& a, int n) {
  U32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] << n;
  return r;
}
JIT_DEV U32 u_shr(const U32& a, int n) {
  U32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] >> n;
  return r;
}
JIT_DEV F32 g_ld(const float* row, const U32& voff, unsigned coff) {
  F32 r;
  for (int i = 0; i < 64; ++i) memcpy(&r.v[i], (const char*)row + (voff.v[i] + coff), 4);
  return r;
}
JIT_DEV void g_st(float* row, const U32& voff, unsigned coff, const F32& x) {
  for (int i = 0; i < 64; ++i) memcpy((char*)row + (voff.v[i] + coff), &x.v[i], 4);
}
JIT_DEV void gm_ld2(const float* ws, const U32& a, unsigned off, F32& x0, F32& x1) {
  for (int i = 0; i < 64; ++i) {
    x0.v[i] = x1.v[i] = 0.f;
    if (!jit_emu_exec[i]) continue;
    const size_t p = (size_t)a.v[i] + off;
    if (p + 8 > jit_emu_ctx.ws_bytes || (p & 7)) __builtin_trap();
    memcpy(&x0.v[i], (const char*)ws + p, 4);
    memcpy(&x1.v[i], (const char*)ws + p + 4, 4);
  }
}
JIT_DEV void gm_st2(float* ws, const U32& a, unsigned off, const F32& x0, const F32& x1) {
  for (int i = 0; i < 64; ++i) {
    if (!jit_emu_exec[i]) continue;
    const size_t p = (size_t)a.v[i] + off;
    if (p + 8 > jit_emu_ctx.ws_bytes || (p & 7)) __builtin_trap();
    memcpy((char*)ws + p, &x0.v[i], 4);
    memcpy((char*)ws + p + 4, &x1.v[i], 4);
  }
}
JIT_DEV F32 g_ld_m(const float* row, const U32& voff, const M64& m) {
  F32 r;
  for (int i = 0; i < 64; ++i) {
    r.v[i] = 0.f;
    if (m.v[i] && jit_emu_exec[i]) memcpy(&r.v[i], (const char*)row + voff.v[i], 4);
  }
  return r;
}
JIT_DEV void g_st_m(float* row, const U32& voff, const M64& m, const F32& x) {
  for (int i = 0; i < 64; ++i)
    if (m.v[i] && jit_emu_exec[i]) memcpy((char*)row + voff.v[i], &x.v[i], 4);
}
JIT_DEV U32 jit_bcast_u(unsigned x) { return U32(x); }
JIT_DEV F32 jit_bcast(float x) { return F32(x); }
JIT_DEV float jit_emu_med3(float a, float b, float c) {      // v_med3_f32 on ordinary values: the middle operand itself
  if (a <= b) return b <= c ? b : (a <= c ? c : a);
  return a <= c ? a : (b <= c ? c : b);
}
JIT_DEV F32 f_med3(const F32& a, const F32& b, const F32& c) {
  F32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = jit_emu_med3(a.v[i], b.v[i], c.v[i]);
  return r;
}
JIT_DEV F32 f_abs(const F32& a) {
  F32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = fabsf(a.v[i]);
  return r;
}
JIT_DEV F32 f_neg(const F32& a) { return F32(-1.f) * a; }
JIT_DEV F32 f_clamp(const F32& x, float lo, float hi) {
  F32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = fminf(fmaxf(x.v[i], lo), hi);
  return r;
}
JIT_DEV F32 f_ge0_10(const F32& x) {
  F32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = (0.f >= x.v[i]) ? 1.f : 0.f;
  return r;
}
JIT_DEV F32 f_sel_gt(const F32& a, const F32& b, const F32& x, const F32& y) {
  F32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] > b.v[i] ? x.v[i] : y.v[i];
  return r;
}
JIT_DEV F32 f_sel_eq(const F32& a, const F32& b, const F32& x, const F32& y) {
  F32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] == b.v[i] ? x.v[i] : y.v[i];
  return r;
}
JIT_DEV U32 f_bits(const F32& a) {
  U32 r;
  memcpy(r.v, a.v, sizeof(r.v));
  return r;
}
JIT_DEV F32 u_float(const U32& a) {
  F32 r;
  memcpy(r.v, a.v, sizeof(r.v));
  return r;
}
JIT_DEV U32 u_min(const U32& a, const U32& b) {
  U32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = a.v[i] < b.v[i] ? a.v[i] : b.v[i];
  return r;
}
JIT_DEV U32 u_here(const U32& a) { return a; }
JIT_DEV U32 u_xor256_here(const U32& a) { return a ^ U32(256u); }
JIT_DEV U32 u_xor4_here(const U32& a) { return a ^ U32(4u); }
JIT_DEV U32 u_andn4_here(const U32& a) { return a & U32(0xfffffffbu); }
#define JIT_KEEP_BRANCH() do { } while (0)
JIT_DEV U32 u_xor3(const U32& a, const U32& b, const U32& c) { return a ^ b ^ c; }
JIT_DEV U32 u_xor_and(const U32& m, const U32& v, unsigned k) { return m ^ (v & U32(k)); }
JIT_DEV void f_pk_add(F32& x0, F32& x1, const F32& c0, const F32& c1) { x0 = x0 + c0; x1 = x1 + c1; }
JIT_DEV void f_pk_sub(F32& e0, F32& e1, const F32& x0, const F32& x1, const F32& c0, const F32& c1) { e0 = x0 - c0; e1 = x1 - c1; }
JIT_DEV void f_pk_fma(F32& d0, F32& d1, const F32& a0, const F32& a1, const F32& b0, const F32& b1, const F32& c0, const F32& c1) {
  F32 r0, r1;
  for (int i = 0; i < 64; ++i) { r0.v[i] = fmaf(a0.v[i], b0.v[i], c0.v[i]); r1.v[i] = fmaf(a1.v[i], b1.v[i], c1.v[i]); }
  d0 = r0; d1 = r1;
}
JIT_DEV void f_pk_fma(F32& d0, F32& d1, const F32& a0, const F32& a1, float b, const F32& c0, const F32& c1) {
  f_pk_fma(d0, d1, a0, a1, F32(b), F32(b), c0, c1);
}
JIT_DEV void f_pk_mul(F32& d0, F32& d1, const F32& a0, const F32& a1, const F32& b0, const F32& b1) {
  const F32 r0 = a0 * b0, r1 = a1 * b1;
  d0 = r0; d1 = r1;
}
JIT_DEV void f_pk_addc(F32& d0, F32& d1, const F32& a0, const F32& a1, float c) {
  const F32 r0 = a0 + F32(c), r1 = a1 + F32(c);
  d0 = r0; d1 = r1;
}
#define JIT_TABLE const
JIT_DEV void jit_tab_lane(F32& a, F32& b, const float (*tab)[2]) {
  for (int i = 0; i < 64; ++i) { a.v[i] = tab[i][0]; b.v[i] = tab[i][1]; }
}
JIT_DEV F32 f_frexp_exp(const F32& x) {
  F32 r;
  for (int i = 0; i < 64; ++i) { unsigned b; memcpy(&b, &x.v[i], 4); r.v[i] = (float)((int)((b >> 23) & 0xffu) - 126); }
  return r;
}
JIT_DEV F32 f_min(const F32& a, float b) {
  F32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = fminf(a.v[i], b);
  return r;
}
JIT_DEV U32 u_and_or(const U32& a, unsigned m, unsigned o) { return (a & U32(m)) | U32(o); }
JIT_DEV U32 u_msb_if_neg(const F32& v) {
  U32 r;
  for (int i = 0; i < 64; ++i) r.v[i] = (v.v[i] < 0.f) ? 0x80000000u : 0u;
  return r;
}
JIT_DEV U32 u_msb_nonzero(const F32& v) {                      // the sign bit; the generated code relies on "never -0": checked here
  U32 r;
  for (int i = 0; i < 64; ++i) {
    unsigned b;
    memcpy(&b, &v.v[i], 4);
    if (b == 0x80000000u && jit_emu_exec[i]) __builtin_trap();
    r.v[i] = b & 0x80000000u;
  }
  return r;
}
JIT_DEV void jit_copy_g2l(const float* g, unsigned lds_byte, unsigned nbytes, int rank, int nranks) {
  for (int lane = 0; lane < 64; ++lane)
    for (unsigned off = ((unsigned)rank * 64u + lane) * 8u; off < nbytes; off += (unsigned)nranks * 512u) {
      if ((size_t)lds_byte + off + 8 > jit_emu_ctx.lds_bytes) __builtin_trap();
      memcpy(jit_emu_ctx.lds + lds_byte + off, (const char*)g + off, 8);
    }
}
JIT_DEV void jit_copy_l2g(float* g, unsigned lds_byte, unsigned nbytes, int rank, int nranks) {
  for (int lane = 0; lane < 64; ++lane)
    for (unsigned off = ((unsigned)rank * 64u + lane) * 8u; off < nbytes; off += (unsigned)nranks * 512u) {
      if ((size_t)lds_byte + off + 8 > jit_emu_ctx.lds_bytes) __builtin_trap();
      memcpy((char*)g + off, jit_emu_ctx.lds + lds_byte + off, 8);
    }
}
JIT_DEV void jit_barrier() { pthread_barrier_wait(jit_emu_ctx.bar); }
JIT_DEV void jit_barrier_g() { pthread_barrier_wait(jit_emu_ctx.bar); }
template <int P>
JIT_DEV void jit_setprio() {}
