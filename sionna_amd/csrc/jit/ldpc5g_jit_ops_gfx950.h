// Per-lane operations of the specialised 5G LDPC decoder as gfx950 code (source TEXT, compiled by hipRTC in front of
// ldpc5g_jit_templates.h and the generated per-wave programs; csrc/ldpc5g_jit.cpp).  A "lane value" is an ordinary
// float / unsigned here: one wavefront = 64 lifted copies of a base-graph node.
typedef float F32;
typedef unsigned U32;
typedef __attribute__((address_space(3))) float jit_lds_f32;
typedef float jit_f32x2 __attribute__((ext_vector_type(2)));
#define JIT_DEV static __device__ __forceinline__
#define JIT_INF __builtin_inff()
#define JIT_BLOCK ((int)blockIdx.x)
#define JIT_GRID ((int)gridDim.x)

JIT_DEV U32 jit_lane4() { return 4u * (threadIdx.x & 63u); }
JIT_DEV int jit_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// LDS is addressed by plain byte offsets: the kernel's only __shared__ object starts at offset 0 (checked at entry)
JIT_DEV F32 lds_ld(U32 a, unsigned off) { return *(jit_lds_f32*)(unsigned long)(a + off); }
// a 4-byte load the backend must not merge with a neighbour into a two-address form (ds_read2_b32: the two values would
// come back in neighbouring registers)
JIT_DEV F32 lds_ld_single(U32 a, unsigned off) { return *(volatile jit_lds_f32*)(unsigned long)(a + off); }
JIT_DEV void lds_st(U32 a, unsigned off, F32 v) { *(jit_lds_f32*)(unsigned long)(a + off) = v; }
// both values of an 8-byte slot with one DS instruction (interleaved layout).  volatile: the backend would merge two such
// accesses 512 bytes apart into ds_read2st64_b64 / ds_write2st64_b64, which move 16 bytes per lane at HALF the rate of two
// ds_read_b64 (MI355X_MICROARCH.md, LDS table: ds_read2_b64 8 cycles against 2 x 2); ordered accesses are not merged.
typedef __attribute__((address_space(3))) volatile jit_f32x2 jit_lds_f32x2;
JIT_DEV void lds_ld2(U32 a, unsigned off, F32& x0, F32& x1) {
  const jit_f32x2 v = *(jit_lds_f32x2*)(unsigned long)(a + off);
  x0 = v.x; x1 = v.y;
}
JIT_DEV void lds_st2(U32 a, unsigned off, F32 x0, F32 x1) { *(jit_lds_f32x2*)(unsigned long)(a + off) = jit_f32x2{x0, x1}; }
typedef bool M64;                                                      // one bit per lane (an SGPR pair)
JIT_DEV M64 u_testbit(U32 x, unsigned mask) { return (x & mask) != 0u; }
JIT_DEV F32 f_sel(M64 m, F32 a, F32 b) { return m ? a : b; }
// lane index, lane masks and per-lane integers of the any-lifting-size programs (jit/ldpc5g_jit_templates.h, JIT_GENERAL)
JIT_DEV unsigned jit_lane() { return threadIdx.x & 63u; }
#define JIT_IF(m) if (m) {                                             // lanes outside m execute nothing up to JIT_END
#define JIT_END }
JIT_DEV M64 u_lt(unsigned a, unsigned b) { return a < b; }
JIT_DEV M64 u_ge(unsigned a, unsigned b) { return a >= b; }
JIT_DEV M64 u_eq(unsigned a, unsigned b) { return a == b; }
JIT_DEV M64 m_and(M64 a, M64 b) { return (bool)((int)a & (int)b); }
JIT_DEV M64 m_not(M64 a) { return !a; }
JIT_DEV M64 m_xor_c(M64 a, bool c) { return a != c; }
JIT_DEV unsigned u_sel(M64 m, unsigned a, unsigned b) { return m ? a : b; }
JIT_DEV unsigned u_div(unsigned a, unsigned d) { return a / d; }
// f_eq / f_sel_m: a comparison whose lane mask is consumed several instructions later.  Written as volatile assembly so that the
// order of the generated text is the order of the instructions: the compiler's scheduler moves a v_cmp next to the v_cndmask that
// reads its mask (shortest scalar live range) and then has to pad the two wait states gfx950 wants between them with s_nop.
typedef unsigned long long M64S;
JIT_DEV M64S f_eq_abs(F32 a, F32 b) {                        // |a| == b
  M64S m;
  asm volatile("v_cmp_eq_f32_e64 %0, |%1|, %2" : "=s"(m) : "v"(a), "v"(b));
  return m;
}
JIT_DEV F32 f_sel_m(M64S m, F32 a, F32 b) {                 // m ? a : b
  F32 r;
  asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
JIT_DEV U32 u_shl(U32 a, int n) { return a << n; }
JIT_DEV U32 u_shr(U32 a, int n) { return a >> n; }
JIT_DEV F32 g_ld(const float* row, U32 voff, unsigned coff) { return *(const float*)((const char*)row + (voff + coff)); }
JIT_DEV void g_st(float* row, U32 voff, unsigned coff, F32 v) { *(float*)((char*)row + (voff + coff)) = v; }
// a message pair in the workgroup's row of the workspace (L2): 8-byte access at a per-lane byte offset
JIT_DEV void gm_ld2(const float* ws, U32 a, unsigned off, F32& x0, F32& x1) {
  const jit_f32x2 v = *(const jit_f32x2*)((const char*)ws + (a + off));
  x0 = v.x; x1 = v.y;
}
JIT_DEV void gm_st2(float* ws, U32 a, unsigned off, F32 x0, F32 x1) { *(jit_f32x2*)((char*)ws + (a + off)) = jit_f32x2{x0, x1}; }
// ... a phase then exchanges data through L2 as well: wait for the vector memory operations too
JIT_DEV void jit_barrier_g() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
JIT_DEV F32 g_ld_m(const float* row, U32 voff, M64 m) { return m ? *(const float*)((const char*)row + voff) : 0.f; }
JIT_DEV void g_st_m(float* row, U32 voff, M64 m, F32 v) {
  if (m) *(float*)((char*)row + voff) = v;
}
JIT_DEV unsigned jit_bcast_u(unsigned x) { return x; }
JIT_DEV F32 jit_bcast(float x) { return x; }
JIT_DEV F32 f_med3(F32 a, F32 b, F32 c) { return __builtin_amdgcn_fmed3f(a, b, c); }
JIT_DEV F32 f_abs(F32 a) { return __builtin_fabsf(a); }
JIT_DEV F32 f_neg(F32 a) { return -1.f * a; }
JIT_DEV F32 f_clamp(F32 x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }   // tf.clip_by_value
JIT_DEV F32 f_ge0_10(F32 x) { return (0.f >= x) ? 1.f : 0.f; }                                         // decoding.py:623-624
JIT_DEV F32 f_sel_gt(F32 a, F32 b, F32 x, F32 y) { return a > b ? x : y; }
JIT_DEV F32 f_sel_eq(F32 a, F32 b, F32 x, F32 y) { return a == b ? x : y; }
JIT_DEV U32 f_bits(F32 a) { return __builtin_bit_cast(unsigned, a); }
JIT_DEV F32 u_float(U32 a) { return __builtin_bit_cast(float, a); }
JIT_DEV U32 u_min(U32 a, U32 b) { return a < b ? a : b; }
// the value itself, opaque to the optimiser (no instruction beyond a register copy): what is derived from it is computed
// where it is written instead of being hoisted out of the loops
JIT_DEV U32 u_here(U32 a) {
  asm volatile("" : "+v"(a));
  return a;
}
// a ^ 256, computed where it is written (volatile: not hoisted out of the iteration loop into a long-lived register)
JIT_DEV U32 u_xor256_here(U32 a) {
  U32 r;
  asm volatile("v_xor_b32_e32 %0, 0x100, %1" : "=v"(r) : "v"(a));
  return r;
}
// a ^ 4 / a & ~4 (the other half of an 8-byte slot / the slot itself), computed where written
JIT_DEV U32 u_xor4_here(U32 a) {
  U32 r;
  asm volatile("v_xor_b32_e32 %0, 4, %1" : "=v"(r) : "v"(a));
  return r;
}
JIT_DEV U32 u_andn4_here(U32 a) {
  U32 r;
  asm volatile("v_and_b32_e32 %0, 0xfffffffb, %1" : "=v"(r) : "v"(a));
  return r;
}
// inside an if-block on a wave-uniform condition: keeps it a branch (the optimiser would turn the block into selections
// executed on every path)
#define JIT_KEEP_BRANCH() asm volatile("" ::: "memory")
JIT_DEV U32 u_xor3(U32 a, U32 b, U32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
JIT_DEV U32 u_xor_and(U32 m, U32 v, unsigned k) { return __builtin_amdgcn_bitop3_b32(m, v, k, 0x78); }   // m ^ (v & k)
// two IEEE additions / subtractions in one issue slot (v_pk_add_f32)
JIT_DEV void f_pk_add(F32& x0, F32& x1, F32 c0, F32 c1) {
  jit_f32x2 x = {x0, x1};
  x += jit_f32x2{c0, c1};
  x0 = x.x; x1 = x.y;
}
JIT_DEV void f_pk_sub(F32& e0, F32& e1, F32 x0, F32 x1, F32 c0, F32 c1) {
  const jit_f32x2 e = jit_f32x2{x0, x1} - jit_f32x2{c0, c1};
  e0 = e.x; e1 = e.y;
}
// packed fused multiply-adds (v_pk_fma_f32): (d0, d1) = (a0, a1) * (b0, b1) + (c0, c1); scalar operands are broadcast
JIT_DEV void f_pk_fma(F32& d0, F32& d1, F32 a0, F32 a1, F32 b0, F32 b1, F32 c0, F32 c1) {
  const jit_f32x2 d = __builtin_elementwise_fma(jit_f32x2{a0, a1}, jit_f32x2{b0, b1}, jit_f32x2{c0, c1});
  d0 = d.x; d1 = d.y;
}
JIT_DEV void f_pk_fma(F32& d0, F32& d1, F32 a0, F32 a1, float b, F32 c0, F32 c1) { f_pk_fma(d0, d1, a0, a1, b, b, c0, c1); }
JIT_DEV void f_pk_mul(F32& d0, F32& d1, F32 a0, F32 a1, F32 b0, F32 b1) {
  const jit_f32x2 d = jit_f32x2{a0, a1} * jit_f32x2{b0, b1};
  d0 = d.x; d1 = d.y;
}
JIT_DEV void f_pk_addc(F32& d0, F32& d1, F32 a0, F32 a1, float c) {
  const jit_f32x2 d = jit_f32x2{a0, a1} + jit_f32x2{c, c};
  d0 = d.x; d1 = d.y;
}
#define JIT_TABLE __device__ const
JIT_DEV void jit_tab_lane(F32& a, F32& b, const float (*tab)[2]) { a = tab[threadIdx.x & 63u][0]; b = tab[threadIdx.x & 63u][1]; }
JIT_DEV F32 f_frexp_exp(F32 x) { return (float)__builtin_amdgcn_frexp_expf(x); }      // exponent e + 1 of a normal x, as a float
JIT_DEV F32 f_min(F32 a, float b) { return __builtin_fminf(a, b); }
JIT_DEV U32 u_and_or(U32 a, unsigned m, unsigned o) { return (a & m) | o; }
JIT_DEV U32 u_msb_if_neg(F32 v) { return (v < 0.f) ? 0x80000000u : 0u; }
JIT_DEV U32 u_msb_nonzero(F32 v) { return __builtin_bit_cast(unsigned, v) & 0x80000000u; }   // = u_msb_if_neg for every v but -0
// the message image of a workgroup pass between LDS and a caller's buffer (state variant): the calling wave's share, 8 bytes
// per lane and step; rank / nranks = this wave among the waves that copy
JIT_DEV void jit_copy_g2l(const float* g, unsigned lds_byte, unsigned nbytes, int rank, int nranks) {
  for (unsigned off = ((unsigned)rank * 64u + (threadIdx.x & 63u)) * 8u; off < nbytes; off += (unsigned)nranks * 512u) {
    const jit_f32x2 v = *(const jit_f32x2*)((const char*)g + off);
    *(jit_lds_f32x2*)(unsigned long)(lds_byte + off) = v;
  }
}
JIT_DEV void jit_copy_l2g(float* g, unsigned lds_byte, unsigned nbytes, int rank, int nranks) {
  for (unsigned off = ((unsigned)rank * 64u + (threadIdx.x & 63u)) * 8u; off < nbytes; off += (unsigned)nranks * 512u) {
    const jit_f32x2 v = *(jit_lds_f32x2*)(unsigned long)(lds_byte + off);
    *(jit_f32x2*)((char*)g + off) = v;
  }
}
// a phase exchanges LDS data only: wait for this wave's LDS operations, then the workgroup barrier
JIT_DEV void jit_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int P>
JIT_DEV void jit_setprio() { __builtin_amdgcn_s_setprio(P); }
