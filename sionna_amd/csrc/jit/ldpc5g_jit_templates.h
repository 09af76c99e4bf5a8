// Node updates of the SPECIALISED 5G LDPC decoder (csrc/ldpc5g_jit.cpp), as source TEXT: this file is embedded in
// libsionna_amd.so and compiled at run time (hipRTC, --offload-arch=gfx950) together with the per-wave programs the
// library generates for ONE code (base graph, lifting size, rate matching).  It is written against a small set of
// per-lane operations (F32 / U32 values, lds_ld / lds_st, f_med3, ...; gfx950 definitions: ldpc5g_jit_ops_gfx950.h) so
// that tests/jit_emu can compile the SAME text with 64-wide CPU stand-ins and run the generated programs against the
// C oracle without a GPU.  Uniform (per-wave) quantities are plain float / unsigned / int / bool.
//
// Arithmetic = csrc/ldpc5g_onchip_ms.inc (ms_cn_row with VAR 1, ms_vn_col), operation for operation: min-sum /
// offset-min-sum of decoding.py:755-953 and vn_update_sum of decoding.py:681-732, hence the same bits as the oracle.
// JIT_Z4 = 4 Z (bytes of one edge block); message layout as in ldpc5g_onchip_bp.hip: block (row, position in the row),
// indexed by the CHECK node's lifted copy.
// Value arrays are [edge][chunk]: the two chunks of an edge - the operands of one packed-fp32 operation - are neighbours
// (with [chunk][edge] the optimiser forms overlapping vector accesses and leaves the arrays in scratch memory).

#define JIT_NOOUT 0xFFFFFFFFu
// JIT_ABL (development builds of the library only - results are WRONG by construction; tools/gpu_jit_ablation.sh): bit 0 =
// no check-node arithmetic (messages stored back as loaded), bit 1 = no variable-node arithmetic, bit 2 = no workgroup
// barriers in the iteration loop.  What remains is timed unchanged: where an iteration's time goes.
#ifndef JIT_ABL
#define JIT_ABL 0
#endif
// JIT_OFFSET0 = 1: the kernel is generated for plain min-sum (offset 0 exactly); 0: offset-min-sum with a run-time offset
// JIT_LAYOUT_B (Z = 128 only): the two 64-lane chunks of an edge block INTERLEAVED - lifted copy z of a check node at byte
// 8 (z mod 64) + 4 (z div 64) of its 512-byte block - so that a pair item moves both of a lane's values with ONE 8-byte DS
// instruction in each phase (ds_read_b64: 2 LDS-pipeline cycles per 512 bytes instead of 4; ds_write_b64: 6 instead of 8
// for two ds_write_b32).  A variable-node lane reads the pair of slot (lane - shift) mod 64; its own two nodes (lane, lane + 64)
// are that pair in this order or swapped - for the lanes below the shift - which costs two selections in and two out per
// edge.  JIT_CH = byte distance between a lane's two chunk values (256 chunk-planar, 4 interleaved).
#ifndef JIT_LAYOUT_B
#define JIT_LAYOUT_B 0
#endif
#define JIT_CH (JIT_LAYOUT_B ? 4u : 256u)

// value written to the output tensor for a VN total x (decoding.py:620-626): clip, then hard decision or the logit
JIT_DEV F32 jit_outval(F32 x, float llr_max, int hard_out) {
  const F32 xc = f_clamp(x, -llr_max, llr_max);
  return hard_out ? f_ge0_10(xc) : f_neg(xc);
}

// channel LLR of one VN chunk (decoding.py:552-565): clip, logit -> LLR, "+ 0" so that no LLR is -0
JIT_DEV F32 jit_chan(F32 raw, float llr_max) { return f_neg(f_clamp(raw, -llr_max, llr_max)) + 0.f; }

// ((l4 + k) mod 4Z) + base for k in [0, 4Z): block position of the lane's VN in an edge block with that shift
JIT_DEV U32 jit_vn_addr(U32 l4, unsigned k, unsigned base) {
  const U32 t = l4 + k;
  return u_min(t, t - (unsigned)JIT_Z4) + base;
}

// l4 + k as a value the compiler treats as computed HERE: the slots of a row are then this one register plus the offset
// fields of the DS instructions.  (Written as plain l4 + k, every distinct slot address of every item is loop invariant
// and is hoisted into a register of its own - hundreds of them, spilled to scratch.)
// (... and the sum is opaque as well: left visible, k is folded into every slot offset, offsets beyond the 16 bits of the DS
// offset field become one address addition per slot, and the two-slot forms ds_read2st64 / ds_write2st64 are lost.)
JIT_DEV U32 jit_base(U32 l4, unsigned k) { return u_here(u_here(l4) + k); }

// interleaved layout, single-chunk variable-node item: byte position of check copy zc, given zc4 = 4 zc = (l4 + k) mod 512
JIT_DEV U32 jit_vn_addr_b1(U32 l4, unsigned k, unsigned base) {
  const U32 zc4 = (l4 + k) & 511u;
  return u_shl(zc4 & 255u, 1) + u_shl(u_shr(zc4, 8), 2) + base;
}

// ---------------------------------------------------------------------------------------------- any even lifting size
// JIT_GENERAL = 1: the interleaved layout for EVERY even Z and several codewords per workgroup.  A lane owns the lifted
// copies (z, z + H), H = Z / 2, of ONE codeword; lane position p = 64 chunk + lane = g H + z runs over the JIT_G codewords
// of the workgroup's group (JIT_P = G H lanes; the lanes beyond it in the last chunk are switched off around every item:
// JIT_IF / JIT_END).  An edge block has JIT_P slots of 8 bytes, slot p = the pair (copy z, copy z + H) of codeword g's check
// node.  A rotation by s maps the pair {z, z + H} onto the pair {z - s, z - s + H} mod Z - one slot, in this order or
// swapped - so both phases move 8 bytes per lane and instruction as at Z = 128 (which is G = 1, H = 64).
// Where a lane's channel values come from and its marginals go is a function of the lane (k, n, fillers and the pruned
// tail are not multiples of 64): byte offsets computed once per launch, or a sentinel.
#ifndef JIT_GENERAL
#define JIT_GENERAL 0
#endif
#if JIT_GENERAL
#define JIT_OFF_NONE 0xFFFFFFFFu                 // no element: channel LLR 0 (punctured / pruned / padding), nothing to store
#define JIT_OFF_FILL 0xFFFFFFFEu                 // filler bit: logit -llr_max (decoding.py:1455-1463)
// slot of the check-node pair a variable-node lane reads through an edge with shift s (sh = s mod H): bytes from the block start
JIT_DEV U32 jit_vn_addr_g(U32 p, U32 z, unsigned sh, unsigned blk) {
  const U32 t = z - sh;                                  // wraps for z < sh
  const U32 zc = u_min(t, t + (unsigned)JIT_H);          // (z - sh) mod H
  return u_shl((p - z) + zc, 3) + blk;
}
// lanes whose own node z sits in the HIGH half of that slot: (z - s) mod Z >= H
JIT_DEV M64 jit_vn_swap_g(U32 z, unsigned sh, bool hi) { return m_xor_c(u_lt(z, sh), hi); }
// a slot = copy z of two codewords: a rotation moves both alike, the pair is never swapped (the selections fold away)
JIT_DEV M64 jit_no_swap() { return u_lt(jit_bcast_u(1u), 0u); }
// rate recovery (decoding.py:1438-1475, encoding.py:238-244) for VN v of the group's codeword g: byte offset of its element
// in the group's received rows, or a sentinel
JIT_DEV U32 jit_in_off(U32 v, U32 g, M64 act) {
  const M64 info = u_lt(v, (unsigned)JIT_K);
  const U32 u = u_sel(info, v, v - (unsigned)JIT_FILLERS);
  const M64 fill = m_and(m_not(info), u_lt(v, (unsigned)JIT_KLDPC));
  const U32 t = u - (unsigned)(2 * JIT_Z);               // wraps for the punctured first columns
  const M64 ok = u_lt(t, (unsigned)JIT_N);
  U32 o = t;
  if (JIT_M > 1u) {
    const U32 tq = u_div(t, (unsigned)JIT_Q);
    o = tq + (t - tq * (unsigned)JIT_Q) * (unsigned)JIT_M;
  }
  U32 r = u_shl(o + g * (unsigned)JIT_N, 2);
  r = u_sel(ok, r, JIT_OFF_NONE);
  r = u_sel(fill, JIT_OFF_FILL, r);
  return u_sel(act, r, JIT_OFF_NONE);
}
// where VN v's marginal goes (decoding.py:1486-1531): information bits in place, or the codeword in the received order
JIT_DEV U32 jit_out_off(U32 v, U32 g, M64 act, U32 in_off) {
  if (!JIT_RIB) return u_sel(u_lt(in_off, JIT_OFF_FILL), in_off, JIT_OFF_NONE);
  return u_sel(m_and(act, u_lt(v, (unsigned)JIT_K)), u_shl(v + g * (unsigned)JIT_NOUT, 2), JIT_OFF_NONE);
}
// received value of a lane (rem = codewords of this group that exist)
JIT_DEV F32 jit_ld_raw(const float* rows, U32 off, U32 g, int rem, float llr_max) {
  const F32 r = g_ld_m(rows, off, m_and(u_lt(off, JIT_OFF_FILL), u_lt(g, (unsigned)rem)));
  return f_sel(u_eq(off, JIT_OFF_FILL), jit_bcast(-llr_max), r);
}
JIT_DEV void jit_st_out(float* orows, U32 off, U32 g, int rem, F32 x, float llr_max, int hard_out) {
  g_st_m(orows, off, m_and(u_lt(off, JIT_OFF_FILL), u_lt(g, (unsigned)rem)), jit_outval(x, llr_max, hard_out));
}
#endif

// ---------------------------------------------------------------------------------------------- where a message pair lives
// Codes whose messages exceed LDS (Z >= 256 at low rate; round 6): the edge blocks of the LAST base rows live in the workgroup's
// row of a caller-owned workspace (it stays in L2) instead of LDS.  SP = the block is there: the address is a byte offset into
// that row.  A column's edges to those rows are its last ones (rows ascending), so the order of every sum is unchanged.
template <bool SP>
JIT_DEV void msg_ld2(const float* ws, U32 a, unsigned off, F32& x0, F32& x1) {
  if (SP) gm_ld2(ws, a, off, x0, x1);
  else lds_ld2(a, off, x0, x1);
}
template <bool SP>
JIT_DEV void msg_st2(float* ws, U32 a, unsigned off, F32 x0, F32 x1) {
  if (SP) gm_st2(ws, a, off, x0, x1);
  else lds_st2(a, off, x0, x1);
}

// ---------------------------------------------------------------------------------------------- check node row
// a0 = byte address of the lane's slot in the row's first block (chunk 0 of the item); NCH chunks of 64 lifted copies
template <int D, int NCH, bool SP = false>
JIT_DEV void jit_cn_load(F32 (&v)[D][NCH], U32 a0, float* ws = nullptr) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    if (JIT_LAYOUT_B && NCH == 2) msg_ld2<SP>(ws, a0, (unsigned)i * JIT_Z4, v[i][0], v[i][NCH - 1]);
    else {
#pragma unroll
      for (int h = 0; h < NCH; ++h) v[i][h] = lds_ld(a0, (unsigned)i * JIT_Z4 + JIT_CH * h);
    }
  }
}
template <int NCH, bool SP = false>
JIT_DEV void jit_cn_store(U32 a0, unsigned off, const F32 (&c)[NCH], float* ws = nullptr) {
  if (JIT_LAYOUT_B && NCH == 2) msg_st2<SP>(ws, a0, off, c[0], c[NCH - 1]);
  else {
#pragma unroll
    for (int h = 0; h < NCH; ++h) lds_st(a0, off + JIT_CH * h, c[h]);
  }
}

// FUSE: the row's last edge goes to a degree-1 VN of the same lane; its update happens here (lf = its channel LLR).
// oc[h] = byte offset (constant part) of that VN's output element, JIT_NOOUT = not part of the output
#ifndef JIT_CMP_AHEAD
#define JIT_CMP_AHEAD 0
#endif
#ifndef JIT_A1_NOCLIP
#define JIT_A1_NOCLIP 0
#endif
// xo[h] (FUSE): total of the fused degree-1 VN, for the generated code's output store after the last iteration.
// PRUNE: lanes of pm[h] are check nodes the rate matching pruned (decoding.py:1344-1373) - they send 0 on every edge.
template <int D, int NCH, bool FUSE, bool PRUNE = false, bool SP = false>
JIT_DEV void jit_cn_update(F32 (&v)[D][NCH], U32 a0, const F32 (&lf)[NCH], float llr_max, float offset, F32 (&xo)[NCH],
                           const M64 (&pm)[NCH], float* ws = nullptr) {
  if (JIT_ABL & 1) {
#pragma unroll
    for (int i = 0; i < D; ++i) jit_cn_store<NCH, SP>(a0, (unsigned)i * JIT_Z4, v[i], ws);
#pragma unroll
    for (int h = 0; h < NCH; ++h) xo[h] = 0.f;
    return;
  }
  if ((JIT_ABL & 8) && JIT_LAYOUT_B && NCH == 2) {
    // estimate of "one total per variable node" (DESIGN section 7 item 1): the check node also reads a second 8-byte slot
    // per edge and forms clip(total - own c2v) itself (two selections, one packed subtraction, two clips); the variable node
    // stores one pair per node instead of one per edge.  Results are WRONG; the traffic and instruction counts are the plan's.
#pragma unroll
    for (int i = 0; i < D; ++i) {
      F32 x0, x1;
      lds_ld2(u_here(a0) ^ 8u, (unsigned)i * JIT_Z4, x0, x1);
      const M64 sw = u_testbit(f_bits(x0), 0x00800000u);
      const F32 y0 = f_sel(sw, x1, x0), y1 = f_sel(sw, x0, x1);
      F32 e0, e1;
      f_pk_sub(e0, e1, y0, y1, v[i][0], v[i][NCH - 1]);
      v[i][0] = f_med3(e0, -llr_max, llr_max);
      v[i][NCH - 1] = f_med3(e1, -llr_max, llr_max);
    }
  }
  F32 m1[NCH], m2[NCH];
  U32 sx[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) { m1[h] = f_abs(v[0][h]); m2[h] = JIT_INF; sx[h] = 0u; }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      // the two smallest magnitudes, the second with multiplicity.  Edge 0: m1 = |v|, m2 = inf (no operation); edge 1:
      // m2 = max, m1 = min; then m2 = med3(m1, m2, |v|), m1 = min(m1, |v|) - min / max of non-negative values as ONE
      // v_med3 against 0 / inf (fminf / fmaxf would add a canonicalising v_max)
      if (i >= 1) {
        const F32 av = f_abs(v[i][h]);
        m2[h] = (i == 1) ? f_med3(m1[h], av, JIT_INF) : f_med3(m1[h], m2[h], av);
        m1[h] = f_med3(m1[h], av, 0.f);
      }
      if (i & 1) sx[h] = u_xor3(sx[h], f_bits(v[i - 1][h]), f_bits(v[i][h]));
      else if (i == D - 1) sx[h] = sx[h] ^ f_bits(v[i][h]);
    }
  F32 a1[NCH], a2[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    // decoding.py:863: min2 = (m2 - m1) + m1 for a unique minimum; a duplicated minimum has m2 == m1 and the same
    // expression returns m1 (0 + m1), so no selection is needed
    const F32 min_e = (m2[h] - m1[h]) + m1[h];
    if (JIT_OFFSET0) {                               // plain min-sum: clip(m - 0, 0, llr_max) = min(m, llr_max), m >= 0
      // JIT_A1_NOCLIP: every v2c was clipped to +-llr_max by its sender (jit_chan, the variable node's med3, the fused column),
      // so m1 <= llr_max and min(m1, llr_max) is m1 itself; (m2 - m1) + m1 can round ONE ulp above m2 and keeps its clip
      a1[h] = JIT_A1_NOCLIP ? m1[h] : f_med3(m1[h], llr_max, 0.f);
      a2[h] = f_med3(min_e, llr_max, 0.f);
    } else {
      a1[h] = f_med3(m1[h] - offset, 0.f, llr_max);
      a2[h] = f_med3(min_e - offset, 0.f, llr_max);
    }
    sx[h] = sx[h] & 0x80000000u;                     // row sign into both candidates (their sign bits are 0)
    a1[h] = u_float(f_bits(a1[h]) | sx[h]);
    a2[h] = u_float(f_bits(a2[h]) | sx[h]);
  }
#if JIT_CMP_AHEAD
  // the comparisons of JIT_CMP_AHEAD edges are issued before the first selection that reads one of them: a v_cmp writes its lane
  // mask to a scalar register pair and gfx950 wants wait states before a v_cndmask reads it - back to back, the compiler fills
  // them with s_nop (587 of them per iteration in the check-node programs of C2, one per 7 vector instructions)
  M64S eq[D][NCH];
#pragma unroll
  for (int i = 0; i < (D < JIT_CMP_AHEAD ? D : JIT_CMP_AHEAD); ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) eq[i][h] = f_eq_abs(v[i][h], m1[h]);
#endif
#pragma unroll
  for (int i = 0; i < D; ++i) {
    F32 c2v[NCH];
#if JIT_CMP_AHEAD
    if (i + JIT_CMP_AHEAD < D) {
#pragma unroll
      for (int h = 0; h < NCH; ++h) eq[i + JIT_CMP_AHEAD][h] = f_eq_abs(v[i + JIT_CMP_AHEAD][h], m1[h]);
    }
#endif
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
#if JIT_CMP_AHEAD
      const F32 mag = f_sel_m(eq[i][h], a2[h], a1[h]);
#else
      const F32 mag = f_sel_eq(f_abs(v[i][h]), m1[h], a2[h], a1[h]);
#endif
      c2v[h] = u_float(u_xor_and(f_bits(mag), f_bits(v[i][h]), 0x80000000u));   // mag ^ (v & msb)
      if (PRUNE) c2v[h] = f_sel(pm[h], jit_bcast(0.f), c2v[h]);
      if (FUSE && i == D - 1) {
        const F32 x = c2v[h] + lf[h];                // (0 + c2v) + llr
        xo[h] = x;
        c2v[h] = f_med3(x - c2v[h], -llr_max, llr_max);   // the slot now holds the next v2c
      }
    }
    jit_cn_store<NCH, SP>(a0, (unsigned)i * JIT_Z4, c2v, ws);
  }
}

// ---------------------------------------------------------------------------------------------- boxplus-phi check node row
// JIT_RULE_PHI = 1: the kernel is generated for cn_update_phi (decoding.py:1045-1166) on the DEFINED exp / log of round 5
// (csrc/bp_math.h phi_exp2_f32 / phi_log2_f32 = oracle/ldpc_bp.c phi_expf / phi_logf, operation for operation).  Packed
// operations pair the two CHUNKS of an edge (the generic engines pair two edges of one chunk - the arithmetic per value is
// the same, so are the bits).  The 64-entry table of the logarithm sits in LDS at JIT_PHI_TAB (8 bytes per entry).
#ifndef JIT_RULE_PHI
#define JIT_RULE_PHI 0
#endif
#if JIT_RULE_PHI
#ifndef JIT_PHI_LEAN
#define JIT_PHI_LEAN 0
#endif
static JIT_TABLE float jit_phi_tab[64][2] = {
JIT_PHI_TAB_ROWS
};
JIT_DEV void jit_phi_exp2(F32& e0, F32& e1, F32 x0, F32 x1) {
  F32 t0, t1, m0, m1, r0, r1, z0, z1, y0, y1;
  f_pk_fma(t0, t1, x0, x1, 1.44269504088896341f, 12582912.0f, 12582912.0f);
  f_pk_addc(m0, m1, t0, t1, -12582912.0f);
  f_pk_fma(r0, r1, m0, m1, -0.693359375f, x0, x1);
  f_pk_fma(r0, r1, m0, m1, 2.12194440e-4f, r0, r1);
  f_pk_mul(z0, z1, r0, r1, r0, r1);
  y0 = 1.9875691500E-4f; y1 = 1.9875691500E-4f;
  f_pk_fma(y0, y1, y0, y1, r0, r1, 1.3981999507E-3f, 1.3981999507E-3f);
  f_pk_fma(y0, y1, y0, y1, r0, r1, 8.3334519073E-3f, 8.3334519073E-3f);
  f_pk_fma(y0, y1, y0, y1, r0, r1, 4.1665795894E-2f, 4.1665795894E-2f);
  f_pk_fma(y0, y1, y0, y1, r0, r1, 1.6666665459E-1f, 1.6666665459E-1f);
  f_pk_fma(y0, y1, y0, y1, r0, r1, 5.0000001201E-1f, 5.0000001201E-1f);
  f_pk_fma(y0, y1, y0, y1, z0, z1, r0, r1);
  f_pk_addc(y0, y1, y0, y1, 1.0f);
  e0 = u_float(f_bits(y0) + u_shl(f_bits(t0), 23));          // 2^m through the exponent field, m = low bits of t
  e1 = u_float(f_bits(y1) + u_shl(f_bits(t1), 23));
}
// JIT_PHI_TAB32 (round 6): the table as two 256-byte planes (factor, then constant) read with FOUR 4-byte loads per pair: each
// value lands in the register its packed operation wants.  The 8-byte entry (factor, constant) of ONE lookup came back in two
// neighbouring registers and the pair (factor of lane value 0, factor of lane value 1) took three v_mov_b32 per logarithm pair to
// assemble.  LDS has the room (0.26 busy); the plain 4-byte loads may meet bank conflicts (64 entries on 32 banks).
#ifndef JIT_PHI_TAB32
#define JIT_PHI_TAB32 0
#endif
JIT_DEV void jit_phi_log2(F32& l0, F32& l1, F32 x0, F32 x1) {      // normal positive arguments
  const U32 b0 = f_bits(x0), b1 = f_bits(x1);
  // e + 1 as a float.  JIT_PHI_LEAN: the biased exponent under the mantissa of 2^23 is the float 2^23 + (e + 127); minus
  // (2^23 + 126) - exact, small integers - gives the same e + 1 with three full-rate operations (v_lshrrev, v_or, v_sub_f32:
  // ~7.4 cycles per wave) instead of v_frexp_exp_i32_f32 + v_cvt_f32_i32 (~8.7)
  const F32 ef0 = JIT_PHI_LEAN ? u_float(u_shr(b0, 23) | 0x4B000000u) - 8388734.0f : f_frexp_exp(x0);
  const F32 ef1 = JIT_PHI_LEAN ? u_float(u_shr(b1, 23) | 0x4B000000u) - 8388734.0f : f_frexp_exp(x1);
  F32 i0, c0, i1, c1, r0, r1, t0, t1;
  if (JIT_PHI_TAB32) {
    const U32 a0 = u_and_or(u_shr(b0, 15), 0xFCu, JIT_PHI_TAB), a1 = u_and_or(u_shr(b1, 15), 0xFCu, JIT_PHI_TAB);
    i0 = lds_ld_single(a0, 0u); i1 = lds_ld_single(a1, 0u);
    c0 = lds_ld_single(a0, 256u); c1 = lds_ld_single(a1, 256u);
  } else {
    lds_ld2(u_and_or(u_shr(b0, 14), 0x1F8u, JIT_PHI_TAB), 0u, i0, c0);
    lds_ld2(u_and_or(u_shr(b1, 14), 0x1F8u, JIT_PHI_TAB), 0u, i1, c1);
  }
  const F32 p0 = u_float(u_and_or(b0, 0x007FFFFFu, 0x3F800000u)), p1 = u_float(u_and_or(b1, 0x007FFFFFu, 0x3F800000u));
  f_pk_fma(r0, r1, p0, p1, i0, i1, -1.0f, -1.0f);
  f_pk_fma(t0, t1, r0, r1, 0.333333343f, -0.5f, -0.5f);
  f_pk_fma(t0, t1, t0, t1, r0, r1, 1.0f, 1.0f);
  f_pk_fma(t0, t1, t0, t1, r0, r1, c0, c1);
  f_pk_fma(l0, l1, ef0, ef1, 0.69314718055994530942f, t0, t1);
}
// JIT_PHI_LEAN (round 6): the clamp as ONE v_med3_f32 that takes |x| as a source modifier (fminf(fmaxf()) compiles to a
// canonicalising v_max_f32 |x|, |x| in front of it; the same value for every non-NaN x), and the sign of a v2c read from its
// sign bit: a v2c is never -0 (a variable node's total starts from +0 and x - c is +0 when x == c; jit_chan adds +0), so the
// bit is exactly "v < 0" - one v_and_b32 instead of v_cmp + v_cndmask.  tests/jit_emu traps if a -0 ever reaches jit_sign_msb.
JIT_DEV U32 jit_sign_msb(F32 v) { return JIT_PHI_LEAN ? u_msb_nonzero(v) : u_msb_if_neg(v); }
JIT_DEV void jit_phi2(F32& y0, F32& y1, F32 x0, F32 x1) {
  F32 e0, e1, a0, a1, b0, b1, p0, p1, q0, q1;
  if (JIT_PHI_LEAN) {
    x0 = f_med3(x0, 8.5e-8f, 16.635532f);
    x1 = f_med3(x1, 8.5e-8f, 16.635532f);
  } else {
    x0 = f_clamp(x0, 8.5e-8f, 16.635532f);
    x1 = f_clamp(x1, 8.5e-8f, 16.635532f);
  }
  jit_phi_exp2(e0, e1, x0, x1);
  f_pk_addc(a0, a1, e0, e1, 1.0f);
  f_pk_addc(b0, b1, e0, e1, -1.0f);
  jit_phi_log2(p0, p1, a0, a1);
  jit_phi_log2(q0, q1, b0, b1);
  f_pk_sub(y0, y1, p0, p1, q0, q1);
  y0 = f_sel_eq(x0, 16.635532f, 0.f, y0);
  y1 = f_sel_eq(x1, 16.635532f, 0.f, y1);
}
// the table of the logarithm into LDS (wave 0, once per launch; the first workgroup barrier orders it before its first use)
JIT_DEV void jit_phi_stage(U32 l4) {
  F32 a, b;
  jit_tab_lane(a, b, jit_phi_tab);
  if (JIT_PHI_TAB32) { lds_st(l4, JIT_PHI_TAB, a); lds_st(l4, JIT_PHI_TAB + 256u, b); }
  else lds_st2(l4 + l4, JIT_PHI_TAB, a, b);
}
// the check-node update of bp_math.h's cn_update_col (boxplus-phi branch).  NCH = 2: the two chunks of an edge share every
// packed operation; NCH = 1 (rows of high degree, one chunk per item: registers): two EDGES share them.
template <int D, int NCH, bool FUSE, bool PRUNE = false>
JIT_DEV void jit_cn_update_phi(F32 (&v)[D][NCH], U32 a0, const F32 (&lf)[NCH], float llr_max, F32 (&xo)[NCH],
                               const M64 (&pm)[NCH]) {
  U32 sg[D][NCH], node[NCH];
  F32 sum[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) { node[h] = 0u; sum[h] = 0.f; }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      sg[i][h] = jit_sign_msb(v[i][h]);                      // sign_nz(v) = -1 <=> v < 0 (a -0 counts as +)
      node[h] = node[h] ^ sg[i][h];
    }
  if (NCH == 2) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      F32 p0, p1;
      jit_phi2(p0, p1, f_abs(v[i][0]), f_abs(v[i][NCH - 1]));
      v[i][0] = p0; v[i][NCH - 1] = p1;
      sum[0] = sum[0] + p0; sum[NCH - 1] = sum[NCH - 1] + p1;
    }
  } else {
#pragma unroll
    for (int i = 0; i < D; i += 2) {
      const int i2 = (i + 1 < D) ? i + 1 : i;
      F32 p0, p1;
      jit_phi2(p0, p1, f_abs(v[i][0]), f_abs(v[i2][0]));
      v[i][0] = p0;
      sum[0] = sum[0] + p0;
      if (i + 1 < D) { v[i2][0] = p1; sum[0] = sum[0] + p1; }
    }
  }
  F32 q[D][NCH];
  if (NCH == 2) {
#pragma unroll
    for (int i = 0; i < D; ++i) jit_phi2(q[i][0], q[i][NCH - 1], f_neg(v[i][0]) + sum[0], f_neg(v[i][NCH - 1]) + sum[NCH - 1]);
  } else {
#pragma unroll
    for (int i = 0; i < D; i += 2) {
      const int i2 = (i + 1 < D) ? i + 1 : i;
      F32 q1;
      jit_phi2(q[i][0], q1, f_neg(v[i][0]) + sum[0], f_neg(v[i2][0]) + sum[0]);
      if (i + 1 < D) q[i2][0] = q1;
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    F32 c2v[NCH];
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      c2v[h] = u_float(f_bits(f_min(q[i][h], llr_max)) ^ (sg[i][h] ^ node[h]));
      if (PRUNE) c2v[h] = f_sel(pm[h], jit_bcast(0.f), c2v[h]);
      if (FUSE && i == D - 1) {
        const F32 x = c2v[h] + lf[h];
        xo[h] = x;
        c2v[h] = f_med3(x - c2v[h], -llr_max, llr_max);
      }
    }
    jit_cn_store<NCH>(a0, (unsigned)i * JIT_Z4, c2v);
  }
}
// The same update with the loops over the row's edges ROLLED (round 6): D is a wave-uniform run-time value, each pass has ONE
// phi body, the phi(|v2c|) of the first pass wait in the row's own message slots (read back by the second pass), the signs in
// one bit per edge.  Unrolled, the 16 wave programs of C2 were 437 KB of code against a 64 KB instruction cache with 1076
// spilled registers (profiles/r05l); rolled they are a tenth.  Same operations on the same values in the same order: same bits.
template <bool FUSE, bool PRUNE>
JIT_DEV void jit_cn_phi_rolled(U32 a0, int D, const F32 (&lf)[2], float llr_max, F32 (&xo)[2], const M64 (&pm)[2]) {
  // the sign of edge i's v2c waits, with phi(|v2c|), in the row's own slot: phi >= 0, so its sign bit is free (the second pass
  // takes |.| as a source modifier).  Until round 6 the signs were collected in one bit per edge of a register (v_and,
  // v_lshrrev by a scalar count, v_or in the first pass; two shifts and a v_xor in the second).  node: their parity in the msb.
  U32 node0 = 0u, node1 = 0u;
  F32 sum0 = 0.f, sum1 = 0.f;
  U32 a = a0;
#pragma unroll 1
  for (int i = 0; i < D; ++i) {
    F32 v0, v1, p0, p1;
    lds_ld2(a, 0u, v0, v1);
    const U32 s0 = jit_sign_msb(v0), s1 = jit_sign_msb(v1);
    node0 = node0 ^ s0; node1 = node1 ^ s1;
    jit_phi2(p0, p1, f_abs(v0), f_abs(v1));
    sum0 = sum0 + p0; sum1 = sum1 + p1;
    lds_st2(a, 0u, u_float(f_bits(p0) | s0), u_float(f_bits(p1) | s1));
    a = a + (unsigned)JIT_Z4;
  }
  a = a0;
#pragma unroll 1
  for (int i = 0; i < D; ++i) {
    F32 p0, p1, q0, q1;
    lds_ld2(a, 0u, p0, p1);
    jit_phi2(q0, q1, sum0 - f_abs(p0), sum1 - f_abs(p1));                       // = (-phi) + sum, decoding.py:1147-1152
    // min(q, llr_max): the defined phi never exceeds phi(8.5e-8) = 16.635532 (all 231.6 M floats of its domain evaluated,
    // tests/test_oracle_pins.py), so the clip can only act below that bound - a branch on the launch parameter
    if (llr_max < 16.635532f) { JIT_KEEP_BRANCH(); q0 = f_min(q0, llr_max); q1 = f_min(q1, llr_max); }
    F32 c0 = u_float(u_xor_and(f_bits(q0) ^ node0, f_bits(p0), 0x80000000u));
    F32 c1 = u_float(u_xor_and(f_bits(q1) ^ node1, f_bits(p1), 0x80000000u));
    if (PRUNE) { c0 = f_sel(pm[0], jit_bcast(0.f), c0); c1 = f_sel(pm[1], jit_bcast(0.f), c1); }
    if (FUSE && i == D - 1) {
      // a BRANCH on the (wave-uniform) trip count: converted to selections, this block ran on every trip and its two
      // v_cndmask_b32 read a lane mask that a SCALAR instruction wrote (s_cselect_b64 vcc) - ~24 cycles each on gfx950
      // (profiles/r06w_valu_rate2.txt) against ~2.5 behind a vector comparison
      JIT_KEEP_BRANCH();
      const F32 x0 = c0 + lf[0], x1 = c1 + lf[1];
      xo[0] = x0; xo[1] = x1;
      c0 = f_med3(x0 - c0, -llr_max, llr_max);
      c1 = f_med3(x1 - c1, -llr_max, llr_max);
    }
    lds_st2(a, 0u, c0, c1);
    a = a + (unsigned)JIT_Z4;
  }
}
#endif

// v2c of iteration 0 for the fused degree-1 column of a row: its channel LLR
template <int D, int NCH, bool SP = false>
JIT_DEV void jit_cn_init_fused(U32 a0, const F32 (&lf)[NCH], float* ws = nullptr) {
  jit_cn_store<NCH, SP>(a0, (unsigned)(D - 1) * JIT_Z4, lf, ws);
}

// ---------------------------------------------------------------------------------------------- variable node column
// a[i][h] = byte address of the lane's slot in the block of the column's i-th edge (rows ascending), chunk h: registers
// that live for the whole launch, or - second chunk at Z = 128 - the first chunk's position with bit 8 flipped,
// ((t + 256) mod 512) = (t mod 512) ^ 256, computed per item (the generated code writes the array out)
template <int D, int NCH>
JIT_DEV void jit_vn_load(F32 (&c)[D][NCH], const U32 (&a)[D][NCH]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) c[i][h] = lds_ld(a[i][h], 0u);
}

template <int D, int NCH>
JIT_DEV void jit_vn_init(const U32 (&a)[D][NCH], const F32 (&l)[NCH]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int h = 0; h < NCH; ++h) lds_st(a[i][h], 0u, l[h]);
}

// ---- interleaved layout, pair items: a[i] = byte address of slot (lane - shift_i) mod 64 in the block of edge i, sw[i] = the
// lanes whose node `lane` (chunk 0) sits in the HIGH half of that slot (then node lane + 64 sits in the low half)
// SPM: bit i = the block of edge i lives in the workspace row (see msg_ld2)
#define JIT_VN_EDGE(i, CALL_SP, CALL_LDS) do { if ((SPM >> (i)) & 1u) { CALL_SP; } else { CALL_LDS; } } while (0)
template <int D, unsigned SPM = 0u>
JIT_DEV void jit_vnb_load(F32 (&c)[D][2], const U32 (&a)[D], float* ws = nullptr) {
#pragma unroll
  for (int i = 0; i < D; ++i) JIT_VN_EDGE(i, gm_ld2(ws, a[i], 0u, c[i][0], c[i][1]), lds_ld2(a[i], 0u, c[i][0], c[i][1]));
}
template <int D, unsigned SPM = 0u>
JIT_DEV void jit_vnb_init(const U32 (&a)[D], const M64 (&sw)[D], const F32 (&l)[2], float* ws = nullptr) {
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const F32 s0 = f_sel(sw[i], l[1], l[0]), s1 = f_sel(sw[i], l[0], l[1]);
    JIT_VN_EDGE(i, gm_st2(ws, a[i], 0u, s0, s1), lds_st2(a[i], 0u, s0, s1));
  }
}
template <int D, unsigned SPM = 0u>
JIT_DEV void jit_vnb_update(F32 (&c)[D][2], const U32 (&a)[D], const M64 (&sw)[D], const F32 (&l)[2], float llr_max, F32 (&xo)[2],
                            float* ws = nullptr) {
  if (JIT_ABL & 2) {
#pragma unroll
    for (int i = 0; i < D; ++i) JIT_VN_EDGE(i, gm_st2(ws, a[i], 0u, c[i][0], c[i][1]), lds_st2(a[i], 0u, c[i][0], c[i][1]));
    xo[0] = 0.f; xo[1] = 0.f;
    return;
  }
  F32 x0 = 0.f, x1 = 0.f;
  F32 v0[D], v1[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {                          // slot order -> node order
    v0[i] = f_sel(sw[i], c[i][1], c[i][0]);
    v1[i] = f_sel(sw[i], c[i][0], c[i][1]);
    f_pk_add(x0, x1, v0[i], v1[i]);
  }
  f_pk_add(x0, x1, l[0], l[1]);
  if (JIT_ABL & 8) {
    lds_st2(a[0], 0u, x0, x1);
    xo[0] = x0; xo[1] = x1;
    return;
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    F32 e0, e1;
    f_pk_sub(e0, e1, x0, x1, v0[i], v1[i]);
    e0 = f_med3(e0, -llr_max, llr_max);
    e1 = f_med3(e1, -llr_max, llr_max);
    const F32 o0 = f_sel(sw[i], e1, e0), o1 = f_sel(sw[i], e0, e1);    // node order -> slot order
    JIT_VN_EDGE(i, gm_st2(ws, a[i], 0u, o0, o1), lds_st2(a[i], 0u, o0, o1));
  }
  xo[0] = x0; xo[1] = x1;
}

// ---- the same pair items with 4-byte STORES (round 6, JIT_VN_ST32): a[i] = the slot's byte address + 4 for the lanes whose
// pair is swapped = where the lane's node `lane` lives; node lane + 64 lives at a[i] ^ 4.  The two results of an edge go out in
// NODE order through those two addresses (ds_write_b32 x 2: 8 LDS-pipeline cycles against 6, two-way bank conflicts that a 4-byte
// store hides) instead of two selections into slot order and one ds_write_b64: a v_cndmask_b32 with its lane mask in an SGPR pair
// issues in ~4.5 cycles per wave, the v_xor_b32 that forms the second address in ~2.5 (profiles/r06w_valu_rate2.txt).  Loads keep the
// 8-byte form (a[i] & ~4) and their two selections: 4-byte loads would meet two-way conflicts that loads do pay for.
template <int D>
JIT_DEV void jit_vnc_load(F32 (&c)[D][2], const U32 (&a)[D]) {
#pragma unroll
  for (int i = 0; i < D; ++i) lds_ld2(u_andn4_here(a[i]), 0u, c[i][0], c[i][1]);
}
template <int D>
JIT_DEV void jit_vnc_init(const U32 (&a)[D], const F32 (&l)[2]) {
#pragma unroll
  for (int i = 0; i < D; ++i) { lds_st(a[i], 0u, l[0]); lds_st(u_xor4_here(a[i]), 0u, l[1]); }
}
template <int D>
JIT_DEV void jit_vnc_update(F32 (&c)[D][2], const U32 (&a)[D], const M64 (&sw)[D], const F32 (&l)[2], float llr_max, F32 (&xo)[2]) {
  F32 x0 = 0.f, x1 = 0.f;
  F32 v0[D], v1[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {                          // slot order -> node order
    v0[i] = f_sel(sw[i], c[i][1], c[i][0]);
    v1[i] = f_sel(sw[i], c[i][0], c[i][1]);
    f_pk_add(x0, x1, v0[i], v1[i]);
  }
  f_pk_add(x0, x1, l[0], l[1]);
#pragma unroll
  for (int i = 0; i < D; ++i) {
    F32 e0, e1;
    f_pk_sub(e0, e1, x0, x1, v0[i], v1[i]);
    lds_st(a[i], 0u, f_med3(e0, -llr_max, llr_max));
    lds_st(u_xor4_here(a[i]), 0u, f_med3(e1, -llr_max, llr_max));
  }
  xo[0] = x0; xo[1] = x1;
}

template <int D, int NCH>
JIT_DEV void jit_vn_update(F32 (&c)[D][NCH], const U32 (&a)[D][NCH], const F32 (&l)[NCH], float llr_max, F32 (&x)[NCH]) {
  if (JIT_ABL & 2) {
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int h = 0; h < NCH; ++h) lds_st(a[i][h], 0u, c[i][h]);
#pragma unroll
    for (int h = 0; h < NCH; ++h) x[h] = 0.f;
    return;
  }
  if (NCH == 2) {
    // both chunks of an edge in one packed-fp32 operation: two IEEE additions, the results of two scalar ones
    F32 x0 = 0.f, x1 = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) f_pk_add(x0, x1, c[i][0], c[i][NCH - 1]);
    f_pk_add(x0, x1, l[0], l[NCH - 1]);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      F32 e0, e1;
      f_pk_sub(e0, e1, x0, x1, c[i][0], c[i][NCH - 1]);
      lds_st(a[i][0], 0u, f_med3(e0, -llr_max, llr_max));
      lds_st(a[i][NCH - 1], 0u, f_med3(e1, -llr_max, llr_max));
    }
    x[0] = x0; x[NCH - 1] = x1;
  } else {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      x[h] = 0.f;
#pragma unroll
      for (int i = 0; i < D; ++i) x[h] = x[h] + c[i][h];
      x[h] = x[h] + l[h];
#pragma unroll
      for (int i = 0; i < D; ++i) lds_st(a[i][h], 0u, f_med3(x[h] - c[i][h], -llr_max, llr_max));
    }
  }
}
