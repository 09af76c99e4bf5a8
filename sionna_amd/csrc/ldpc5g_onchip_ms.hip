// On-chip flooding min-sum / offset-min-sum decoder with EXPLICIT messages: one float per edge in LDS.
//
// Replaces LDPC5GDecoder.call = rate recovery + LDPCBPDecoder._bp_iter x num_iter +
// cn_update_(offset_)minsum + vn_update_sum + output mapping (reference
// src/sionna/phy/fec/ldpc/decoding.py:1427-1536, 416-524, 681-953) for codes whose E 4-byte messages
// fit in 160 KB (C2: 316 Z edges = 158 KB, channel LLRs in an L2 workspace row).
//
// Why a third min-sum engine: ldpc5g_onchip.hip keeps the check-node state COMPRESSED (M1, M2, index,
// signs - 12 B per check node) because messages + LLRs + totals of C2 do not fit in LDS together; the price
// is that every edge's c2v is rebuilt from the compressed state twice per iteration (27 VALU operations
// per edge, the kernel is 85 % VALU busy).  With the channel LLRs moved to L2 the 158 KB of messages fit
// by themselves, and the per-edge work drops to
//   CN: read v2c, m2 = med3(m1, m2, |v|), m1 = min(m1, |v|), sign ^= v;  then  mag = (|v| == m1) ? a2 : a1,
//       c2v = ((v ^ sign) & msb) | mag, write                                   (~9 VALU)
//   VN: read c2v, x += c;  then  v2c = med3(x - c, -L, L), write                (~5 VALU)
// Same arithmetic as the compressed engine and the oracle (bit-exact): the second minimum is tracked with
// multiplicity, a unique minimum gives min_e = (m2 - m1) + m1 (decoding.py:863), ties give min_e = m1;
// v2c is never -0 (the channel LLR is never -0), so the raw sign bit equals (v2c < 0).
//
// Layout, work split and tables are those of ldpc5g_onchip_bp.hip (edge blocks indexed by the check node's
// lifted copy; (row, chunk) / (column, chunk) items balanced over the waves on the host); every row /
// column runs an instantiation for its exact degree, table entries are wave-uniform scalars that feed
// the VALU operations directly (no scalar unpacking per edge).
#define SAMD_MS_MAIN_TU
#include "ldpc5g_onchip_ms.inc"
#include "ldpc5g_jit.h"

namespace samd {

int launch_onchip_ms_phi(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                         float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                         size_t workspace_bytes, hipStream_t st);     // ldpc5g_onchip_ms_phi.hip
int launch_onchip_ms_phi_fast(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                              float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                              size_t workspace_bytes, hipStream_t st);     // ldpc5g_onchip_ms_phi_fast.hip
int launch_onchip_ms(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                     float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                     size_t workspace_bytes, hipStream_t st) {
  if (cn_mode == SAMD_CN_BOXPLUS_PHI) {
    // the kernel generated for this code (ldpc5g_jit.cpp) when there is one - the same defined phi, the same bits
    const int rcj = launch_onchip_jit(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out, return_infobits, workspace, workspace_bytes, st);
    if (rcj != SAMD_ERR_UNSUPPORTED) return rcj;
  }
  if (cn_mode == SAMD_CN_BOXPLUS_PHI)
    return launch_onchip_ms_phi(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out, return_infobits, workspace,
                                workspace_bytes, st);
  if (cn_mode == SAMD_CN_BOXPLUS_PHI_FAST)
    return launch_onchip_ms_phi_fast(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out, return_infobits, workspace,
                                     workspace_bytes, st);
  if (cn_mode == SAMD_CN_BOXPLUS) {
    // the tanh rule stays on the first boxplus kernel (ldpc5g_onchip_bp.hip): this engine gains it 5 %, and its 12
    // kernels with the inlined tanh / atanh cost three minutes of compile time
    set_error("boxplus (tanh) runs on ldpc5g_decode_bp_kernel");
    return SAMD_ERR_UNSUPPORTED;
  }
  {
    // the kernel generated for this code (ldpc5g_jit.cpp) when there is one; else the lists below
    const int rc = launch_onchip_jit(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out, return_infobits, workspace, workspace_bytes, st);
    if (rc != SAMD_ERR_UNSUPPORTED) return rc;
  }
  return launch_onchip_ms_mode<SAMD_CN_MINSUM>(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out,
                                               return_infobits, workspace, workspace_bytes, st);
}

}  // namespace samd
