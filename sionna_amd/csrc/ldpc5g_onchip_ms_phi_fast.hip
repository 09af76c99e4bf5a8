// The explicit-message engine (ldpc5g_onchip_ms.hip / .inc) instantiated for boxplus-phi on the hardware transcendentals
// (SAMD_CN_BOXPLUS_PHI_FAST: v_exp_f32 / v_log_f32 instead of the defined exp / log of bp_math.h).
#include "ldpc5g_onchip_ms.inc"

namespace samd {

int launch_onchip_ms_phi_fast(const samd_ldpc5g* h, const float* llr, float* out, int batch, int num_iter, int cn_mode,
                              float llr_max, float offset, int hard_out, int return_infobits, void* workspace,
                              size_t workspace_bytes, hipStream_t st) {
  return launch_onchip_ms_mode<SAMD_CN_BOXPLUS_PHI_FAST>(h, llr, out, batch, num_iter, cn_mode, llr_max, offset, hard_out,
                                                         return_infobits, workspace, workspace_bytes, st);
}

}  // namespace samd
