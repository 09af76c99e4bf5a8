"""ctypes binding of the C-ABI (include/sionna_amd.h) and device/stream plumbing.

PyTorch is used only as the carrier of device memory and streams: every call passes raw
``data_ptr()`` addresses and the current HIP stream handle to ``libsionna_amd.so``.
There is NO CPU fallback: if the library is missing or no GPU is visible, compute calls
raise immediately.
"""
import ctypes as C
import os
import re

import torch  # imported first so that libamdhip64 is resolved once, process-wide

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAMD_LIB") or os.path.join(_HERE, "lib", "libsionna_amd.so")   # SAMD_LIB: development builds
HEADER_PATH = os.path.join(_HERE, "..", "include", "sionna_amd.h")

OK, ERR_INVALID, ERR_HIP, ERR_UNSUPPORTED, ERR_WORKSPACE = 0, -1, -2, -3, -4
# "boxplus-phi-fast": the same rule on the GPU's transcendental unit (SAMD_CN_BOXPLUS_PHI_FAST, include/sionna_amd.h) -
# an addition to the reference's rule names; "boxplus-phi" evaluates phi on the defined float32 exp / log and is
# bit-identical to the specification in oracle/ldpc_bp.c.  "fast" runs the DEFINED form on the float64 decoder, on the
# callback (torch) engine of phy/fec/ldpc/custom.py and on the first on-chip boxplus engine (csrc/ldpc5g_onchip_bp.hip): which arithmetic "fast" means
# depends on the engine, "boxplus-phi" means the same bits everywhere.
CN_MODES = {"boxplus": 0, "boxplus-phi": 1, "minsum": 2, "min": 2, "offset-minsum": 3, "boxplus-phi-fast": 4}

_lib = None

_vp, _i32, _i64, _u64, _f32, _sz, _f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_size_t, C.c_double
_SIGNATURES = {
    "samd_last_error": (C.c_char_p, []),
    "samd_version": (_i32, []),
    "samd_device_count": (_i32, []),
    "samd_ldpc_graph_create": (_i32, [_vp, _vp, _i32, _i32, _i32, C.POINTER(_vp)]),
    "samd_ldpc_graph_destroy": (None, [_vp]),
    "samd_ldpc_bp_workspace_bytes": (_sz, [_vp, _i32]),
    "samd_ldpc_bp_decode_f32": (_i32, [_vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32,
                                       _i32, _vp, _sz, _vp]),
    "samd_ldpc_schedule_create": (_i32, [_vp, _vp, _i32, _i32, C.POINTER(_vp)]),
    "samd_ldpc_schedule_destroy": (None, [_vp]),
    "samd_ldpc_bp_decode_scheduled_f32": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _f32,
                                                 _f32, _i32, _vp, _sz, _vp]),
    "samd_ldpc_bp_workspace_bytes_f64": (_sz, [_vp, _i32]),
    "samd_ldpc_bp_decode_f64": (_i32, [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, C.c_double,
                                       C.c_double, _i32, _vp, _sz, _vp]),
    "samd_ldpc5g_rate_recover_f64": (_i32, [_vp, _vp, _vp, _i32, C.c_double, _vp]),
    "samd_ldpc5g_extract_codeword_f64": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "samd_lmmse_equalizer_c128": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "samd_qam_demap_f64": (_i32, [_vp, _vp, _i64, _vp, _i32, _i64, _vp, _i64, _i32, _i32, _vp, _vp]),
    "samd_symbol_demap_f64": (_i32, [_vp, _vp, _i64, _vp, _i32, _i64, _vp, _i64, _i32, _vp, _vp, _vp]),
    "samd_symbol_logits2llrs_f64": (_i32, [_vp, _i32, _i64, _vp, _i64, _i32, _i32, _vp, _vp]),
    "samd_llrs2symbol_logits_f64": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    "samd_symbol_logits2moments_c128": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp, _vp]),
    "samd_pam2qam_logits_f64": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp]),
    "samd_ml_detect_f64": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_ep_f64": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f64, _f64, _f64, _i32, _vp, _vp]),
    "samd_inv_cholesky_c64": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "samd_inv_cholesky_c128": (_i32, [_vp, _i64, _i32, _vp, _vp]),
    "samd_matrix_pinv_c64": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "samd_matrix_pinv_c128": (_i32, [_vp, _i64, _i32, _i32, _vp, _vp]),
    "samd_whiten_channel_c64": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "samd_whiten_channel_c128": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "samd_lmmse_matrix_c64": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "samd_lmmse_matrix_c128": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "samd_cir_to_time_c128": (_i32, [_f64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_apply_time_channel_c128": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_scramble_f64": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "samd_awgn_c128": (_i32, [_vp, _vp, _i64, _u64, _u64, _i64, _vp, _vp]),
    "samd_rg_map_c128": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_tdl_cir_c128": (_i32, [_u64, _u64, _i32, _i32, _i32, _i32, _i32, _i32, _f64, _vp, _f64, _f64, _i32, _f64, _f64, _vp, _vp]),
    "samd_cir_to_ofdm_c128": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_apply_ofdm_channel_c128": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_ls_gather_scale_c128": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_ofdm_modulate_c128": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "samd_ofdm_demodulate_c128": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "samd_ldpc5g_create": (_i32, [_i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.POINTER(_vp)]),
    "samd_ldpc5g_destroy": (None, [_vp]),
    "samd_ldpc5g_encode_f32": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "samd_ldpc5g_rate_recover_f32": (_i32, [_vp, _vp, _vp, _i32, _f32, _vp]),
    "samd_ldpc5g_extract_codeword_f32": (_i32, [_vp, _vp, _vp, _i32, _vp]),
    "samd_ldpc5g_decode_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "samd_ldpc5g_decode_engine": (_i32, [_vp, _i32]),
    "samd_ldpc5g_decode_layered_supported": (_i32, [_vp, _i32]),
    "samd_ldpc5g_decode_layered_workspace_bytes": (_sz, [_vp, _i32]),
    "samd_ldpc5g_decode_layered_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _i32, _vp, _sz, _vp]),
    "samd_ldpc5g_jit_supported": (_i32, [_vp]),
    "samd_ldpc5g_jit_prepare": (_i32, [_vp, _i32, _i32]),
    "samd_ldpc5g_jit_source": (C.c_long, [_vp, _i32, _i32, _i32, _vp, _sz]),
    "samd_ldpc5g_jit_code": (C.c_long, [_vp, _i32, _i32, _vp, _sz]),
    "samd_ldpc5g_jit_launches": (C.c_long, [_vp]),
    "samd_ldpc5g_state_layout": (_i32, [_vp, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "samd_ldpc5g_state_map": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "samd_ldpc5g_state_convert_f32": (_i32, [_vp, _i32, _i32, C.c_long, _vp, _vp, _i32, _vp]),
    "samd_ldpc5g_decode_state_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _i32, _vp]),
    "samd_ldpc5g_jit_cache_stats": (_i32, [_vp]),
    "samd_ldpc5g_decode_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _f32, _f32, _i32, _i32, _vp, _sz, _vp]),
    "samd_qam_map_c64": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp]),
    "samd_qam_demap_f32": (_i32, [_vp, _vp, _i64, _vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    "samd_qam_demap_prior_f32": (_i32, [_vp, _vp, _i64, _vp, _i32, _i64, _vp, _i64, _i32, _i32, _vp, _vp]),
    "samd_symbol_demap_f32": (_i32, [_vp, _vp, _i64, _vp, _i32, _i64, _vp, _i64, _i32, _vp, _vp, _vp]),
    "samd_symbol_logits2llrs_f32": (_i32, [_vp, _i32, _i64, _vp, _i64, _i32, _i32, _vp, _vp]),
    "samd_llrs2symbol_logits_f32": (_i32, [_vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    "samd_symbol_logits2moments_c64": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp, _vp]),
    "samd_pam2qam_logits_f32": (_i32, [_vp, _vp, _i32, _i64, _vp, _vp]),
    "samd_square_qam_demap_f32": (_i32, [_vp, _vp, _i64, _vp, _i32, _i64, _i32, _i32, _vp, _vp]),
    "samd_binary_source_f32": (_i32, [_u64, _u64, _i64, _vp, _vp]),
    "samd_awgn_c64": (_i32, [_vp, _vp, _i64, _u64, _u64, _i64, _vp, _vp]),
    "samd_rg_map_c64": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_gather3": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_tdl_cir_c64": (_i32, [_u64, _u64, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _f32, _f32, _i32, _f32,
                                _f32, _vp, _vp]),
    "samd_spatial_corr_c64": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "samd_spatial_corr_c128": (_i32, [_vp, _vp, _i32, _i32, _i32, _i64, _vp, _vp]),
    "samd_cdl_workspace_bytes": (_sz, [_i32, _i32]),
    "samd_cdl_cir_c64": (_i32, [_u64, _u64, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _f32, _f32, _f32, _f32, _vp, _sz, _vp, _vp]),
    "samd_cdl_cir_c128": (_i32, [_u64, _u64, _i32, _i32, _i32, _i32, _i32, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _f64, _f64, _f64, _f64, _vp, _sz, _vp, _vp]),
    "samd_cir_to_ofdm_c64": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_apply_ofdm_channel_c64": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_ofdm_channel_fused_c64": (_i32, [_vp, _vp, _vp, _vp, _vp, _u64, _u64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_ls_gather_scale_c64": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_gf2_encode_f32": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "samd_scramble_f32": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "samd_nr_prng_seq_f32": (_i32, [C.c_uint32, _i64, _vp, _vp]),
    "samd_ofdm_modulate_c64": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "samd_ofdm_demodulate_c64": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "samd_cir_to_time_c64": (_i32, [C.c_float, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                    _vp, _vp, _vp]),
    "samd_apply_time_channel_c64": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_lin_interp_c64": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_lin_interp_c128": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_lmmse_equalizer_c64": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, _vp]),
    "samd_ofdm_lmmse_c64": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                   _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "samd_mmse_pic_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_ofdm_mmse_pic_f32": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                      _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_ep_f32": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "samd_ofdm_ep_f32": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _i32, _vp, _vp]),
    "samd_kbest_f32": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp]),
    "samd_ml_detect_f32": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_ofdm_ml_f32": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_ofdm_kbest_f32": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                   _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp]),
    "samd_kbest_real_f32": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp]),
    "samd_ofdm_kbest_real_f32": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                   _i32, _i32, _i32, _i32, _i32, _f32, _i32, _vp, _vp]),
    "samd_crc_f32": (_i32, [_vp, _i64, _i32, C.c_uint32, _i32, _i32, _vp, _vp]),
    "samd_polar_encode_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "samd_polar_scl_register_stages": (_i32, [_i32, _i32, _i32]),
    "samd_polar_scl_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "samd_polar_scl_decode_f32": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.c_uint32, _i32, _vp, _vp,
                                         _vp, _sz, _vp]),
    "samd_polar5g_scl_decode_f32": (_i32, [_vp, _i32, _vp, _vp, _f32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.c_uint32, _i32,
                                           _vp, _vp, _vp, _sz, _vp]),
    "samd_polar_scl_decode_f64": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, C.c_uint32, _i32, _vp, _vp,
                                         _vp, _sz, _vp]),
    "samd_polar_scl_workspace_bytes_f64": (_sz, [_i32, _i32, _i32]),
    "samd_polar_bp_workspace_bytes": (_sz, [_i32, _i32]),
    "samd_polar_bp_decode_f32": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "samd_polar_bp_decode_f64": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _sz, _vp]),
    "samd_polar_bp_workspace_bytes_f64": (_sz, [_i32, _i32]),
    "samd_comm_unique_id": (_i32, [_vp]),
    "samd_comm_create": (_i32, [_vp, _i32, _i32, C.POINTER(_vp)]),
    "samd_comm_rank": (_i32, [_vp]),
    "samd_comm_world_size": (_i32, [_vp]),
    "samd_comm_allreduce_sum_i64": (_i32, [_vp, _vp, _i64, _vp]),
    "samd_comm_destroy": (None, [_vp]),
    "samd_count_errors_f32": (_i32, [_vp, _vp, _i64, _i64, _i32, _vp, _vp]),
    "samd_ofdm_lsnn_lmmse_c64": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32,
                                        _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "samd_debug_set_option": (_i32, [C.c_char_p, C.c_char_p]),
    "samd_debug_options_generation": (_i32, []),
    "samd_debug_get_option": (C.c_long, [C.c_char_p, C.c_char_p, _sz]),
}


def declared_symbols():
    """Names of all functions declared in include/sionna_amd.h."""
    with open(HEADER_PATH) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(samd_[a-z0-9_]+)\s*\(", src)))


class SamdError(RuntimeError):
    pass


def lib():
    """Load libsionna_amd.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C sionna_amd/csrc` (there is no CPU fallback).")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def set_option(key, value=None):
    """Development switch of the library (include/sionna_amd.h ``samd_debug_set_option``): ``value=None`` removes it.
    Takes effect for handles created afterwards - the Python blocks key their handle caches on the options generation,
    so the next call of a block builds a fresh handle."""
    check(lib().samd_debug_set_option(key.encode(), None if value is None else str(value).encode()), "samd_debug_set_option")


def get_option(key):
    """Current value of a development switch (str), or None when it is not set."""
    n = lib().samd_debug_get_option(key.encode(), None, 0)
    if n < 0:
        return None
    buf = C.create_string_buffer(n + 1)
    lib().samd_debug_get_option(key.encode(), buf, n + 1)
    return buf.value.decode()


class option:
    """``with _ffi.option("SAMD_X", 1): ...`` - set a development switch for the duration of a block."""

    def __init__(self, key, value="1"):
        self.key, self.value = key, value

    def __enter__(self):
        self.previous = get_option(self.key)                     # nested / pre-set switches come back as they were
        set_option(self.key, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.key, self.previous)
        return False


def options_generation():
    return int(lib().samd_debug_options_generation())


def check(rc, what=""):
    if rc == OK:
        return
    msg = lib().samd_last_error().decode()
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {msg}")
    if rc == ERR_INVALID:
        raise ValueError(f"{what}: {msg}")
    raise SamdError(f"{what}: error {rc}: {msg}")


_device = None


def device():
    """The compute device of this process: cuda:LOCAL_RANK.  Fails loudly without a GPU."""
    global _device
    if _device is None:
        if not torch.cuda.is_available():
            raise RuntimeError("sionna_amd needs a HIP device (MI355X); none is visible and there is "
                               "no CPU fallback.")
        idx = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        torch.cuda.set_device(idx)
        _device = torch.device("cuda", idx)
    return _device


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous tensor (or None)."""
    if t is None:
        return None
    if "_samd_pending" in getattr(t, "__dict__", ()):          # a deferred block output: fill it before its address is used
        from .phy.block import materialize
        materialize(t)
    assert t.is_cuda and t.is_contiguous()
    if t.numel() == 0:
        # an empty tensor has no storage (data_ptr() == 0), but the C entry points reject NULL before they look at the
        # sizes: hand them a valid address that a zero-sized operation never dereferences
        global _EMPTY
        if _EMPTY is None or _EMPTY.device != t.device:
            _EMPTY = torch.zeros(64, dtype=torch.float32, device=t.device)
        return C.c_void_p(_EMPTY.data_ptr())
    return C.c_void_p(t.data_ptr())


_EMPTY = None


_NP_OF = {torch.float32: "float32", torch.float64: "float64", torch.complex64: "complex64", torch.complex128: "complex128",
          torch.int32: "int32", torch.int64: "int64"}
_SCALARS = {}          # (value, dtype, device index) -> 0-dim device tensor, read-only by convention


def to_device(x, dtype):
    """numpy / python / torch -> contiguous device tensor of ``dtype``.

    Host scalars (a noise variance handed to every block of a chain, every Monte-Carlo iteration) are converted once per value:
    a pageable host-to-device copy blocks the host until the stream has drained - in the C4 chain the five of them per
    iteration left the GPU idle for 0.34 of 2.5 ms (profiles/r06j kernel trace).  The tensors are shared: treat them as
    read-only.  Host arrays are cast on the host (one copy, no conversion kernel)."""
    import numpy as np
    if isinstance(x, torch.Tensor):
        t = x
    else:
        a = np.asarray(x)
        npd = _NP_OF.get(dtype)
        if npd is not None and a.dtype != np.dtype(npd) and (a.dtype.kind != "c" or npd.startswith("complex")):
            a = a.astype(npd)
        if a.ndim == 0 and npd is not None:
            key = (a.item(), dtype, device().index)
            t = _SCALARS.get(key)
            if t is None:
                if len(_SCALARS) > 256:
                    _SCALARS.clear()
                t = torch.from_numpy(np.array(a)).to(device=device())
                _SCALARS[key] = t
            return t
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a if a.flags.writeable else a.copy())
    if t.dtype != dtype or t.device != device():
        t = t.to(device=device(), dtype=dtype)
    return t.contiguous()


class Workspace:
    """Grow-only scratch buffer owned by a block (the C-ABI never allocates)."""

    def __init__(self):
        self._buf = None

    def get(self, nbytes):
        if nbytes == 0:
            return None, 0
        if self._buf is None or self._buf.numel() < nbytes:
            self._buf = None
            self._buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device())
        return self._buf, self._buf.numel()
