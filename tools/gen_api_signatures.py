#!/usr/bin/env python3
"""Generates tests/golden/api_signatures.json: the constructor / call / function signatures of the reference's hot-path
surface (SURVEY.md appendix B + section 8), read from the reference's source files with ``ast`` (nothing is executed):
for every class its ``__init__`` and ``call`` parameters (names, order, defaults as source text), for every function its
parameters.  tests/test_api_signatures.py holds ``sionna_amd.phy`` to it: the drop-in boundary of SURVEY 8(b) - same
names, same order, same defaults.  Run here (needs /root/reference)."""
import ast
import json
import os

REF = "/root/reference/src/sionna/phy"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "api_signatures.json")

SURFACE = {      # reference file -> (import path in sionna.phy, names)
    "mapping.py": ("mapping", ["Constellation", "Mapper", "Demapper", "SymbolDemapper", "BinarySource", "QAMSource", "SymbolLogits2LLRs",
                               "LLRs2SymbolLogits", "SymbolLogits2Moments", "SymbolInds2Bits", "QAM2PAM", "PAM2QAM", "qam", "pam", "pam_gray"]),
    "fec/ldpc/encoding.py": ("fec.ldpc", ["LDPC5GEncoder"]),
    "fec/ldpc/decoding.py": ("fec.ldpc", ["LDPCBPDecoder", "LDPC5GDecoder", "vn_update_sum", "cn_update_minsum", "cn_update_offset_minsum",
                                          "cn_update_phi", "cn_update_tanh"]),
    "fec/polar/encoding.py": ("fec.polar", ["PolarEncoder", "Polar5GEncoder"]),
    "fec/polar/decoding.py": ("fec.polar", ["PolarSCDecoder", "PolarSCLDecoder", "PolarBPDecoder", "Polar5GDecoder"]),
    "fec/polar/utils.py": ("fec.polar.utils", ["generate_5g_ranking", "generate_rm_code", "generate_polar_transform_mat", "generate_dense_polar"]),
    "fec/crc.py": ("fec.crc", ["CRCEncoder", "CRCDecoder"]),
    "fec/scrambling.py": ("fec.scrambling", ["Scrambler", "TB5GScrambler", "Descrambler"]),
    "fec/interleaving.py": ("fec.interleaving", ["RowColumnInterleaver", "RandomInterleaver", "Deinterleaver"]),
    "fec/linear/encoding.py": ("fec.linear", ["LinearEncoder"]),
    "fec/utils.py": ("fec.utils", ["GaussianPriorSource", "llr2mi", "j_fun", "j_fun_inv", "load_parity_check_examples", "alist2mat", "load_alist",
                                   "make_systematic", "gm2pcm", "pcm2gm", "verify_gm_pcm", "bin2int", "int2bin", "bin2int_tf", "int2bin_tf",
                                   "int_mod_2", "generate_reg_ldpc"]),
    "nr/tb_encoder.py": ("nr", ["TBEncoder"]),
    "nr/tb_decoder.py": ("nr", ["TBDecoder"]),
    "channel/awgn.py": ("channel", ["AWGN"]),
    "channel/ofdm_channel.py": ("channel", ["OFDMChannel"]),
    "channel/generate_ofdm_channel.py": ("channel", ["GenerateOFDMChannel"]),
    "channel/apply_ofdm_channel.py": ("channel", ["ApplyOFDMChannel"]),
    "channel/time_channel.py": ("channel", ["TimeChannel"]),
    "channel/generate_time_channel.py": ("channel", ["GenerateTimeChannel"]),
    "channel/apply_time_channel.py": ("channel", ["ApplyTimeChannel"]),
    "channel/rayleigh_block_fading.py": ("channel", ["RayleighBlockFading"]),
    "channel/flat_fading_channel.py": ("channel", ["GenerateFlatFadingChannel", "ApplyFlatFadingChannel", "FlatFadingChannel"]),
    "channel/spatial_correlation.py": ("channel", ["KroneckerModel", "PerColumnModel"]),
    "channel/utils.py": ("channel", ["subcarrier_frequencies", "cir_to_ofdm_channel", "cir_to_time_channel", "time_lag_discrete_time_channel",
                                     "exp_corr_mat", "one_ring_corr_mat"]),
    "channel/tr38901/tdl.py": ("channel.tr38901", ["TDL"]),
    "channel/tr38901/cdl.py": ("channel.tr38901", ["CDL"]),
    "channel/tr38901/antenna.py": ("channel.tr38901", ["Antenna", "AntennaArray", "PanelArray"]),
    "mimo/stream_management.py": ("mimo", ["StreamManagement"]),
    "mimo/equalization.py": ("mimo", ["lmmse_matrix", "lmmse_equalizer", "zf_equalizer", "mf_equalizer"]),
    "mimo/utils.py": ("mimo", ["whiten_channel", "complex2real_vector", "real2complex_vector", "complex2real_matrix", "real2complex_matrix",
                               "complex2real_covariance", "real2complex_covariance", "complex2real_channel", "real2complex_channel"]),
    "utils/linalg.py": ("utils", ["inv_cholesky", "matrix_pinv"]),
    "mimo/detection.py": ("mimo", ["LinearDetector", "KBestDetector", "EPDetector", "MMSEPICDetector", "MaximumLikelihoodDetector"]),
    "ofdm/resource_grid.py": ("ofdm", ["ResourceGrid", "ResourceGridMapper", "ResourceGridDemapper", "RemoveNulledSubcarriers"]),
    "ofdm/pilot_pattern.py": ("ofdm", ["PilotPattern", "EmptyPilotPattern", "KroneckerPilotPattern"]),
    "ofdm/channel_estimation.py": ("ofdm", ["LSChannelEstimator", "NearestNeighborInterpolator", "LinearInterpolator", "LMMSEInterpolator",
                                            "tdl_freq_cov_mat", "tdl_time_cov_mat"]),
    "ofdm/equalization.py": ("ofdm", ["OFDMEqualizer", "LMMSEEqualizer", "ZFEqualizer", "MFEqualizer"]),
    "ofdm/detection.py": ("ofdm", ["LinearDetector", "KBestDetector", "EPDetector", "MMSEPICDetector", "MaximumLikelihoodDetector",
                                    "MaximumLikelihoodDetectorWithPrior"]),
    "ofdm/modulator.py": ("ofdm", ["OFDMModulator"]),
    "ofdm/demodulator.py": ("ofdm", ["OFDMDemodulator"]),
    "utils/misc.py": ("utils", ["ebnodb2no", "hard_decisions", "sim_ber", "complex_normal"]),
    "utils/metrics.py": ("utils", ["compute_ber", "compute_bler", "count_errors", "count_block_errors"]),
    "utils/plotting.py": ("utils", ["PlotBER", "plot_ber"]),
}


def params(fn):
    a = fn.args
    pos = a.posonlyargs + a.args
    defaults = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
    out = [[p.arg, d] for p, d in zip(pos, defaults)]
    if a.vararg:
        out.append(["*" + a.vararg.arg, None])
    for p, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append([p.arg, None if d is None else ast.unparse(d)])
    if a.kwarg:
        out.append(["**" + a.kwarg.arg, None])
    return [p for p in out if p[0] not in ("self", "cls")]


def main():
    table = {}
    for rel, (mod, names) in SURFACE.items():
        path = os.path.join(REF, rel)
        tree = ast.parse(open(path).read())
        found = {}
        for node in tree.body:
            if isinstance(node, ast.ClassDef) and node.name in names:
                entry = {"kind": "class", "public": []}
                for item in node.body:
                    if isinstance(item, ast.FunctionDef) and item.name in ("__init__", "call", "__call__"):
                        entry[item.name] = params(item)
                    if isinstance(item, ast.FunctionDef) and not item.name.startswith("_") and item.name not in ("call", "build"):
                        is_prop = any((isinstance(d, ast.Name) and d.id == "property") or (isinstance(d, ast.Attribute) and d.attr in ("setter", "getter"))
                                      for d in item.decorator_list)
                        if item.name not in [q[0] for q in entry["public"]]:
                            entry["public"].append([item.name, "property" if is_prop else "method", None if is_prop else params(item)])
                found[node.name] = entry
            elif isinstance(node, ast.FunctionDef) and node.name in names:
                found[node.name] = {"kind": "function", "params": params(node)}
        missing = [n for n in names if n not in found]
        assert not missing, (rel, missing)
        for n, e in found.items():
            table[f"{mod}.{n}"] = dict(e, file=rel)
    with open(OUT, "w") as f:
        json.dump({"_comment": "reference signatures by ast (tools/gen_api_signatures.py); defaults as source text", "signatures": table}, f, indent=1)
    print(len(table), "signatures ->", OUT)


if __name__ == "__main__":
    main()
