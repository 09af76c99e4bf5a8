"""ctypes binding of oracle/ldpc_bp.c (test infrastructure, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_ldpc.so")
_lib = None

CN_MODES = {"boxplus": 0, "boxplus-phi": 1, "minsum": 2, "min": 2, "offset-minsum": 3}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "ldpc_bp.c")):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_ldpc_bp_decode.restype = C.c_int
        _lib.oracle_ldpc_bp_decode.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                               C.c_int, C.c_int]
        _lib.oracle_ldpc_bp_decode_simd.restype = C.c_int
        _lib.oracle_ldpc_bp_decode_simd.argtypes = _lib.oracle_ldpc_bp_decode.argtypes
        _lib.oracle_num_threads.restype = C.c_int
        for fn in (_lib.oracle_phi_f32, _lib.oracle_spec_exp_f32, _lib.oracle_spec_log_f32, _lib.oracle_phi_exp_f32, _lib.oracle_phi_log_f32):
            fn.restype = None
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    return _lib


def _elementwise(fn, x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    fn(x.ctypes.data, y.ctypes.data, x.size)
    return y


def phi_f32(x):
    """phi of the boxplus-phi rule on the DEFINED float32 exp / log (ldpc_bp.c: Cephes / Eigen restatement)."""
    return _elementwise(lib().oracle_phi_f32, x)


def spec_exp_f32(x):
    return _elementwise(lib().oracle_spec_exp_f32, x)


def spec_log_f32(x):
    return _elementwise(lib().oracle_spec_log_f32, x)


def phi_exp_f32(x):
    """exp of the boxplus-phi rule (round 5: magic-number rounding, exponent-field scaling; ldpc_bp.c phi_expf)"""
    return _elementwise(lib().oracle_phi_exp_f32, x)


def phi_log_f32(x):
    """log of the boxplus-phi rule (round 5: table-driven, ldpc_bp.c phi_logf)"""
    return _elementwise(lib().oracle_phi_log_f32, x)


def bp_decode(dec, llr, num_iter=None, hard_out=None, offset=0.5, nthreads=0, simd=False):
    """Run the C oracle on the graph of ``dec`` (an oracle.ldpc_bp.LDPCBPDecoder).
    llr [B, N_vn] logits -> [B, N_vn].  simd=True: the 8-codewords-at-a-time form of the same decoder (bit-identical
    outputs; bench.py's CPU baseline) - min-sum, offset-min-sum, boxplus-phi."""
    llr = np.ascontiguousarray(llr, np.float32)
    assert llr.ndim == 2 and llr.shape[1] == dec.num_vns
    cn = np.ascontiguousarray(dec.cn_idx, np.int32)
    vn = np.ascontiguousarray(dec.vn_idx, np.int32)
    out = np.empty_like(llr)
    mode = CN_MODES[[k for k, v in __import__("oracle.ldpc_bp", fromlist=["_CN"])._CN.items()
                     if v is dec._cn_update][0]]
    fn = lib().oracle_ldpc_bp_decode_simd if simd else lib().oracle_ldpc_bp_decode
    rc = fn(dec.num_edges, dec.num_cns, dec.num_vns, cn.ctypes.data, vn.ctypes.data,
                                     llr.ctypes.data, out.ctypes.data, llr.shape[0],
                                     dec.num_iter if num_iter is None else num_iter, mode, float(dec.llr_max),
                                     float(offset), int(dec.hard_out if hard_out is None else hard_out), nthreads)
    assert rc == 0
    return out


def num_threads():
    return lib().oracle_num_threads()
