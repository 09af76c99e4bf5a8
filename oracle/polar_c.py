"""ctypes binding of oracle/polar_scl.c (TEST INFRASTRUCTURE, see oracle/__init__.py) and the CRC-aided candidate
selection of PolarSCLDecoder.call (/root/reference/src/sionna/phy/fec/polar/decoding.py:1396-1419) /
Polar5GDecoder.call (:1999-2086) around it."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import polar as op

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_polar.so")
_lib = None
F = np.float32


def lib():
    global _lib
    if _lib is None:
        srcs = [os.path.join(_HERE, f) for f in ("polar_scl.c", "polar_scl_body.inc")]
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _lib = C.CDLL(_SO)
        _lib.oracle_polar_scl_decode.restype = C.c_int
        _lib.oracle_polar_scl_decode.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_int]
        for name, args in (("oracle_scl_T_f32", 1), ("oracle_scl_softplus_f32", 1), ("oracle_scl_cn_op_f32", 2)):
            fn = getattr(_lib, name)
            fn.restype, fn.argtypes = C.c_float, [C.c_float] * args
        _lib.oracle_scl_block_sum_f32.restype = C.c_float
        _lib.oracle_scl_block_sum_f32.argtypes = [C.c_void_p, C.c_int]
    return _lib


def num_threads():
    fn = lib().oracle_polar_num_threads
    fn.restype = C.c_int
    return int(fn())


def scl_list_decode(logits, frozen_pos, n, list_size, use_fast_scl=True, precision="f32", nthreads=0):
    """logits [B, n] float32 -> (uhat_list uint8 [B, 2L, n] in final sorted order, pm float64 [B, 2L])."""
    logits = np.ascontiguousarray(logits, F).reshape(-1, n)
    frozen = np.zeros(n, np.int32)
    frozen[np.asarray(frozen_pos, int)] = 1
    B = logits.shape[0]
    uhat = np.empty((B, 2 * list_size, n), np.uint8)
    pm = np.empty((B, 2 * list_size), np.float64)
    rc = lib().oracle_polar_scl_decode(n, list_size, frozen.ctypes.data, int(use_fast_scl), {"f32": 0, "f64": 1}[precision],
                                       logits.ctypes.data, B, uhat.ctypes.data, pm.ctypes.data, nthreads)
    assert rc == 0, rc
    return uhat, pm


def sc_decode(llr_logits, frozen_pos, n, precision="f32"):
    """PolarSCDecoder.call in the float32 specification arithmetic (precision "f64": float64 with the literal boxplus of the
    reference's NumPy twin): logits [B,n] -> u_hat at the info positions."""
    logits = np.ascontiguousarray(llr_logits, F if precision == "f32" else np.float64).reshape(-1, n)
    frozen = np.zeros(n, np.int32)
    frozen[np.asarray(frozen_pos, int)] = 1
    out = np.empty((logits.shape[0], n), np.uint8)
    fn = lib().oracle_polar_sc_decode if precision == "f32" else lib().oracle_polar_sc_decode_f64
    fn.restype, fn.argtypes = C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    assert fn(n, frozen.ctypes.data, logits.ctypes.data, logits.shape[0], out.ctypes.data) == 0
    return out[:, np.setdiff1d(np.arange(n), np.asarray(frozen_pos, int))].astype(F)


class SCLDecoder:
    """Drop-in for oracle.polar.SCLDecoder on the C list decoder: decode(logits) -> (u_hat [B,k], crc_status | None)."""

    def __init__(self, frozen_pos, n, list_size=8, crc_degree=None, use_fast_scl=True, ind_iil_inv=None, precision="f32"):
        self.n, self.L = n, list_size
        self.frozen_pos = np.asarray(frozen_pos, int)
        self.info_pos = np.setdiff1d(np.arange(n), self.frozen_pos)
        self.k = len(self.info_pos)
        self.crc_degree, self.fast, self.ind_iil_inv, self.precision = crc_degree, use_fast_scl, ind_iil_inv, precision

    def decode(self, llr_logits, nthreads=0):
        uhat, pm = scl_list_decode(llr_logits, self.frozen_pos, self.n, self.L, self.fast, self.precision, nthreads)
        rd = F if self.precision == "f32" else np.float64
        pm = pm.astype(rd)
        B = uhat.shape[0]
        crc_valid = None
        if self.crc_degree is not None:                                    # :1396-1412
            u_list = uhat[:, :, self.info_pos].astype(F)
            if self.ind_iil_inv is not None:
                u_list = u_list[:, :, self.ind_iil_inv]
            _, crc_valid = op.crc_check(u_list, self.crc_degree)
            pm = pm + (rd(1.) - crc_valid[..., 0].astype(rd)) * rd(op.LLR_MAX) * rd(self.k)
        cand = np.argmin(pm, axis=-1)                                      # first minimum (:1415)
        c_hat = uhat[np.arange(B), cand, :].astype(F)
        status = crc_valid[np.arange(B), cand, 0] if crc_valid is not None else None
        return c_hat[:, self.info_pos], status


def polar5g_decode(code, llr_logits, list_size=8, precision="f32", return_crc_status=False, dec_type="SCL"):
    """oracle.polar.polar5g_decode on the C decoders (dec_type "SCL" or "SC")."""
    llr = np.asarray(llr_logits, F)
    lead = llr.shape[:-1]
    llr = llr.reshape(-1, code.n_target)
    n, npol = code.n_target, code.n_polar
    if code.channel_type == "uplink":
        llr = llr[:, np.argsort(op.channel_interleaver(np.arange(n)))]
    if n >= npol:
        n_rep = n - npol
        dem = np.concatenate([llr[:, :n_rep] + llr[:, npol:], llr[:, n_rep:npol]], 1)
    elif code.k_polar / n <= 7 / 16:
        dem = np.concatenate([np.zeros([llr.shape[0], npol - n], F), llr], 1)
    else:
        dem = np.concatenate([llr, -F(100.) * np.ones([llr.shape[0], npol - n], F)], 1)
    dec_in = dem[:, np.argsort(op.subblock_interleaving(np.arange(npol)))]
    iil_inv = np.argsort(code.ind_input_int) if code.channel_type == "downlink" else None
    if dec_type == "SC":
        u_crc = sc_decode(dec_in, code.frozen_pos, npol)
    else:
        u_crc, _ = SCLDecoder(code.frozen_pos, npol, list_size, code.crc_degree, ind_iil_inv=iil_inv,
                              precision=precision).decode(dec_in)
    if iil_inv is not None:
        u_crc = u_crc[:, iil_inv]
    status = op.crc_check(u_crc, code.crc_degree)[1][..., 0]            # the dedicated CRC decoder of :2062-2067
    out = u_crc[:, :-code.k_crc].reshape(lead + (code.k_target,))
    return (out, status.reshape(lead)) if return_crc_status else out
