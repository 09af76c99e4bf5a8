"""Polar decoders - mirror of ``sionna.phy.fec.polar.PolarSCDecoder`` / ``PolarSCLDecoder`` / ``PolarBPDecoder`` /
``Polar5GDecoder`` (reference src/sionna/phy/fec/polar/decoding.py:15-263, 266-1437, 1440-1771, 1774-2086).

The decoding tree of the reference's recursion (decoding.py:919-1005, with the fast-SCL
rate-0 / repetition shortcuts of :525-599) is flattened ONCE on the host into a list of operations
that ``samd_polar_scl_decode_f32`` interprets with one wave per codeword: the engine of
csrc/polar_scl_reg.hip (SC and list sizes 1..32 at n >= 64: low tree stages in registers, whole sub-trees as
one schedule record) or the generic engine of csrc/polar.hip (everything in LDS / L2 scratch).  The hybrid
mode runs SC first and the list decoder on the words whose CRC fails, like the reference.  ``PolarBPDecoder`` is
``samd_polar_bp_decode_f32`` (csrc/polar_bp.hip): the whole iterative decode of a codeword in the LDS of one workgroup."""
import numbers

import numpy as np
import torch

from .... import _ffi
from ...block import Block
from ..crc import CRCEncoder, CRCDecoder, crc_poly_mask
from .encoding import Polar5GEncoder, _channel_pattern, _subblock_pattern, _input_pattern

OP_F, OP_G, OP_LEAF, OP_RATE0, OP_REP, OP_COMBINE, OP_END, OP_SUBTREE = range(8)


def build_schedule(frozen_ind, use_fast=True, use_rep=True, subtree_stage=None):
    """Flatten the SC(L) decoding recursion into [num_ops, 4] int32 records (op, a0, a1, a2).

    F/G:      a0 = stage s of the inputs (outputs go to stage s-1)
    LEAF:     a0 = 0, a1 = side (0 left / 1 right result bank), a2 = bit index (info) or -1-index (frozen)
    RATE0:    a0 = stage, a1 = side
    REP:      a0 = stage, a1 = side, a2 = index of the node's only information bit (its last)
    COMBINE:  a0 = stage of the two children, a1 = side of the parent
    SUBTREE:  a0 = stage, a1 = side, a2 = index of its first bit (+4096 when the fast-SCL shortcuts apply inside):
              the whole node as ONE record - emitted for every node of stage ``subtree_stage`` (the engine whose low
              stages live in registers decodes it from the frozen pattern, ``samd_polar_scl_register_stages``)
    """
    frozen_ind = np.asarray(frozen_ind).astype(int)
    ops = []

    def node(start, s, side, root):
        size = 1 << s
        if s == 0:
            ops.append((OP_LEAF, 0, side, -1 - start if frozen_ind[start] else start))
            return
        blk = frozen_ind[start:start + size]
        if subtree_stage is not None and s == subtree_stage and not root:
            ops.append((OP_SUBTREE, s, side, start + (4096 if use_fast else 0)))
            return
        if use_fast:
            if blk.sum() == size:
                ops.append((OP_RATE0, s, side, 0))
                return
            if use_rep and blk[-1] == 0 and blk[:-1].sum() == size - 1:
                ops.append((OP_REP, s, side, start + size - 1))
                return
        ops.append((OP_F, s, 0, 0))
        node(start, s - 1, 0, False)
        ops.append((OP_G, s, 0, 0))
        node(start + size // 2, s - 1, 1, False)
        if not root:
            ops.append((OP_COMBINE, s - 1, side, 0))

    node(0, int(np.log2(len(frozen_ind))), 0, True)
    ops.append((OP_END, 0, 0, 0))
    return np.asarray(ops, np.int32)


def fuse_schedule(ops):
    """Replace every stage-1 node with two information leaves - the five records F(1), LEAF(i), G(1), LEAF(i+1),
    COMBINE(0, side) - by one SUBTREE record of stage 1 (a1 = side, a2 = i), which every engine accepts.  Same
    operations in the same order; the kernels save four of five schedule dispatches on what is 44 % of the schedule
    of a rate-1/2 n = 1024 code."""
    ops = np.asarray(ops, np.int32)
    out, i, n = [], 0, len(ops)
    while i < n:
        w = ops[i:i + 5]
        if (len(w) == 5 and w[0, 0] == OP_F and w[0, 1] == 1 and w[1, 0] == OP_LEAF and w[1, 3] >= 0 and w[1, 2] == 0
                and w[2, 0] == OP_G and w[2, 1] == 1 and w[3, 0] == OP_LEAF and w[3, 3] == w[1, 3] + 1 and w[3, 2] == 1
                and w[4, 0] == OP_COMBINE and w[4, 1] == 0):
            out.append((OP_SUBTREE, 1, w[4, 2], w[1, 3]))
            i += 5
        else:
            out.append(tuple(ops[i]))
            i += 1
    return np.asarray(out, np.int32)


def pack_schedule(ops):
    """One int32 per operation: op | stage<<3 | side<<7 | (a2+2048)<<8 (bit 20 = the +4096 of SUBTREE records),
    stage-1 nodes of two information leaves fused (:func:`fuse_schedule`)."""
    ops = np.asarray(fuse_schedule(ops), np.int64)
    a0 = np.where(ops[:, 0] == OP_LEAF, 0, ops[:, 1])
    return (ops[:, 0] | (a0 << 3) | (ops[:, 2] << 7) | ((ops[:, 3] + 2048) << 8)).astype(np.int32)


class _PolarListDecoderBase(Block):
    """Shared engine of the SC and SCL decoders."""

    def _setup(self, frozen_pos, n, list_size, sc_mode, crc_degree, use_fast, ind_iil_inv):
        if not isinstance(n, numbers.Number):
            raise TypeError("n must be a number.")
        n = int(n)
        frozen_pos = np.asarray(frozen_pos)
        if not np.issubdtype(frozen_pos.dtype, np.integer):
            raise TypeError("frozen_pos contains non int.")
        if len(frozen_pos) > n:
            raise ValueError("Num. of elements in frozen_pos cannot be greater than n.")
        if np.log2(n) != int(np.log2(n)):
            raise ValueError("n must be a power of 2.")
        self._n = n
        self._frozen_pos = frozen_pos
        self._k = n - len(frozen_pos)
        self._info_pos = np.setdiff1d(np.arange(n), frozen_pos)
        if self._k != len(self._info_pos):
            raise ArithmeticError("Internal error: invalid info_pos generated.")
        self._frozen_ind = np.zeros(n, int)
        self._frozen_ind[frozen_pos] = 1
        self._list_size, self._sc_mode = int(list_size), int(sc_mode)
        self._llr_max = 30.
        # the SC decoder of the reference prunes rate-0 sub-trees only (decoding.py:186-190)
        self._use_fast = bool(use_fast)
        self._ops = build_schedule(self._frozen_ind, use_fast, use_rep=not sc_mode)
        self._crc_len, self._crc_mask = (crc_poly_mask(crc_degree) if crc_degree is not None else (0, 0))
        self._ind_iil_inv = None if ind_iil_inv is None else np.asarray(ind_iil_inv, np.int32)
        self._dev = None

    n = property(lambda self: self._n)
    k = property(lambda self: self._k)
    frozen_pos = property(lambda self: self._frozen_pos)
    info_pos = property(lambda self: self._info_pos)
    llr_max = property(lambda self: self._llr_max)

    def _decode_2d(self, llr, want_status=False, rm=None):
        """rm = (src_a, src_b or None, fill): Polar5GDecoder's rate recovery as an index inside the kernel's channel-LLR load
        (samd_polar5g_scl_decode_f32; ``llr`` then has the received length) - the caller checked ``_rm_in_kernel()``"""
        if self._dev is None or self._dev_gen != _ffi.options_generation():   # (a development switch changed: rebuild)
            self._dev_gen = _ffi.options_generation()
            i32 = lambda a: _ffi.to_device(np.ascontiguousarray(a, np.int32), torch.int32)
            # the engine that will run says which stages it keeps in registers: it decodes a whole node one stage above
            # them (f / g from memory around two register-stage subtrees) without further schedule dispatch
            # (precision="double": always the generic engine, csrc/polar.hip polar_scl_kernel<64, double>)
            r = -1 if self.precision == "double" else _ffi.lib().samd_polar_scl_register_stages(self._n, self._list_size, self._sc_mode)
            ops = self._ops if r < 1 else build_schedule(self._frozen_ind, self._use_fast, use_rep=not self._sc_mode,
                                                         subtree_stage=r + 1)
            self._dev = (i32(pack_schedule(ops)), i32(self._info_pos),
                         i32(self._ind_iil_inv) if self._ind_iil_inv is not None else None)
        ops, info, iil = self._dev
        b = llr.shape[0]
        dbl = self.precision == "double"
        lib = _ffi.lib()
        u_hat = torch.empty((b, self._k), dtype=self.rdtype, device=llr.device)
        status = torch.empty((b,), dtype=self.rdtype, device=llr.device) if want_status else None
        if b > 0:
            if getattr(self, "_ws", None) is None:
                self._ws = _ffi.Workspace()
            ws, ws_bytes = self._ws.get((lib.samd_polar_scl_workspace_bytes_f64 if dbl else lib.samd_polar_scl_workspace_bytes)(
                b, self._n, self._list_size))
            if rm is not None:
                _ffi.check(lib.samd_polar5g_scl_decode_f32(
                    _ffi.ptr(llr), int(llr.shape[1]), _ffi.ptr(rm[0]), _ffi.ptr(rm[1]), float(rm[2]), _ffi.ptr(ops), ops.numel(),
                    _ffi.ptr(info), _ffi.ptr(iil), b, self._n, self._k, self._list_size, self._sc_mode, self._crc_mask, self._crc_len,
                    _ffi.ptr(u_hat), _ffi.ptr(status), _ffi.ptr(ws), ws_bytes, _ffi.stream()), type(self).__name__)
            else:
                _ffi.check((lib.samd_polar_scl_decode_f64 if dbl else lib.samd_polar_scl_decode_f32)(
                    _ffi.ptr(llr), _ffi.ptr(ops), ops.numel(), _ffi.ptr(info), _ffi.ptr(iil), b, self._n, self._k, self._list_size,
                    self._sc_mode, self._crc_mask, self._crc_len, _ffi.ptr(u_hat), _ffi.ptr(status), _ffi.ptr(ws), ws_bytes,
                    _ffi.stream()), type(self).__name__)
        return u_hat, status

    def _rm_in_kernel(self):
        """the register engine takes Polar5GDecoder's rate recovery as an index table (single precision)"""
        return self.precision != "double" and _ffi.lib().samd_polar_scl_register_stages(self._n, self._list_size, self._sc_mode) >= 0

    def build(self, input_shape):
        if input_shape[-1] != self._n:
            raise ValueError("Invalid input shape.")


class PolarSCDecoder(_PolarListDecoderBase):
    """``PolarSCDecoder(frozen_pos, n)(llr_ch[..., n]) -> u_hat[..., k]`` (decoding.py:15-263)."""

    def __init__(self, frozen_pos, n, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        self._setup(frozen_pos, n, 1, 1, None, True, None)     # rate-0 shortcut = use_fast_sc (:186-190)

    def call(self, llr_ch, /):
        llr = _ffi.to_device(llr_ch, self.rdtype)
        if llr.shape[-1] != self._n:
            raise ValueError("Invalid input shape.")
        u_hat, _ = self._decode_2d(llr.reshape(-1, self._n))
        return u_hat.reshape(tuple(llr.shape[:-1]) + (self._k,))


class PolarSCLDecoder(_PolarListDecoderBase):
    """``PolarSCLDecoder(frozen_pos, n, list_size=8, crc_degree=None, use_hybrid_sc=False,
    use_fast_scl=True, cpu_only=False, use_scatter=False, ind_iil_inv=None, return_crc_status=False)``
    (decoding.py:266-1437).  ``cpu_only`` / ``use_scatter`` only select between equivalent TF
    implementations in the reference and are accepted and ignored."""

    def __init__(self, frozen_pos, n, list_size=8, crc_degree=None, use_hybrid_sc=False, use_fast_scl=True,
                 cpu_only=False, use_scatter=False, ind_iil_inv=None, return_crc_status=False, precision=None,
                 **kwargs):
        super().__init__(precision=precision, **kwargs)
        for name, v in (("cpu_only", cpu_only), ("use_scatter", use_scatter), ("use_fast_scl", use_fast_scl),
                        ("use_hybrid_sc", use_hybrid_sc), ("return_crc_status", return_crc_status)):
            if not isinstance(v, bool):
                raise TypeError(f"{name} must be bool.")
        if not isinstance(list_size, int):
            raise TypeError("list_size must be integer.")
        if np.log2(list_size) != int(np.log2(list_size)):
            raise ValueError("list_size must be a power of 2.")
        self._use_hybrid_sc = use_hybrid_sc
        self._setup(frozen_pos, n, list_size, 0, crc_degree, use_fast_scl, ind_iil_inv)
        if use_hybrid_sc:
            # SC first, SCL only for the words whose CRC fails after SC (decoding.py:474-480, 1292-1334)
            if crc_degree is None:
                raise ValueError("Hybrid SC requires outer CRC.")
            self._decoder_sc = PolarSCDecoder(frozen_pos, n, precision=precision)
        if crc_degree is not None:
            self._crc_encoder = CRCEncoder(crc_degree, precision=precision)
            self._crc_decoder = CRCDecoder(self._crc_encoder, precision=precision)
            self._k_crc = self._crc_encoder.crc_length
        else:
            self._k_crc = 0
        if self._k < self._k_crc:
            raise ValueError("Value of k is too small for given CRC_degree.")
        if crc_degree is None and return_crc_status:
            raise ValueError("Returning CRC status requires given crc_degree.")
        if ind_iil_inv is not None and len(ind_iil_inv) != self._k:
            raise ValueError("ind_int must be of length k+k_crc.")
        self._return_crc_status = return_crc_status

    list_size = property(lambda self: self._list_size)
    k_crc = property(lambda self: self._k_crc)

    def call(self, llr_ch):
        llr = _ffi.to_device(llr_ch, self.rdtype)
        if llr.shape[-1] != self._n:
            raise ValueError("Invalid input shape.")
        llr2d = llr.reshape(-1, self._n)
        if self._use_hybrid_sc and llr2d.shape[0] > 0:
            u_hat, _ = self._decoder_sc._decode_2d(llr2d)
            _, valid = self._crc_decoder(u_hat)                       # as the reference: no de-interleaving here
            valid = valid.reshape(-1)
            redo = torch.nonzero(~valid).reshape(-1)                   # indices of the words that need the list decoder
            status = None
            if self._return_crc_status:
                chk = u_hat if self._ind_iil_inv is None else \
                    u_hat[:, _ffi.to_device(np.asarray(self._ind_iil_inv, np.int64), torch.int64)]
                status = self._crc_decoder(chk.contiguous())[1].reshape(-1).to(self.rdtype)
            if redo.numel() > 0:
                u_scl, st = self._decode_2d(llr2d.index_select(0, redo).contiguous(), self._return_crc_status)
                u_hat.index_copy_(0, redo, u_scl)
                if self._return_crc_status:
                    status.index_copy_(0, redo, st)
        else:
            u_hat, status = self._decode_2d(llr2d, self._return_crc_status)
        u_hat = u_hat.reshape(tuple(llr.shape[:-1]) + (self._k,))
        if self._return_crc_status:
            return u_hat, (status > 0.5).reshape(tuple(llr.shape[:-1]))
        return u_hat


class PolarBPDecoder(Block):
    """``PolarBPDecoder(frozen_pos, n, num_iter=20, hard_out=True)(llr_ch[..., n]) -> u_hat[..., k]``
    (decoding.py:1440-1771): flooding belief propagation on the polar factor graph, boxplus evaluated literally on
    inputs clipped to +-19.3 (:1587-1603); hard bits or soft logits of the k information positions.  One launch of
    ``samd_polar_bp_decode_f32`` runs all iterations (the reference notes that unrolling its graph "can become time and
    memory consuming" - here the messages of a codeword never leave LDS up to n = 1024)."""

    def __init__(self, frozen_pos, n, num_iter=20, hard_out=True, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(n, numbers.Number):
            raise TypeError("n must be a number.")
        n = int(n)
        frozen_pos = np.asarray(frozen_pos)
        if not np.issubdtype(frozen_pos.dtype, np.integer):
            raise TypeError("frozen_pos contains non int.")
        if len(frozen_pos) > n:
            raise ValueError("Num. of elements in frozen_pos cannot be greater than n.")
        if np.log2(n) != int(np.log2(n)):
            raise ValueError("n must be a power of 2.")
        if not isinstance(hard_out, bool):
            raise TypeError("hard_out must be boolean.")
        self._n = n
        self._frozen_pos = frozen_pos
        self._k = n - len(frozen_pos)
        self._info_pos = np.setdiff1d(np.arange(n), frozen_pos)
        if self._k != len(self._info_pos):
            raise ArithmeticError("Internal error: invalid info_pos generated.")
        if not isinstance(num_iter, int):
            raise TypeError("num_iter must be integer.")
        if num_iter <= 0:
            raise ValueError("num_iter must be a positive value.")
        self._num_iter = num_iter
        self._llr_max = 19.3
        self._hard_out = hard_out
        self._n_stages = int(np.log2(n))
        self._dev = None
        self._ws = None

    n = property(lambda self: self._n)
    k = property(lambda self: self._k)
    frozen_pos = property(lambda self: self._frozen_pos)
    info_pos = property(lambda self: self._info_pos)
    llr_max = property(lambda self: self._llr_max)
    hard_out = property(lambda self: self._hard_out)

    @property
    def num_iter(self):
        return self._num_iter

    @num_iter.setter
    def num_iter(self, num_iter):
        if not isinstance(num_iter, int):
            raise ValueError("num_iter must be int.")
        if num_iter < 0:
            raise ValueError("num_iter cannot be negative.")
        self._num_iter = num_iter

    def build(self, input_shape):
        if input_shape[-1] != self._n:
            raise ValueError("Invalid input shape")

    def call(self, llr_ch):
        dbl = self.precision == "double"        # float64: polar_bp_kernel<.., double> on libm exp / log
        llr = _ffi.to_device(llr_ch, self.rdtype)
        if llr.shape[-1] != self._n:
            raise ValueError("Invalid input shape")
        if self._num_iter < 1:
            raise ValueError("num_iter must be a positive value.")
        if self._dev is None or self._dev_gen != _ffi.options_generation():   # (a development switch changed: rebuild)
            self._dev_gen = _ffi.options_generation()
            prior = np.zeros(self._n, np.float64 if dbl else np.float32)
            prior[self._frozen_pos] = np.float32(self._llr_max)        # decoding.py:1632-1636 (19.3 as a float32, cast)
            self._dev = (_ffi.to_device(prior, self.rdtype),
                         _ffi.to_device(np.ascontiguousarray(self._info_pos, np.int32), torch.int32))
        prior, info = self._dev
        llr2d = llr.reshape(-1, self._n)
        b = llr2d.shape[0]
        u_hat = torch.empty((b, self._k), dtype=self.rdtype, device=llr.device)
        if b > 0:
            if self._ws is None:
                self._ws = _ffi.Workspace()
            lib = _ffi.lib()
            ws, ws_bytes = self._ws.get((lib.samd_polar_bp_workspace_bytes_f64 if dbl else lib.samd_polar_bp_workspace_bytes)(b, self._n))
            _ffi.check((lib.samd_polar_bp_decode_f64 if dbl else lib.samd_polar_bp_decode_f32)(
                _ffi.ptr(llr2d), _ffi.ptr(prior), _ffi.ptr(info), b, self._n, self._k, self._num_iter, int(self._hard_out),
                _ffi.ptr(u_hat), _ffi.ptr(ws), ws_bytes, _ffi.stream()), "PolarBPDecoder")
        return u_hat.reshape(tuple(llr.shape[:-1]) + (self._k,))


class Polar5GDecoder(Block):
    """``Polar5GDecoder(enc_polar, dec_type="SC", list_size=8, num_iter=20, return_crc_status=False)``:
    rate recovery (channel de-interleaving, repetition combining / puncturing zeros / shortening
    LLRs, sub-block de-interleaving), SC or CRC-aided SCL decoding, CRC removal
    (decoding.py:1774-2086)."""

    def __init__(self, enc_polar, dec_type="SC", list_size=8, num_iter=20, return_crc_status=False,
                 precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(enc_polar, Polar5GEncoder):
            raise TypeError("enc_polar must be Polar5GEncoder.")
        if not isinstance(dec_type, str):
            raise TypeError("dec_type must be str.")
        if not isinstance(return_crc_status, bool):
            raise TypeError("return_crc_status must be bool.")
        self._enc_polar = enc_polar
        self._k_target, self._n_target = enc_polar.k_target, enc_polar.n_target
        self._k_polar, self._n_polar = enc_polar.k_polar, enc_polar.n_polar
        self._k_crc = enc_polar.enc_crc.crc_length
        self._bil = enc_polar._channel_type == "uplink"
        self._iil = enc_polar._channel_type == "downlink"
        self._llr_max = 100.
        self._dec_type = dec_type
        self._return_crc_status = return_crc_status
        self._ind_iil_inv = np.argsort(_input_pattern(self._k_polar)) if self._iil else None
        if dec_type == "SC":
            self._polar_dec = PolarSCDecoder(enc_polar.frozen_pos, self._n_polar, precision=precision)
        elif dec_type == "SCL":
            self._polar_dec = PolarSCLDecoder(enc_polar.frozen_pos, self._n_polar, crc_degree=enc_polar.enc_crc.crc_degree,
                                              list_size=list_size, ind_iil_inv=self._ind_iil_inv, precision=precision)
        elif dec_type == "hybSCL":
            self._polar_dec = PolarSCLDecoder(enc_polar.frozen_pos, self._n_polar, crc_degree=enc_polar.enc_crc.crc_degree,
                                              list_size=list_size, use_hybrid_sc=True, ind_iil_inv=self._ind_iil_inv,
                                              precision=precision)
        elif dec_type == "BP":
            if not isinstance(num_iter, int):
                raise TypeError("num_iter must be int.")
            if num_iter <= 0:
                raise ValueError("num_iter must be positive.")
            self._num_iter = num_iter
            self._polar_dec = PolarBPDecoder(enc_polar.frozen_pos, self._n_polar, num_iter=num_iter, hard_out=True,
                                             precision=precision)
        else:
            raise ValueError("Unknown value for dec_type.")
        self._dec_crc = CRCDecoder(enc_polar.enc_crc, precision=precision) if return_crc_status else None
        # rate-recovery gather: dematched[j] = sum of llr[src] over the received positions mapped to j
        n, npol = self._n_target, self._n_polar
        ch_inv = np.argsort(_channel_pattern(n)) if self._bil else np.arange(n)
        sub_inv = np.argsort(_subblock_pattern(npol))
        self._fill = 0.0
        if n >= npol:                       # repetition: positions 0..n_rep-1 are received twice
            # (n > 2 n_polar: like the reference, whose concat / gather keeps llr[j] + llr[n_polar + j] for every j
            # and drops further repetitions, decoding.py:2027-2034 + 2052)
            n_rep = n - npol
            a = np.arange(npol)
            b = np.where(np.arange(npol) < n_rep, npol + np.arange(npol), -1)
        elif self._k_polar / n <= 7 / 16:   # puncturing: first n_polar-n positions unknown (LLR 0)
            a = np.concatenate([np.full(npol - n, -1), np.arange(n)])
            b = np.full(npol, -1)
        else:                               # shortening: last positions are known zeros (logit -llr_max)
            a = np.concatenate([np.arange(n), np.full(npol - n, -2)])
            b = np.full(npol, -1)
        a, b = a[sub_inv], b[sub_inv]
        self._src_a = np.where(a >= 0, ch_inv[np.clip(a, 0, n - 1)], a)
        self._src_b = np.where(b >= 0, ch_inv[np.clip(b, 0, n - 1)], b)
        self._dev = None

    k_target = property(lambda self: self._k_target)
    n_target = property(lambda self: self._n_target)
    k_polar = property(lambda self: self._k_polar)
    n_polar = property(lambda self: self._n_polar)
    frozen_pos = property(lambda self: self._enc_polar.frozen_pos)
    info_pos = property(lambda self: self._enc_polar.info_pos)
    llr_max = property(lambda self: self._llr_max)
    dec_type = property(lambda self: self._dec_type)
    polar_dec = property(lambda self: self._polar_dec)

    def build(self, input_shape):
        if input_shape[-1] != self._n_target:
            raise ValueError("Invalid input shape.")

    def call(self, llr_ch):
        llr = _ffi.to_device(llr_ch, self.rdtype)
        if llr.shape[-1] != self._n_target:
            raise ValueError("Invalid input shape.")
        lead = tuple(llr.shape[:-1])
        x = llr.reshape(-1, self._n_target)
        if self._dev is None or self._dev_gen != _ffi.options_generation():   # (a development switch changed: rebuild)
            self._dev_gen = _ffi.options_generation()
            dev = x.device
            t = lambda a: torch.from_numpy(np.asarray(a, np.int64)).to(dev)
            self._dev = (t(np.clip(self._src_a, 0, None)), torch.from_numpy(self._src_a >= 0).to(dev),
                         torch.from_numpy(self._src_a == -2).to(dev), t(np.clip(self._src_b, 0, None)),
                         torch.from_numpy(self._src_b >= 0).to(dev), bool((self._src_b >= 0).any()))
        ia, ma, sa, ib, mb, has_b = self._dev
        pd = self._polar_dec
        if isinstance(pd, _PolarListDecoderBase) and not getattr(pd, "_use_hybrid_sc", False) and pd._rm_in_kernel() and x.shape[0] > 0:
            # the rate recovery (decoding.py:2018-2052) as an index inside the decoder's channel-LLR load: no gather launches
            if getattr(self, "_rm_dev", None) is None or self._rm_gen != _ffi.options_generation():
                self._rm_gen = _ffi.options_generation()
                i32 = lambda a_: _ffi.to_device(np.ascontiguousarray(a_, np.int32), torch.int32)
                self._rm_dev = (i32(self._src_a), i32(self._src_b) if has_b else None)
            u_crc, _ = pd._decode_2d(x.contiguous(), False, rm=(self._rm_dev[0], self._rm_dev[1], self._llr_max))
        else:
            # index plumbing of the rate recovery (gathers only)
            dec_in = torch.where(ma, x.index_select(1, ia), torch.zeros((), dtype=x.dtype, device=x.device))
            dec_in = torch.where(sa, torch.full((), -self._llr_max, dtype=x.dtype, device=x.device), dec_in)
            if has_b:
                dec_in = dec_in + torch.where(mb, x.index_select(1, ib), torch.zeros((), dtype=x.dtype, device=x.device))
            u_crc = self._polar_dec(dec_in.contiguous()).as_subclass(torch.Tensor)
        if self._iil:
            u_crc = u_crc.index_select(1, torch.from_numpy(self._ind_iil_inv.astype(np.int64)).to(u_crc.device))
        if self._return_crc_status:
            u_hat, status = self._dec_crc(u_crc.contiguous())
            return u_hat.reshape(lead + (self._k_target,)), status.reshape(lead)
        return u_crc[:, :-self._k_crc].reshape(lead + (self._k_target,))
