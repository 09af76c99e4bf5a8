"""LDPC belief-propagation decoders - host-side mirror of
``sionna.phy.fec.ldpc.LDPCBPDecoder`` / ``LDPC5GDecoder``
(reference src/sionna/phy/fec/ldpc/decoding.py:13-637, 1169-1536).

HIP engines behind the same Block API (the library picks per code and rule, ``samd_ldpc5g_decode_engine``):

* generic (any parity-check matrix, all check-node rules, array schedules, IDD state in / out):
  ``samd_ldpc_bp_decode_f32`` - messages HBM-resident, batch-last, two fused passes per iteration
  (csrc/ldpc_bp_generic.hip);
* 5G on-chip, flooding: ``samd_ldpc5g_decode_f32`` - rate recovery + all iterations + output mapping in ONE kernel
  with the messages in LDS (explicit messages csrc/ldpc5g_onchip_ms.inc, compressed check-node state
  csrc/ldpc5g_onchip.hip, part of the messages in L2 csrc/ldpc5g_onchip_mss.hip; min-sum family and boxplus rules);
* 5G on-chip, ``cn_schedule="layered"``: ``samd_ldpc5g_decode_layered_f32`` (csrc/ldpc5g_onchip_ly.hip).

All use the arithmetic and summation order of oracle/ldpc_bp.py / oracle/ldpc_bp.c, which tests/test_oracle_ref_exec.py
pins to the reference's own decoding.py executed under a NumPy stand-in for TensorFlow (min-sum family and VN update
bit for bit).  Python callables for node updates (``cn_update=fn`` / ``vn_update=fn``) and the message callbacks
(``v2c_callbacks`` / ``c2v_callbacks``) run on the device through the torch engine of ``custom.py`` (ragged messages
as padded device tensors): same interface as the reference's, slower than the built-in rules, still no CPU fallback.
"""
import ctypes as C
import types
import weakref

import numpy as np
import scipy.sparse as sp
import torch

from .... import _ffi
from ...block import Block, wrap, defer, Pending, pending_of
from .encoding import LDPC5GEncoder

# workspace cap of one generic-decoder launch; larger batches are processed in slices
_CACHE_SLICE_BYTES = 192 << 20
_MAX_WORKSPACE_BYTES = 48 << 30


class LDPCBPDecoder(Block):
    # pylint: disable=line-too-long
    """``LDPCBPDecoder(pcm, cn_update="boxplus-phi", vn_update="sum", cn_schedule="flooding",
    hard_out=True, num_iter=20, llr_max=20., v2c_callbacks=None, c2v_callbacks=None,
    return_state=False, precision=None)`` (reference decoding.py:175-345).

    ``call(llr_ch, /, *, num_iter=None, msg_v2c=None)``: ``llr_ch`` [..., n] logits;
    returns hard bits / soft logits [..., n] and, with ``return_state``, the v2c messages
    [num_edges, batch] (edge order: VN-major, ascending CN inside a VN).
    """

    def __init__(self, pcm, cn_update="boxplus-phi", vn_update="sum", cn_schedule="flooding",
                 hard_out=True, num_iter=20, llr_max=20., v2c_callbacks=None, c2v_callbacks=None,
                 return_state=False, precision=None, **kwargs):
        super().__init__(precision=precision, **kwargs)
        if not isinstance(hard_out, bool):
            raise TypeError("hard_out must be bool.")
        if not isinstance(num_iter, int):
            raise TypeError("num_iter must be int.")
        if num_iter < 0:
            raise ValueError("num_iter cannot be negative.")
        if not isinstance(return_state, bool):
            raise TypeError("return_state must be bool.")
        if isinstance(pcm, np.ndarray):
            if not np.array_equal(pcm, pcm.astype(bool)):
                raise ValueError("PC matrix must be binary.")
        elif isinstance(pcm, (sp.csr_matrix, sp.csc_matrix)):
            if not np.array_equal(pcm.data, pcm.data.astype(bool)):
                raise ValueError("PC matrix must be binary.")
        else:
            raise TypeError("Unsupported dtype of pcm.")
        if "cn_type" in kwargs:
            raise TypeError("'cn_type' is deprecated; use 'cn_update' instead.")
        if not isinstance(llr_max, (int, float)):
            raise TypeError("llr_max must be int or float.")

        self._pcm = pcm
        self._hard_out = hard_out
        self._num_iter = num_iter
        self._return_state = return_state
        self._num_cns, self._num_vns = pcm.shape
        self._llr_max = float(llr_max)

        cb_lists = {}
        for name, cbs in (("v2c_callbacks", v2c_callbacks), ("c2v_callbacks", c2v_callbacks)):
            if cbs is None:
                cb_lists[name] = []
            elif isinstance(cbs, (list, tuple)):
                cb_lists[name] = list(cbs)
            elif callable(cbs):
                cb_lists[name] = [cbs]                 # a single callable is accepted (decoding.py:236-239)
            else:
                raise TypeError(f"{name} must be a list of callables.")
        self._v2c_callbacks, self._c2v_callbacks = cb_lists["v2c_callbacks"], cb_lists["c2v_callbacks"]

        if isinstance(cn_schedule, str) and cn_schedule == "flooding":
            self._scheduling = "flooding"
        elif isinstance(cn_schedule, (np.ndarray, torch.Tensor)):
            sched = np.asarray(cn_schedule.cpu() if isinstance(cn_schedule, torch.Tensor) else cn_schedule)
            if sched.ndim != 2:
                raise ValueError("cn_schedule must be of rank 2.")
            if sched.max() >= self._num_cns:
                raise ValueError("cn_schedule can only contain values smaller number_cns.")
            if sched.min() < 0:
                raise ValueError("cn_schedule cannot contain negative values.")
            for row in sched:
                if len(np.unique(row)) != len(row):
                    raise ValueError("cn_schedule rows must not hold a check node twice.")
            self._scheduling = "custom"
            self._cn_schedule = np.ascontiguousarray(sched, dtype=np.int32)
        else:
            raise ValueError("cn_schedule can be 'flooding' or an array of ints.")

        # node updates (decoding.py:291-322): the string rules - and the exported functions that implement them -
        # run on the HIP engines; any other callable, "identity" and message callbacks run on the torch engine of
        # custom.py (same loop, device tensors, differentiable)
        from . import custom
        if isinstance(cn_update, types.FunctionType) and cn_update in custom.FUNCTION_TO_RULE:
            cn_update = custom.FUNCTION_TO_RULE[cn_update]
        if vn_update is custom.vn_update_sum:
            vn_update = "sum"
        self._cn_mode = None
        if isinstance(cn_update, str) and cn_update in _ffi.CN_MODES:
            self._cn_mode = _ffi.CN_MODES[cn_update]
            cn_fn = custom.BUILTIN_CN[cn_update]
        elif cn_update == "identity":
            cn_fn = custom.cn_node_update_identity
        elif isinstance(cn_update, types.FunctionType):
            cn_fn = cn_update
        else:
            raise TypeError("Provided cn_update not supported.")
        self._cn_update_name = cn_update
        if isinstance(vn_update, str) and vn_update in custom.BUILTIN_VN:
            vn_fn = custom.BUILTIN_VN[vn_update]
        elif isinstance(vn_update, types.FunctionType):
            vn_fn = vn_update
        else:
            raise TypeError("Provided vn_update not supported.")
        self._custom = (self._cn_mode is None or vn_update != "sum" or bool(self._v2c_callbacks)
                        or bool(self._c2v_callbacks))
        self._custom_fns = (cn_fn, vn_fn)
        self._custom_engine = None
        self._offset = 0.5                       # reference default of cn_update_offset_minsum

        # graph: VN-major edge list, ascending CN inside a VN (decoding.py:277-292, stable order)
        coo = sp.coo_matrix(pcm)
        nz = coo.data != 0
        cn_idx, vn_idx = coo.row[nz].astype(np.int64), coo.col[nz].astype(np.int64)
        order = np.lexsort((cn_idx, vn_idx))
        self._cn_idx = np.ascontiguousarray(cn_idx[order], dtype=np.int32)
        self._vn_idx = np.ascontiguousarray(vn_idx[order], dtype=np.int32)
        self._num_edges = len(self._vn_idx)
        self._graph = None
        self._sched = None
        self._ws = _ffi.Workspace()

    # ------------------------------------------------------------ properties (decoding.py:351-410)
    pcm = property(lambda self: self._pcm)
    num_cns = property(lambda self: self._num_cns)
    num_vns = property(lambda self: self._num_vns)
    n = property(lambda self: self._num_vns)
    num_edges = property(lambda self: self._num_edges)
    return_state = property(lambda self: self._return_state)

    @property
    def coderate(self):
        return (self._num_vns - self._num_cns) / self._num_vns

    @property
    def num_iter(self):
        return self._num_iter

    @num_iter.setter
    def num_iter(self, num_iter):
        if not isinstance(num_iter, int):
            raise TypeError("num_iter must be int.")
        if num_iter < 0:
            raise ValueError("num_iter cannot be negative.")
        self._num_iter = num_iter

    @property
    def llr_max(self):
        return self._llr_max

    @llr_max.setter
    def llr_max(self, value):
        if value < 0:
            raise ValueError("llr_max cannot be negative.")
        self._llr_max = float(value)

    # ------------------------------------------------------------ engine
    def _graph_handle(self):
        if self._graph is None:
            h = C.c_void_p()
            _ffi.device()
            _ffi.check(_ffi.lib().samd_ldpc_graph_create(
                self._cn_idx.ctypes.data_as(C.c_void_p), self._vn_idx.ctypes.data_as(C.c_void_p),
                self._num_edges, self._num_cns, self._num_vns, C.byref(h)), "samd_ldpc_graph_create")
            self._graph = h
        return self._graph

    def _schedule_handle(self):
        if self._sched is None:
            h = C.c_void_p()
            sc = self._cn_schedule
            _ffi.check(_ffi.lib().samd_ldpc_schedule_create(self._graph_handle(), sc.ctypes.data_as(C.c_void_p),
                                                            sc.shape[0], sc.shape[1], C.byref(h)),
                       "samd_ldpc_schedule_create")
            self._sched = h
        return self._sched

    def __del__(self):
        try:
            if self._sched is not None:
                _ffi.lib().samd_ldpc_schedule_destroy(self._sched)
            if self._graph is not None:
                _ffi.lib().samd_ldpc_graph_destroy(self._graph)
        except Exception:  # pylint: disable=broad-except
            pass

    def _decode_2d(self, llr, out_cols, num_iter, msg_v2c, hard_out=None):
        """llr [B, N_vn] device float32 -> (x_hat [B, out_cols], state or None)."""
        hard = self._hard_out if hard_out is None else hard_out
        batch = llr.shape[0]
        if self._custom:
            return self._decode_custom(llr, out_cols, num_iter, msg_v2c, hard)
        if self.precision == "double":
            return self._decode_2d_f64(llr, out_cols, num_iter, msg_v2c, hard)
        lib, g = _ffi.lib(), self._graph_handle()
        out = torch.empty((batch, out_cols), dtype=torch.float32, device=llr.device)
        want_state = self._return_state
        state = None
        if msg_v2c is not None:
            state = _ffi.to_device(msg_v2c, torch.float32)
            if tuple(state.shape) != (self._num_edges, batch):
                raise ValueError("msg_v2c must have shape [num_edges, batch_size]")
            state = state.clone() if want_state else state
        elif want_state:
            state = torch.empty((self._num_edges, batch), dtype=torch.float32, device=llr.device)
        if batch == 0:
            return out, state
        per_cw = lib.samd_ldpc_bp_workspace_bytes(g, 64) // 64
        step = batch
        if state is not None and per_cw * batch > _MAX_WORKSPACE_BYTES:
            # the [num_edges, batch] state is batch-last, so a batch slice is not a contiguous view: one launch,
            # whose workspace ((E + 2 N) * 4 bytes per codeword) has to fit
            raise ValueError(f"return_state / msg_v2c: the message workspace of {batch} codewords "
                             f"({per_cw * batch / 2 ** 30:.1f} GiB) exceeds {_MAX_WORKSPACE_BYTES >> 30} GiB; "
                             f"decode at most {_MAX_WORKSPACE_BYTES // per_cw} codewords per call")
        if state is None and per_cw * batch > _CACHE_SLICE_BYTES:
            # slices whose whole message state stays in the 256 MiB Infinity Cache: measured +20 %
            # (min-sum) / +4 % (phi) at config C2 over one 65536-codeword pass per launch
            step = max(256, (_CACHE_SLICE_BYTES // per_cw) // 64 * 64)
        for b0 in range(0, batch, step):
            nb = min(step, batch - b0)
            need = lib.samd_ldpc_bp_workspace_bytes(g, nb)
            ws, ws_bytes = self._ws.get(need)
            args = (_ffi.ptr(llr[b0:b0 + nb]), _ffi.ptr(out[b0:b0 + nb]), out_cols, _ffi.ptr(state),
                    int(msg_v2c is not None), int(want_state), nb, int(num_iter), self._cn_mode,
                    self._llr_max, self._offset, int(bool(hard)), _ffi.ptr(ws), ws_bytes, _ffi.stream())
            if self._scheduling == "flooding":
                _ffi.check(lib.samd_ldpc_bp_decode_f32(g, *args), "LDPCBPDecoder")
            else:
                _ffi.check(lib.samd_ldpc_bp_decode_scheduled_f32(g, self._schedule_handle(), *args),
                           "LDPCBPDecoder(scheduled)")
        return out, (state if want_state else None)

    def _decode_2d_f64(self, llr, out_cols, num_iter, msg_v2c, hard):
        """precision="double": the float64 engine samd_ldpc_bp_decode_f64 (csrc/f64.hip), flooding and schedules."""
        lib, g = _ffi.lib(), self._graph_handle()
        sched = self._schedule_handle() if self._scheduling != "flooding" else None
        batch = llr.shape[0]
        out = torch.empty((batch, out_cols), dtype=torch.float64, device=llr.device)
        state = None
        if msg_v2c is not None:
            state = _ffi.to_device(msg_v2c, torch.float64)
            if tuple(state.shape) != (self._num_edges, batch):
                raise ValueError("msg_v2c must have shape [num_edges, batch_size]")
            state = state.clone() if self._return_state else state
        elif self._return_state:
            state = torch.empty((self._num_edges, batch), dtype=torch.float64, device=llr.device)
        if batch == 0:
            return out, state
        ws, ws_bytes = self._ws.get(lib.samd_ldpc_bp_workspace_bytes_f64(g, batch))
        _ffi.check(lib.samd_ldpc_bp_decode_f64(g, sched, _ffi.ptr(llr), _ffi.ptr(out), out_cols, _ffi.ptr(state),
                                               int(msg_v2c is not None), int(self._return_state), batch, int(num_iter),
                                               1 if self._cn_mode == 4 else self._cn_mode,   # float64 has one phi form
                                               self._llr_max, self._offset, int(bool(hard)), _ffi.ptr(ws),
                                               ws_bytes, _ffi.stream()), "LDPCBPDecoder(double)")
        return out, (state if self._return_state else None)

    def _decode_custom(self, llr, out_cols, num_iter, msg_v2c, hard):
        """Custom node updates / message callbacks: the torch engine of custom.py on the device tensors."""
        from . import custom
        if self._custom_engine is None:
            sched = self._cn_schedule if self._scheduling == "custom" else None
            self._custom_engine = custom.CustomBPEngine(self._cn_idx, self._vn_idx, self._num_cns, self._num_vns,
                                                        self._custom_fns[0], self._custom_fns[1], self._c2v_callbacks,
                                                        self._v2c_callbacks, sched)
        state_in = None
        if msg_v2c is not None:
            state_in = _ffi.to_device(msg_v2c, llr.dtype)
            if tuple(state_in.shape) != (self._num_edges, llr.shape[0]):
                raise ValueError("msg_v2c must have shape [num_edges, batch_size]")
        x_hat, v2c = self._custom_engine.decode(llr, num_iter, self._llr_max, state_in)
        x_hat = x_hat[:, :out_cols]
        out = (0 >= x_hat).to(llr.dtype) if hard else -1.0 * x_hat              # decoding.py:620-626
        return out.contiguous(), ((-1.0 * v2c).contiguous() if self._return_state else None)

    # ------------------------------------------------------------ Block interface
    def build(self, input_shape, **kwargs):
        assert input_shape[-1] == self._num_vns, "Last dimension must be of length n."

    def call(self, llr_ch, /, *, num_iter=None, msg_v2c=None):
        if num_iter is None:
            num_iter = self._num_iter
        llr_ch = _ffi.to_device(llr_ch, self.rdtype)            # float32, or float64 for precision="double"
        assert llr_ch.shape[-1] == self._num_vns, "Last dimension must be of length n."
        shape = tuple(llr_ch.shape)
        x, state = self._decode_2d(llr_ch.reshape(-1, self._num_vns), self._num_vns, num_iter, msg_v2c)
        x = x.reshape(shape)
        return (x, state) if self._return_state else x


class LDPC5GDecoder(LDPCBPDecoder):
    # pylint: disable=line-too-long
    """``LDPC5GDecoder(encoder, cn_update="boxplus-phi", vn_update="sum", cn_schedule="flooding",
    hard_out=True, return_infobits=True, num_iter=20, llr_max=20., v2c_callbacks=None,
    c2v_callbacks=None, prune_pcm=True, return_state=False, precision=None)``
    (reference decoding.py:1302-1403); ``call(llr_ch[..., n])`` -> [..., k] or [..., n]."""

    def __init__(self, encoder, cn_update="boxplus-phi", vn_update="sum", cn_schedule="flooding",
                 hard_out=True, return_infobits=True, num_iter=20, llr_max=20., v2c_callbacks=None,
                 c2v_callbacks=None, prune_pcm=True, return_state=False, precision=None, **kwargs):
        if not isinstance(encoder, LDPC5GEncoder):
            raise TypeError("encoder must be of class LDPC5GEncoder.")
        self._encoder = encoder
        if not isinstance(return_infobits, bool):
            raise TypeError("return_info must be bool.")
        self._return_infobits = return_infobits
        if not isinstance(return_state, bool):
            raise TypeError("return_state must be bool.")
        if "cn_type" in kwargs:
            raise TypeError("'cn_type' is deprecated; use 'cn_update' instead.")
        if not isinstance(prune_pcm, bool):
            raise TypeError("prune_pcm must be bool.")
        self._prune_pcm = prune_pcm
        pcm = encoder.pcm
        k_filler = encoder.k_ldpc - encoder.k
        nb_punc_bits = (encoder.n_ldpc - k_filler) - encoder.n - 2 * encoder.z
        if prune_pcm:
            # trailing degree-1 VNs that are punctured never contribute (decoding.py:1337-1378)
            dv = np.diff(sp.csc_matrix(pcm).indptr)
            tail = np.nonzero(dv[::-1] != 1)[0]
            last_pos = encoder.n_ldpc - (int(tail[0]) if len(tail) else encoder.n_ldpc - 1)
            last_pos = max(last_pos, 1)
            if isinstance(cn_schedule, str) and cn_schedule == "layered":
                nb_punc_bits = int(np.floor(nb_punc_bits / encoder.z) * encoder.z)
            self._n_pruned = int(max(last_pos, encoder.n_ldpc - nb_punc_bits))
            self._nb_pruned_nodes = encoder.n_ldpc - self._n_pruned
            if self._nb_pruned_nodes < 0:
                raise ArithmeticError("Internal error: number of pruned nodes must be positive.")
            if self._nb_pruned_nodes > 0:
                pcm = pcm[:-self._nb_pruned_nodes, :-self._nb_pruned_nodes]
        else:
            self._nb_pruned_nodes = 0
            self._n_pruned = encoder.n_ldpc
        self._layered5g = isinstance(cn_schedule, str) and cn_schedule == "layered"
        if isinstance(cn_schedule, str) and cn_schedule == "layered":       # decoding.py:1383-1389
            z = encoder.z
            cn_schedule = np.stack([np.arange(z) + i * z for i in range(pcm.shape[0] // z)], axis=0)
        super().__init__(sp.csr_matrix(pcm), cn_update=cn_update, vn_update=vn_update,
                         cn_schedule=cn_schedule, hard_out=hard_out, num_iter=num_iter, llr_max=llr_max,
                         v2c_callbacks=v2c_callbacks, c2v_callbacks=c2v_callbacks,
                         return_state=return_state, precision=precision, **kwargs)
        self._onchip_ok = True

    encoder = property(lambda self: self._encoder)

    def build(self, input_shape, **kwargs):
        if input_shape[-1] != self._encoder.n:
            raise ValueError("Last dimension must be of length n.")

    def _try_onchip(self, llr2d, num_iter):
        """Whole decode in one kernel: min-sum family for every 5G code (compressed check-node state), boxplus /
        boxplus-phi when the per-edge messages fit in LDS (n=8448 rate 1/3 and everything smaller)."""
        enc = self._encoder
        out_cols = enc.k if self._return_infobits else enc.n
        out = torch.empty((llr2d.shape[0], out_cols), dtype=torch.float32, device=llr2d.device)
        lib, h = _ffi.lib(), enc._handle(self._nb_pruned_nodes)
        need = lib.samd_ldpc5g_decode_workspace_bytes(h, llr2d.shape[0], self._cn_mode)   # 0 unless the LLRs live in L2
        ws, ws_bytes = self._ws.get(need) if need else (None, 0)
        rc = lib.samd_ldpc5g_decode_f32(
            h, _ffi.ptr(llr2d), _ffi.ptr(out), llr2d.shape[0], int(num_iter),
            self._cn_mode, self._llr_max, self._offset, int(self._hard_out), int(self._return_infobits),
            _ffi.ptr(ws), ws_bytes, _ffi.stream())
        if rc == _ffi.ERR_UNSUPPORTED:
            self._onchip_ok = False
            return None
        _ffi.check(rc, "LDPC5GDecoder(on-chip)")
        return out

    _defer_state = True      # return_state on the generated kernels: the [num_edges, batch] tensor is formed on first use
    _state_last = None
    _state_kept = None

    def _state_layout(self):
        """(img_floats, cw_per_pass, device table) of the generated kernel's message image for this code and rule, or None:
        samd_ldpc5g_state_layout / samd_ldpc5g_state_map (include/sionna_amd.h) sorted into the reference's edge order
        (edges by variable node, then check node: decoding.py:282-288 = self._cn_idx / self._vn_idx)."""
        key = (self._cn_mode, _ffi.options_generation())
        if getattr(self, "_state_lay", (None,))[0] == key:
            return self._state_lay[1]
        lib, h = _ffi.lib(), self._encoder._handle(self._nb_pruned_nodes)
        img, cwpp = C.c_int(), C.c_int()
        lay = None
        if lib.samd_ldpc5g_state_layout(h, self._cn_mode, C.byref(img), C.byref(cwpp)) == 0:
            img, cwpp = img.value, cwpp.value
            cw, cn, vn = (np.empty(img, np.int32) for _ in range(3))
            _ffi.check(lib.samd_ldpc5g_state_map(h, self._cn_mode, cw.ctypes.data_as(C.c_void_p), cn.ctypes.data_as(C.c_void_p),
                                                 vn.ctypes.data_as(C.c_void_p)), "samd_ldpc5g_state_map")
            key_e = self._vn_idx.astype(np.int64) * self._num_cns + self._cn_idx           # ascending: the reference's order
            live = cw >= 0
            want = vn[live].astype(np.int64) * self._num_cns + cn[live]
            e = np.searchsorted(key_e, want)
            if (cwpp < 128 and self._num_edges < (1 << 24) and int(live.sum()) == cwpp * self._num_edges
                    and np.array_equal(key_e[np.minimum(e, self._num_edges - 1)], want)):
                table = np.full(img, -1, np.int32)
                table[live] = (cw[live].astype(np.int64) << 24 | e).astype(np.int32)
                lay = (img, cwpp, torch.from_numpy(table).to(_ffi.device()))
        self._state_lay = (key, lay)
        return lay

    def _try_onchip_state(self, llr2d, num_iter, msg_v2c):
        """return_state / msg_v2c on the generated kernel (decoding.py:573, 636-637): the kernel keeps the messages on chip for
        the whole call and moves their image in and out once; the [num_edges, batch] tensor of the reference's interface is
        formed from the image by one more launch - and NOT re-read when the caller hands back the very tensor a previous call
        returned, unmodified (the iterative detection-and-decoding loop): its image is still here.  None: not covered."""
        lay = self._state_layout()
        if lay is None or num_iter < 1:
            return None
        img, cwpp, table = lay
        enc, lib = self._encoder, _ffi.lib()
        batch = llr2d.shape[0]
        passes = -(-batch // cwpp)
        image_in = None
        if msg_v2c is not None:
            if not isinstance(msg_v2c, torch.Tensor):
                msg_v2c = _ffi.to_device(msg_v2c, torch.float32)
            if tuple(msg_v2c.shape) != (self._num_edges, batch):
                raise ValueError("msg_v2c must have shape [num_edges, batch_size]")
            pend, kept = pending_of(msg_v2c), getattr(self, "_state_kept", None)
            if pend is not None and pend.kind == "ldpc5g_state" and pend.layout is table and pend.image.shape[0] == passes:
                image_in = pend.image                   # a state this decoder returned and nobody has looked at: never formed
            elif kept is not None and kept[0]() is msg_v2c and kept[1] == msg_v2c._version and kept[2].shape[0] == passes:
                image_in = kept[2]                      # formed, but not modified since
            else:
                state = _ffi.to_device(msg_v2c, torch.float32)
                image_in = torch.empty((passes, img), dtype=torch.float32, device=llr2d.device)
                _ffi.check(lib.samd_ldpc5g_state_convert_f32(_ffi.ptr(table), img, cwpp, batch, _ffi.ptr(image_in), _ffi.ptr(state),
                                                             0, _ffi.stream()), "state_convert")
        image_out = torch.empty((passes, img), dtype=torch.float32, device=llr2d.device) if self._return_state else None
        out_cols = enc.k if self._return_infobits else enc.n
        out = torch.empty((batch, out_cols), dtype=torch.float32, device=llr2d.device)
        rc = lib.samd_ldpc5g_decode_state_f32(
            enc._handle(self._nb_pruned_nodes), _ffi.ptr(llr2d), _ffi.ptr(out), _ffi.ptr(image_in), _ffi.ptr(image_out), batch,
            int(num_iter), self._cn_mode, self._llr_max, self._offset, int(self._hard_out), int(self._return_infobits), _ffi.stream())
        if rc == _ffi.ERR_UNSUPPORTED:
            self._state_lay = (self._state_lay[0], None)
            return None
        _ffi.check(rc, "LDPC5GDecoder(on-chip, state)")
        state_out = None
        if self._return_state:
            # the reference's [num_edges, batch] tensor is DEFERRED (phy/block.py): one launch forms it from the image when somebody
            # reads it; handed back to this decoder untouched, it never exists
            state_out = torch.empty((self._num_edges, batch), dtype=torch.float32, device=llr2d.device)
            dec_ref = weakref.ref(self)

            def fill(plain, image=image_out, table=table, img=img, cwpp=cwpp, batch=batch):
                _ffi.check(_ffi.lib().samd_ldpc5g_state_convert_f32(_ffi.ptr(table), img, cwpp, batch, _ffi.ptr(image), _ffi.ptr(plain),
                                                                    1, _ffi.stream()), "state_convert")
                d = dec_ref()
                if d is not None and d._state_last is not None and d._state_last[0]() is not None:
                    t = d._state_last[0]()
                    if t.data_ptr() == plain.data_ptr():
                        d._state_kept = (d._state_last[0], t._version, image)
            if self._defer_state:
                state_out = defer(state_out, Pending("ldpc5g_state", fill, image=image_out, layout=table))
                self._state_last = (weakref.ref(state_out),)
            else:
                state_out = wrap(state_out)
                self._state_last = (weakref.ref(state_out),)
                fill(state_out.as_subclass(torch.Tensor))
        return out, state_out

    def _try_onchip_layered(self, llr2d, num_iter):
        """Whole layered decode in one kernel (csrc/ldpc5g_onchip_ly.hip); None when the code is not covered."""
        enc = self._encoder
        lib, h = _ffi.lib(), enc._handle(self._nb_pruned_nodes)
        if not lib.samd_ldpc5g_decode_layered_supported(h, self._cn_mode):
            return None
        out_cols = enc.k if self._return_infobits else enc.n
        out = torch.empty((llr2d.shape[0], out_cols), dtype=torch.float32, device=llr2d.device)
        ws, ws_bytes = self._ws.get(lib.samd_ldpc5g_decode_layered_workspace_bytes(h, llr2d.shape[0]))
        rc = lib.samd_ldpc5g_decode_layered_f32(
            h, _ffi.ptr(llr2d), _ffi.ptr(out), llr2d.shape[0], int(num_iter), self._cn_mode, self._llr_max, self._offset,
            int(self._hard_out), int(self._return_infobits), _ffi.ptr(ws), ws_bytes, _ffi.stream())
        if rc == _ffi.ERR_UNSUPPORTED:
            return None
        _ffi.check(rc, "LDPC5GDecoder(on-chip layered)")
        return out

    def call(self, llr_ch, /, *, num_iter=None, msg_v2c=None):
        enc = self._encoder
        if num_iter is None:
            num_iter = self._num_iter
        double = self.precision == "double"
        llr_ch = _ffi.to_device(llr_ch, self.rdtype)
        if llr_ch.shape[-1] != enc.n:
            raise ValueError("Last dimension must be of length n.")
        shape = tuple(llr_ch.shape)
        llr2d = llr_ch.reshape(-1, enc.n)
        batch = llr2d.shape[0]
        out_shape = shape[:-1] + ((enc.k,) if self._return_infobits else (enc.n,))

        use_onchip = (self._onchip_ok and not double and not self._custom and self._cn_mode in (0, 1, 2, 3, 4) and not self._return_state
                      and msg_v2c is None and batch > 0 and self._scheduling == "flooding")
        if use_onchip:
            out = self._try_onchip(llr2d, num_iter)
            if out is not None:
                return out.reshape(out_shape)
        if (self._onchip_ok and not double and not self._custom and self._cn_mode in (0, 1, 2, 3, 4) and batch > 0
                and self._scheduling == "flooding" and (self._return_state or msg_v2c is not None)):
            res = self._try_onchip_state(llr2d, int(num_iter), msg_v2c)
            if res is not None:
                return (res[0].reshape(out_shape), res[1]) if self._return_state else res[0].reshape(out_shape)
        # cn_schedule="layered" (one sub-iteration per base row): the on-chip layered engine when the code is covered
        if (self._layered5g and self._onchip_ok and not double and not self._custom and self._cn_mode in (1, 2, 3, 4)
                and not self._return_state and msg_v2c is None and batch > 0):
            out = self._try_onchip_layered(llr2d, num_iter)
            if out is not None:
                return out.reshape(out_shape)

        # generic engine: rate recovery -> BP -> output mapping (decoding.py:1431-1536)
        h = enc._handle(self._nb_pruned_nodes)
        lib = _ffi.lib()
        recover = lib.samd_ldpc5g_rate_recover_f64 if double else lib.samd_ldpc5g_rate_recover_f32
        extract = lib.samd_ldpc5g_extract_codeword_f64 if double else lib.samd_ldpc5g_extract_codeword_f32
        llr_5g = torch.empty((batch, self._num_vns), dtype=llr2d.dtype, device=llr2d.device)
        if batch > 0:
            _ffi.check(recover(h, _ffi.ptr(llr2d), _ffi.ptr(llr_5g), batch, self._llr_max, _ffi.stream()), "rate_recover")
        if self._return_infobits:
            x, state = self._decode_2d(llr_5g, enc.k, num_iter, msg_v2c)
            res = x.reshape(out_shape)
        else:
            x, state = self._decode_2d(llr_5g, self._num_vns, num_iter, msg_v2c)
            res = torch.empty((batch, enc.n), dtype=llr2d.dtype, device=llr2d.device)
            if batch > 0:
                _ffi.check(extract(h, _ffi.ptr(x), _ffi.ptr(res), batch, _ffi.stream()), "extract_codeword")
            res = res.reshape(out_shape)
        return (res, state) if self._return_state else res
