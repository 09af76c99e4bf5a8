"""Oracle (NumPy, CPU): flooding belief-propagation LDPC decoder + 5G rate recovery.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Restates ``fec/ldpc/decoding.py`` of
the reference op for op, on batch-LAST ``[num_edges, batch]`` message tensors exactly
like the reference (decoding.py:561-562):

* graph / edge order         decoding.py:277-345   (edges VN-major; we fix the reference's
                                                    unspecified argsort kind to "stable":
                                                    ascending CN inside a VN, ascending VN
                                                    inside a CN  - SURVEY Appendix A.8)
* main loop                  decoding.py:416-524, 544-637
* vn_update_sum              decoding.py:681-732
* cn_update_offset_minsum    decoding.py:755-909   (incl. 1e5 sentinel and "double_min")
* cn_update_minsum           decoding.py:911-953
* cn_update_tanh             decoding.py:955-1043
* cn_update_phi              decoding.py:1045-1166
* LDPC5GDecoder              decoding.py:1302-1403 (pruning), 1427-1536 (rate recovery)

Ragged reductions are written as explicit sequential loops over the position inside a
node, so the floating-point summation order is DEFINED: edge order inside the node,
first edge first (TensorFlow leaves it unspecified).  The HIP kernels use the same
order, which is what makes min-sum parity bit-exact.
"""
import numpy as np
import scipy.sparse as sp

_F32 = np.float32


class _Ragged:
    """Row-partition helper: ``ids`` = sorted node id per entry (value_rowids)."""

    def __init__(self, ids, num_nodes):
        ids = np.asarray(ids, np.int64)
        assert np.all(np.diff(ids) >= 0)
        self.ids = ids
        self.num_nodes = num_nodes
        self.deg = np.bincount(ids, minlength=num_nodes)
        self.start = np.concatenate([[0], np.cumsum(self.deg)])[:-1]
        self.pos = np.arange(len(ids)) - self.start[ids]     # position inside the node
        self.max_deg = int(self.deg.max()) if len(ids) else 0
        # entries at position j of every node that has > j entries
        self.by_pos = [np.nonzero(self.pos == j)[0] for j in range(self.max_deg)]

    def reduce(self, flat, op, init):
        """Sequential reduction over the ragged axis -> [num_nodes, batch]."""
        out = np.full((self.num_nodes,) + flat.shape[1:], init, dtype=flat.dtype)
        for j in range(self.max_deg):
            e = self.by_pos[j]
            n = self.ids[e]
            if j == 0:
                out[n] = flat[e] if init is None else op(out[n], flat[e])
            else:
                out[n] = op(out[n], flat[e])
        return out


def _sum0(r, flat):
    """reduce_sum over a ragged row: ((0 + x0) + x1) + ...  (0 + x0 == x0 exactly)."""
    return r.reduce(flat, np.add, 0)


def _prod1(r, flat):
    return r.reduce(flat, np.multiply, 1)


def _min(r, flat):
    return r.reduce(flat, np.minimum, np.inf)


# --------------------------------------------------------------------------- node updates
def vn_update_sum(r, msg_c2v, llr_ch, llr_clipping=None):
    """decoding.py:681-732.  msg_c2v: [E,B] in VN-sorted edge order; llr_ch: [N_vn,B]."""
    x = _sum0(r, msg_c2v)
    x_tot = x + llr_ch
    x_e = -1. * msg_c2v + x_tot[r.ids]
    if llr_clipping is not None:
        x_e = np.clip(x_e, -llr_clipping, llr_clipping)
        x_tot = np.clip(x_tot, -llr_clipping, llr_clipping)
    return x_e.astype(msg_c2v.dtype), x_tot.astype(msg_c2v.dtype)


def _sign_no_zero(msg):
    s = np.sign(msg)
    return np.where(s == 0, np.ones_like(s), s)


def cn_update_offset_minsum(r, msg_v2c, llr_clipping=None, offset=0.5):
    """decoding.py:755-909.  msg_v2c: [E,B] in CN-sorted edge order."""
    dt = msg_v2c.dtype.type
    large_val = dt(100000.)
    msg = np.clip(msg_v2c, -large_val, large_val)
    sign_val = _sign_no_zero(msg)
    sign_node = _prod1(r, sign_val)
    sign_val = sign_val * sign_node[r.ids]
    msg = np.abs(msg)
    min_val = _min(r, msg)                                    # [N,B]
    msg_min1 = msg - min_val[r.ids]
    msg = np.where(msg_min1 == 0, large_val, msg_min1)
    min_val_2 = _min(r, msg) + min_val
    node_sum = _sum0(r, msg) - dt(2 * 100000. - 1.)
    double_min = dt(0.5) * (1 - np.sign(node_sum))
    min_val_e = (1 - double_min) * min_val + double_min * min_val_2
    min_1 = min_val[r.ids]
    min_e = min_val_e[r.ids]
    msg_e = np.where(msg == large_val, min_e, min_1)
    msg_e = np.maximum(msg_e - dt(offset), dt(0))
    out = sign_val * msg_e
    if llr_clipping is not None:
        out = np.clip(out, -llr_clipping, llr_clipping)
    return out.astype(msg_v2c.dtype)


def cn_update_minsum(r, msg_v2c, llr_clipping=None):
    """decoding.py:911-953"""
    return cn_update_offset_minsum(r, msg_v2c, llr_clipping, offset=0.)


def cn_update_tanh(r, msg, llr_clipping=None):
    """decoding.py:955-1043"""
    dt = msg.dtype.type
    atanh_clip_value = dt(1 - 1e-7)
    msg = msg / dt(2)
    msg = np.tanh(msg)
    msg = np.where(msg == 0, dt(1e-12), msg)
    msg_prod = _prod1(r, msg)
    msg = (msg ** -1) * msg_prod[r.ids]
    msg = np.where(np.abs(msg) < dt(1e-7), dt(0), msg)
    msg = np.clip(msg, -atanh_clip_value, atanh_clip_value)
    msg = dt(2) * np.arctanh(msg)
    if llr_clipping is not None:
        msg = np.clip(msg, -llr_clipping, llr_clipping)
    return msg


def _phi(x):
    """decoding.py:1092-1120 (literal form, dtype dependent clip)."""
    if x.dtype == np.float32:
        # float32: the defined exp / log (Cephes / Eigen restatement in oracle/ldpc_bp.c; NumPy has no fused
        # multiply-add, so the element-wise function is the C oracle's) - one bit-level definition for both oracles
        from . import cbind
        return cbind.phi_f32(x)
    elif x.dtype == np.float64:
        x = np.clip(x, 1e-12, 28.324079)
    else:
        raise TypeError("Unsupported dtype for phi function.")
    one = x.dtype.type(1)
    return np.log(np.exp(x) + one) - np.log(np.exp(x) - one)


def cn_update_phi(r, msg, llr_clipping=None):
    """decoding.py:1045-1166"""
    sign_val = _sign_no_zero(msg)
    sign_node = _prod1(r, sign_val)
    sign_val = sign_val * sign_node[r.ids]
    msg = np.abs(msg)
    msg = _phi(msg)
    msg_sum = _sum0(r, msg)
    msg = -1. * msg + msg_sum[r.ids]
    msg_e = sign_val * _phi(msg.astype(sign_val.dtype))
    if llr_clipping is not None:
        msg_e = np.clip(msg_e, -llr_clipping, llr_clipping)
    return msg_e


def cn_update_identity(r, msg, llr_clipping=None):
    return msg


def vn_update_identity(r, msg_c2v, llr_ch, llr_clipping=None):
    """decoding.py:644-679"""
    return msg_c2v, _sum0(r, msg_c2v) + llr_ch


_CN = {"boxplus": cn_update_tanh, "boxplus-phi": cn_update_phi, "minsum": cn_update_minsum,
       "min": cn_update_minsum, "offset-minsum": cn_update_offset_minsum,
       "identity": cn_update_identity}
_VN = {"sum": vn_update_sum, "identity": vn_update_identity}


# --------------------------------------------------------------------------- decoder
class LDPCBPDecoder:
    """decoding.py:13-637 (flooding or array CN schedule, built-in node updates, no callbacks)."""

    def __init__(self, pcm, cn_update="boxplus-phi", vn_update="sum", hard_out=True,
                 num_iter=20, llr_max=20., return_state=False, precision="single", cn_schedule="flooding"):
        if isinstance(pcm, np.ndarray):
            if not np.array_equal(pcm, pcm.astype(bool)):
                raise ValueError("PC matrix must be binary.")
            pcm = sp.csr_matrix(pcm)
        self.pcm = sp.csr_matrix(pcm)
        self.dtype = np.float32 if precision == "single" else np.float64
        self.num_cns, self.num_vns = self.pcm.shape
        coo = self.pcm.tocoo()
        cn_idx, vn_idx = coo.row.astype(np.int64), coo.col.astype(np.int64)
        nz = coo.data != 0
        cn_idx, vn_idx = cn_idx[nz], vn_idx[nz]
        # VN-major edge order, ascending CN inside a VN (decoding.py:282-288, stable)
        order = np.lexsort((cn_idx, vn_idx))
        self.cn_idx, self.vn_idx = cn_idx[order], vn_idx[order]
        self.num_edges = len(self.vn_idx)
        # CN view (decoding.py:329-345)
        self.v2c_perm = np.argsort(self.cn_idx, kind="stable")
        self.v2c_perm_inv = np.argsort(self.v2c_perm, kind="stable")
        self._cn_rag = _Ragged(self.cn_idx[self.v2c_perm], self.num_cns)
        self._vn_rag = _Ragged(self.vn_idx, self.num_vns)
        self._cn_update = _CN[cn_update]
        self._vn_update = _VN[vn_update]
        self.hard_out, self.num_iter, self.return_state = hard_out, num_iter, return_state
        self.llr_max = self.dtype(llr_max)
        # decoding.py:252-270
        if isinstance(cn_schedule, str) and cn_schedule == "flooding":
            self.cn_schedule = None
        elif isinstance(cn_schedule, np.ndarray):
            cn_schedule = cn_schedule.astype(np.int32)
            if cn_schedule.ndim != 2:
                raise ValueError("cn_schedule must be of rank 2.")
            if cn_schedule.max() >= self.num_cns:
                raise ValueError("cn_schedule can only contain values smaller number_cns.")
            if cn_schedule.min() < 0:
                raise ValueError("cn_schedule cannot contain negative values.")
            self.cn_schedule = cn_schedule
            cn_sorted = self.cn_idx[self.v2c_perm]
            starts = np.searchsorted(cn_sorted, np.arange(self.num_cns), "left")
            ends = np.searchsorted(cn_sorted, np.arange(self.num_cns), "right")
            self._sched = []
            for row in cn_schedule:
                pos = np.concatenate([np.arange(starts[c], ends[c]) for c in row])   # CN-view positions
                rid = np.concatenate([np.full(ends[c] - starts[c], i) for i, c in enumerate(row)])
                self._sched.append((pos, _Ragged(rid, len(row))))
        else:
            raise ValueError("cn_schedule can be 'flooding' or an array of ints.")

    def decode(self, llr_ch, num_iter=None, msg_v2c=None):
        """decoding.py:544-637.  llr_ch: [...,N_vn] logits; returns like the reference."""
        if num_iter is None:
            num_iter = self.num_iter
        llr_ch = np.asarray(llr_ch, self.dtype)
        assert llr_ch.shape[-1] == self.num_vns, "Last dimension must be of length n."
        shape = llr_ch.shape
        llr_ch = np.clip(llr_ch, -self.llr_max, self.llr_max)
        llr_ch = llr_ch.reshape(-1, self.num_vns).T.copy()      # [N_vn, B]
        llr_ch = llr_ch * self.dtype(-1.)
        if msg_v2c is None:
            msg_v2c = llr_ch[self.vn_idx]
        else:
            msg_v2c = np.asarray(msg_v2c, self.dtype) * self.dtype(-1)
        msg_c2v = np.zeros_like(msg_v2c)
        x_hat = llr_ch
        for _ in range(int(num_iter)):
            if self.cn_schedule is None:
                msg_cn = msg_v2c[self.v2c_perm]                                   # :479
                msg_c2v = self._cn_update(self._cn_rag, msg_cn, self.llr_max)      # :482,500
                msg_vn = msg_c2v[self.v2c_perm_inv]                               # :506
                msg_v2c, x_hat = self._vn_update(self._vn_rag, msg_vn, llr_ch, self.llr_max)
                continue
            # array schedule (:463-520): msg_c2v lives in CN-view order between sub-iterations
            for pos, rag in self._sched:
                msg_cn = msg_v2c[self.v2c_perm[pos]]                              # :474-479
                msg_c2v = msg_c2v.copy()
                msg_c2v[pos] = self._cn_update(rag, msg_cn, self.llr_max)          # :488-497
                msg_vn = msg_c2v[self.v2c_perm_inv]
                msg_v2c, x_hat = self._vn_update(self._vn_rag, msg_vn, llr_ch, self.llr_max)
        x_hat = x_hat.T
        if self.hard_out:
            x_hat = (0 >= x_hat).astype(self.dtype)                           # :623
        else:
            x_hat = x_hat * self.dtype(-1.)
        x_hat = x_hat.reshape(shape[:-1] + (self.num_vns,))
        if self.return_state:
            return x_hat, msg_v2c * self.dtype(-1)
        return x_hat


class LDPC5GDecoder(LDPCBPDecoder):
    """decoding.py:1169-1536.  ``code`` is an ``oracle.ldpc5g.LDPC5GCode``."""

    def __init__(self, code, cn_update="boxplus-phi", vn_update="sum", hard_out=True,
                 return_infobits=True, num_iter=20, llr_max=20., prune_pcm=True,
                 return_state=False, precision="single", cn_schedule="flooding"):
        self.code = code
        pcm = code.pcm
        self.return_infobits = return_infobits
        k_filler = code.k_ldpc - code.k
        nb_punc = (code.n_ldpc - k_filler) - code.n - 2 * code.z
        if prune_pcm:                                                          # :1344-1378
            dv = np.asarray(pcm.sum(axis=0)).reshape(-1)
            last_pos = code.n_ldpc
            for idx in range(code.n_ldpc - 1, 0, -1):
                if dv[idx] == 1:
                    last_pos = idx
                else:
                    break
            if isinstance(cn_schedule, str) and cn_schedule == "layered":      # :1359-1364
                nb_punc = int(np.floor(nb_punc / code.z) * code.z)
            self.n_pruned = int(max(last_pos, code.n_ldpc - nb_punc))
            self.nb_pruned = code.n_ldpc - self.n_pruned
            if self.nb_pruned > 0:
                pcm = pcm[:-self.nb_pruned, :-self.nb_pruned]
        else:
            self.nb_pruned, self.n_pruned = 0, code.n_ldpc
        if isinstance(cn_schedule, str) and cn_schedule == "layered":          # :1383-1389
            z = code.z
            cn_schedule = np.stack([np.arange(z) + i * z for i in range(pcm.shape[0] // z)], axis=0)
        super().__init__(pcm, cn_update, vn_update, hard_out, num_iter, llr_max,
                         return_state, precision, cn_schedule)

    def rate_recover(self, llr_ch):
        """decoding.py:1431-1475: [B,n] -> [B,N_vn] logits."""
        c = self.code
        b = llr_ch.shape[0]
        if c.num_bits_per_symbol is not None:
            llr_ch = llr_ch[:, c.out_int_inv]
        k_filler = c.k_ldpc - c.k
        nb_punc = (c.n_ldpc - k_filler) - c.n - 2 * c.z
        llr_5g = np.concatenate([np.zeros([b, 2 * c.z], self.dtype), llr_ch,
                                 np.zeros([b, nb_punc - self.nb_pruned], self.dtype)], 1)
        x1 = llr_5g[:, :c.k]
        nb_par = c.n_ldpc - k_filler - c.k - self.nb_pruned
        x2 = llr_5g[:, c.k:c.k + nb_par]
        z = -self.llr_max * np.ones([b, k_filler], self.dtype)
        return np.concatenate([x1, z, x2], axis=1)

    def decode5g(self, llr_ch, num_iter=None, msg_v2c=None):
        """decoding.py:1427-1536"""
        c = self.code
        llr_ch = np.asarray(llr_ch, self.dtype)
        if llr_ch.shape[-1] != c.n:
            raise ValueError("Last dimension must be of length n.")
        shape = llr_ch.shape
        llr_5g = self.rate_recover(llr_ch.reshape(-1, c.n))
        out = self.decode(llr_5g, num_iter=num_iter, msg_v2c=msg_v2c)
        x_hat, state = out if self.return_state else (out, None)
        if self.return_infobits:
            res = x_hat[:, :c.k].reshape(shape[:-1] + (c.k,))
        else:
            x = x_hat.reshape(-1, self.n_pruned)
            x_nf = np.concatenate([x[:, :c.k], x[:, c.k_ldpc:]], axis=1)
            x_short = x_nf[:, 2 * c.z:2 * c.z + c.n]
            if c.num_bits_per_symbol is not None:
                x_short = x_short[:, c.out_int]
            res = x_short.reshape(shape)
        return (res, state) if self.return_state else res
