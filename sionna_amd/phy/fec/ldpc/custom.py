"""Extension points of the BP decoders: node-update FUNCTIONS and message CALLBACKS.

The reference lets users replace the check / variable node update by a callable and register ``c2v_callbacks`` /
``v2c_callbacks`` that see every iteration's messages as a ragged tensor [num_nodes, None, batch]
(src/sionna/phy/fec/ldpc/decoding.py:231-251, 309-322, 416-524; callbacks: fec/ldpc/utils.py:12-236).  Python
callables cannot run inside the hand-written HIP engines, so a decoder with a custom callable or a callback runs
its message passing HERE: the same loop as ``_bp_iter`` on device tensors with torch operations - segment
reductions over the node groups, differentiable (weighted BP trains through it).  It is the slow, general path; the
string rules without callbacks stay on the HIP engines.  There is no CPU path: tensors live on the HIP device.

``RaggedMessages`` is the stand-in for ``tf.RaggedTensor``: ``flat_values`` [num_values, batch] ordered by node,
``row_splits`` [num_nodes + 1].  The exported node updates ``cn_update_minsum`` / ``cn_update_offset_minsum`` /
``cn_update_phi`` / ``cn_update_tanh`` / ``vn_update_sum`` follow the reference formulas (decoding.py:681-1166) on it;
passing one of them as ``cn_update=`` selects the corresponding HIP rule."""
import numpy as np
import torch


class RaggedMessages:
    """Messages grouped by node (a ragged [num_nodes, None, batch] tensor)."""

    def __init__(self, flat_values, row_splits, row_ids=None):
        self.flat_values = flat_values
        self.row_splits = row_splits                                  # int64 [num_rows + 1], same device
        self._row_ids = row_ids

    @property
    def nrows(self):
        return int(self.row_splits.numel()) - 1

    @property
    def shape(self):
        return (self.nrows, None) + tuple(self.flat_values.shape[1:])

    def row_lengths(self):
        return self.row_splits[1:] - self.row_splits[:-1]

    def value_rowids(self):
        if self._row_ids is None:
            self._row_ids = torch.repeat_interleave(torch.arange(self.nrows, device=self.row_splits.device),
                                                    self.row_lengths())
        return self._row_ids

    def with_flat_values(self, v):
        return RaggedMessages(v, self.row_splits, self._row_ids)

    def map_flat_values(self, fn, *args):
        return self.with_flat_values(fn(self.flat_values, *args))

    def _reduce(self, kind):
        lengths = self.row_lengths()
        return torch.segment_reduce(self.flat_values, kind, lengths=lengths, axis=0, unsafe=True)

    def reduce_sum(self):
        return self._reduce("sum")

    def reduce_prod(self):
        return self._reduce("prod")

    def reduce_min(self):
        return self._reduce("min")

    def reduce_max(self):
        return self._reduce("max")

    def expand(self, per_row):
        """[num_rows, batch] -> [num_values, batch]"""
        return per_row[self.value_rowids()]

    def __mul__(self, other):
        return self.with_flat_values(self.flat_values * other)

    __rmul__ = __mul__

    def __neg__(self):
        return self.with_flat_values(-self.flat_values)


def _clip(x, v):
    return x if v is None else torch.clamp(x, -v, v)


def _sign_no_zero(x):
    return torch.where(x < 0, -torch.ones_like(x), torch.ones_like(x))


def vn_update_sum(msg_c2v_rag, llr_ch, llr_clipping=None):
    """decoding.py:681-732: x_tot = sum of incoming c2v + llr_ch; v2c = x_tot - own c2v; both clipped."""
    x_tot = msg_c2v_rag.reduce_sum() + llr_ch
    x_e = -1.0 * msg_c2v_rag.flat_values + msg_c2v_rag.expand(x_tot)
    return msg_c2v_rag.with_flat_values(_clip(x_e, llr_clipping)), _clip(x_tot, llr_clipping)


def vn_node_update_identity(msg_c2v_rag, llr_ch, llr_clipping=None):  # decoding.py:644-679
    return msg_c2v_rag, msg_c2v_rag.reduce_sum() + llr_ch


def cn_node_update_identity(msg_v2c_rag, llr_clipping=None):  # decoding.py:735-753
    return msg_v2c_rag


def cn_update_offset_minsum(msg_v2c_rag, llr_clipping=None, offset=0.5):
    """decoding.py:755-909: sign product x (minimum of the OTHER magnitudes - offset)_+."""
    large = 100000.0
    msg = torch.clamp(msg_v2c_rag.flat_values, -large, large)
    rag = msg_v2c_rag.with_flat_values
    sign_val = _sign_no_zero(msg)
    sign_val = sign_val * msg_v2c_rag.expand(rag(sign_val).reduce_prod())
    msg = msg.abs()
    min_val = rag(msg).reduce_min()
    msg_min1 = msg - msg_v2c_rag.expand(min_val)
    msg = torch.where(msg_min1 == 0, torch.full_like(msg, large), msg_min1)
    min_val_2 = rag(msg).reduce_min() + min_val
    node_sum = rag(msg).reduce_sum() - (2 * large - 1.0)
    double_min = 0.5 * (1 - torch.sign(node_sum))
    min_val_e = (1 - double_min) * min_val + double_min * min_val_2
    msg_e = torch.where(msg == large, msg_v2c_rag.expand(min_val_e), msg_v2c_rag.expand(min_val))
    msg_e = torch.clamp(msg_e - offset, min=0.0)
    return rag(_clip(sign_val * msg_e, llr_clipping))


def cn_update_minsum(msg_v2c_rag, llr_clipping=None):
    """decoding.py:911-953"""
    return cn_update_offset_minsum(msg_v2c_rag, llr_clipping, offset=0.0)


def cn_update_tanh(msg, llr_clipping=None):
    """decoding.py:955-1043: 2 atanh(prod of tanh(x/2) of the other edges)."""
    x = torch.tanh(msg.flat_values / 2)
    x = torch.where(x == 0, torch.full_like(x, 1e-12), x)
    prod = msg.with_flat_values(x).reduce_prod()
    x = (1.0 / x) * msg.expand(prod)
    x = torch.where(x.abs() < 1e-7, torch.zeros_like(x), x)
    x = torch.clamp(x, -(1 - 1e-7), 1 - 1e-7)
    return msg.with_flat_values(_clip(2 * torch.atanh(x), llr_clipping))


def _phi(x):
    """decoding.py:1092-1120"""
    if x.dtype == torch.float32:
        x = torch.clamp(x, 8.5e-8, 16.635532)
    else:
        x = torch.clamp(x, 1e-12, 28.324079)
    return torch.log(torch.exp(x) + 1) - torch.log(torch.exp(x) - 1)


def cn_update_phi(msg, llr_clipping=None):
    """decoding.py:1045-1166"""
    v = msg.flat_values
    sign_val = _sign_no_zero(v)
    sign_val = sign_val * msg.expand(msg.with_flat_values(sign_val).reduce_prod())
    p = _phi(v.abs())
    tot = msg.with_flat_values(p).reduce_sum()
    out = sign_val * _phi(-1.0 * p + msg.expand(tot))
    return msg.with_flat_values(_clip(out, llr_clipping))


BUILTIN_CN = {"boxplus": cn_update_tanh, "boxplus-phi": cn_update_phi, "boxplus-phi-fast": cn_update_phi, "minsum": cn_update_minsum,
              "min": cn_update_minsum, "offset-minsum": cn_update_offset_minsum, "identity": cn_node_update_identity}
BUILTIN_VN = {"sum": vn_update_sum, "identity": vn_node_update_identity}
# exported functions given as ``cn_update=`` select the HIP rule of the same name
FUNCTION_TO_RULE = {cn_update_tanh: "boxplus", cn_update_phi: "boxplus-phi", cn_update_minsum: "minsum",
                    cn_update_offset_minsum: "offset-minsum"}


class CustomBPEngine:
    """``LDPCBPDecoder._bp_iter`` (decoding.py:416-524) on device tensors.  Edge order: VN-major, ascending CN inside a
    VN (the decoder's state order), so the c2v buffer is already grouped by VN."""

    def __init__(self, cn_idx, vn_idx, num_cns, num_vns, cn_update, vn_update, c2v_callbacks, v2c_callbacks,
                 cn_schedule=None):
        self.cn_idx, self.vn_idx = np.asarray(cn_idx, np.int64), np.asarray(vn_idx, np.int64)
        self.num_cns, self.num_vns = num_cns, num_vns
        self.cn_update, self.vn_update = cn_update, vn_update
        self.c2v_callbacks, self.v2c_callbacks = list(c2v_callbacks), list(v2c_callbacks)
        self.cn_schedule = None if cn_schedule is None else np.asarray(cn_schedule, np.int64)
        self._dev = None

    def _tables(self, device):
        if self._dev is None or self._dev["device"] != device:
            t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int64, device=device)
            perm = np.argsort(self.cn_idx, kind="stable")                 # edges in CN-major order (decoding.py:327)
            cn_splits = np.concatenate([[0], np.cumsum(np.bincount(self.cn_idx, minlength=self.num_cns))])
            vn_splits = np.concatenate([[0], np.cumsum(np.bincount(self.vn_idx, minlength=self.num_vns))])
            subs = []
            if self.cn_schedule is not None:
                for row in self.cn_schedule:                              # active CNs of every sub-iteration
                    pos = np.concatenate([perm[cn_splits[c]:cn_splits[c + 1]] for c in row])
                    lens = np.array([cn_splits[c + 1] - cn_splits[c] for c in row])
                    subs.append((t(pos), t(np.concatenate([[0], np.cumsum(lens)]))))
            self._dev = {"device": device, "perm": t(perm), "cn_splits": t(cn_splits), "vn_splits": t(vn_splits),
                         "vn_idx": t(self.vn_idx), "subs": subs}
        return self._dev

    def decode(self, llr, num_iter, llr_max, msg_v2c=None):
        """llr [B, N_vn] logits (device) -> (x_hat [B, N_vn] in LLR sign before the final sign change, v2c [E, B])."""
        d = self._tables(llr.device)
        llr_t = -1.0 * torch.clamp(llr, -llr_max, llr_max).transpose(0, 1)       # [N, B]   decoding.py:552-565
        v2c = llr_t[d["vn_idx"]] if msg_v2c is None else -1.0 * msg_v2c
        if self.v2c_callbacks:
            # decoding.py:583-594: every v2c callback sees the initial (or state-in) messages once, iteration 0, with the
            # channel LLRs as third argument - weighted BP weights them, an EXIT callback records mi[0]
            vrag = RaggedMessages(v2c, d["vn_splits"], d["vn_idx"])
            for cb in self.v2c_callbacks:
                vrag = cb(vrag, 0, llr_t)
            v2c = vrag.flat_values
        c2v = torch.zeros_like(v2c)
        x_hat = llr_t
        for it in range(int(num_iter)):
            subs = d["subs"] if self.cn_schedule is not None else [(d["perm"], d["cn_splits"])]
            for pos, splits in subs:
                rag = RaggedMessages(v2c[pos], splits)
                rag = self.cn_update(rag, llr_max)
                for cb in self.c2v_callbacks:
                    rag = cb(rag, it)
                c2v = c2v.index_copy(0, pos, rag.flat_values)          # flooding: every edge; schedule: active CNs
                vrag = RaggedMessages(c2v, d["vn_splits"], d["vn_idx"])
                vrag, x_hat = self.vn_update(vrag, llr_t, llr_max)
                for cb in self.v2c_callbacks:
                    vrag = cb(vrag, it + 1, x_hat)
                v2c = vrag.flat_values
        return x_hat.transpose(0, 1), v2c


# ------------------------------------------------------------------ callbacks (fec/ldpc/utils.py:12-236)
class EXITCallback:
    """Tracks the mutual information of the messages after every iteration (all-zero codeword simulations);
    register as ``c2v_callbacks`` or ``v2c_callbacks``."""

    def __init__(self, num_iter):
        self._mi = np.zeros(num_iter + 1, np.float64)
        self._num_samples = np.zeros(num_iter + 1, np.float64)

    @property
    def mi(self):
        with np.errstate(invalid="ignore", divide="ignore"):
            return self._mi / self._num_samples

    def __call__(self, msg, it, *args, **kwargs):
        from ..utils import llr2mi
        self._mi[it] += float(llr2mi(-1 * msg.flat_values.detach()))
        self._num_samples[it] += 1.0
        return msg


class DecoderStatisticsCallback:
    """Counts, per iteration, the codewords whose check nodes are all satisfied (``c2v_callbacks``)."""

    def __init__(self, num_iter):
        self._num_iter = num_iter
        self.reset_stats()

    num_samples = property(lambda self: self._num_samples)
    num_decoded_cws = property(lambda self: self._decoded_samples)

    @property
    def success_rate(self):
        return self._decoded_samples.astype(np.float64) / self._num_samples.astype(np.float64)

    @property
    def avg_number_iterations(self):
        active = self._num_samples.astype(np.float64) - self._decoded_samples.astype(np.float64)
        return float(np.sum(active) / self._num_samples[0])

    def reset_stats(self):
        self._num_samples = np.zeros(self._num_iter, np.int64)
        self._decoded_samples = np.zeros(self._num_iter, np.int64)

    def __call__(self, msg, it, *args, **kwargs):
        sign_node = msg.with_flat_values(_sign_no_zero(msg.flat_values.detach())).reduce_prod()
        cw_success = torch.all(sign_node > 0, dim=0)
        self._num_samples[it] += int(msg.flat_values.shape[-1])
        self._decoded_samples[it] += int(cw_success.sum())
        return msg


class WeightedBPCallback:
    """Weighted BP [Nachmani]: multiplies every edge message by a trainable weight (``torch.nn.Parameter``,
    initialised to one); the custom engine is differentiable, so the weights train with torch autograd."""

    def __init__(self, num_edges, precision=None, **kwargs):  # pylint: disable=unused-argument
        from .... import _ffi
        dt = torch.float64 if precision == "double" else torch.float32
        self._edge_weights = torch.nn.Parameter(torch.ones(num_edges, dtype=dt, device=_ffi.device()))

    weights = property(lambda self: self._edge_weights)

    def show_weights(self, size=7):
        import matplotlib.pyplot as plt
        plt.figure(figsize=(size, size))
        plt.hist(self._edge_weights.detach().cpu().numpy(), density=True, bins=20, align="mid")
        plt.xlabel("weight value")
        plt.ylabel("density")
        plt.grid(True, which="both", axis="both")
        plt.title("Weight Distribution")

    def __call__(self, msg, *args):
        return msg.with_flat_values(msg.flat_values * self._edge_weights.to(msg.flat_values.dtype)[:, None])
