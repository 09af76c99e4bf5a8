"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

CPU restatement of the 5G NR transport-block chain (38.212 Sec. 5/6.2, 38.214 Sec. 6.1.4.2):
  calculate_tb_size   /root/reference/src/sionna/phy/nr/utils.py:473-805
  TBEncoder           /root/reference/src/sionna/phy/nr/tb_encoder.py:15-435
  TBDecoder           /root/reference/src/sionna/phy/nr/tb_decoder.py:15-213

Pinned against the reference's eight transport-block vectors test/unit/nr/tb_refs/*.npz
(re-packed into tests/golden/tb_golden.npz by tools/gen_golden.py)."""
import numpy as np

from . import ldpc5g, ldpc_bp, polar as opolar, scrambling as osc

_TAB51321 = np.array([-1, 24, 32, 40, 48, 56, 64, 72, 80, 88, 96, 104, 112, 120, 128, 136, 144, 152, 160, 168, 176, 184, 192,
                      208, 224, 240, 256, 272, 288, 304, 320, 336, 352, 368, 384, 408, 432, 456, 480, 504, 528, 552, 576,
                      608, 640, 672, 704, 736, 768, 808, 848, 888, 928, 984, 1032, 1064, 1128, 1160, 1192, 1224, 1256,
                      1288, 1320, 1352, 1416, 1480, 1544, 1608, 1672, 1736, 1800, 1864, 1928, 2024, 2088, 2152, 2216,
                      2280, 2408, 2472, 2536, 2600, 2664, 2728, 2792, 2856, 2976, 3104, 3240, 3368, 3496, 3624, 3752,
                      3824], np.float32)


def calculate_tb_size(modulation_order, target_coderate, target_tb_size=None, num_coded_bits=None, num_layers=1):
    """nr/utils.py:617-805 for scalar arguments, float32 arithmetic like the reference's rdtype.
    Returns (tb_size, cb_size, num_cb, tb_crc_length, cb_crc_length, cw_lengths)."""
    f = np.float32
    assert num_coded_bits % num_layers == 0, "num_coded_bits must be a multiple of num_layers."
    if target_tb_size is not None:
        t = f(target_tb_size)
        assert t < f(num_coded_bits), "target_tb_size must be less than num_coded_bits."
    else:
        t = f(target_coderate) * f(num_coded_bits)
    if t <= 3824:
        n = max(f(3.0), f(np.floor(np.log(t) / f(np.log(2.0))) - 6))
        n_info_q = max(f(24.0), f(f(2) ** n * np.floor(t / f(2) ** n)))
    else:
        n = np.floor(np.log(t - f(24)) / np.log(f(2.0))) - f(5.)
        n_info_q = max(f(3840.0), f(f(2) ** n * np.round((t - f(24)) / f(2) ** n)))
    if n_info_q <= 3824:
        num_cb = f(1)
    elif target_coderate <= 1 / 4:
        num_cb = f(np.ceil((n_info_q + f(24)) / f(3816)))
    elif n_info_q > 8424:
        num_cb = f(np.ceil((n_info_q + f(24)) / f(8424)))
    else:
        num_cb = f(1)
    if n_info_q <= 3824:
        ge = _TAB51321 >= n_info_q
        ind = int(np.argmax(np.cumsum(1 - 2 * ge.astype(np.float32))))
        tb_size = int(_TAB51321[min(ind + 1, len(_TAB51321) - 1)])
    else:
        tb_size = int(f(8) * num_cb * np.ceil((n_info_q + f(24)) / (f(8) * num_cb)) - f(24))
    num_cb = int(num_cb)
    tb_crc = 24 if tb_size > 3824 else 16
    cb_crc = 24 if num_cb > 1 else 0
    cb_size = int((tb_size + tb_crc) / num_cb) + cb_crc
    q = num_layers * modulation_order
    num_last = int(num_coded_bits / q) % num_cb
    len_last = q * int(np.ceil(num_coded_bits / (q * num_cb)))
    len_first = q * int(np.floor(num_coded_bits / (q * num_cb)))
    cw = np.array([len_first] * (num_cb - num_last) + [len_last] * num_last, np.int64)
    return tb_size, cb_size, num_cb, tb_crc, cb_crc, cw


class TBEncoder:
    """tb_encoder.py:109-435 (single stream; lists of n_rnti/n_id = one stream per entry on axis -2)."""

    def __init__(self, target_tb_size, num_coded_bits, target_coderate, num_bits_per_symbol, num_layers=1, n_rnti=1,
                 n_id=1, channel_type="PUSCH", codeword_index=0, use_scrambler=True):
        self.n_rnti = list(n_rnti) if isinstance(n_rnti, (list, tuple)) else [n_rnti]
        self.n_id = list(n_id) if isinstance(n_id, (list, tuple)) else [n_id]
        self.num_tx = len(self.n_id)
        (self.tb_size, self.cb_size, self.num_cbs, self.tb_crc_length, self.cb_crc_length,
         self.cw_lengths) = calculate_tb_size(num_bits_per_symbol, target_coderate, target_tb_size, num_coded_bits,
                                              num_layers)
        assert self.tb_size <= self.tb_crc_length + np.sum(self.cw_lengths), "Invalid TB parameters."
        self.k_padding = self.tb_size - int(target_tb_size)
        self.k = int(target_tb_size)
        self.n = int(np.sum(self.cw_lengths))
        self.tb_crc = "CRC16" if self.tb_crc_length == 16 else "CRC24A"
        self.use_scrambler = use_scrambler
        self.c_init = [osc.tb5g_c_init(r, i, channel_type, codeword_index) for r, i in zip(self.n_rnti, self.n_id)]
        lmin, lmax = int(np.min(self.cw_lengths)), int(np.max(self.cw_lengths))
        self.code = ldpc5g.LDPC5GCode(self.cb_size, lmax, 1)
        p_short, _ = ldpc5g.generate_out_int(lmin, num_bits_per_symbol)
        p_long, _ = ldpc5g.generate_out_int(lmax, num_bits_per_symbol)
        perm, punc, pos = [], [], 0
        for l in self.cw_lengths:                                          # :252-280
            if l == lmin:
                perm.append(p_short + pos)
                punc.append(np.arange(pos + lmin, pos + lmax))
                pos += lmax
            else:
                perm.append(p_long + pos)
                pos += l
        self.output_perm = np.concatenate(perm + punc).astype(np.int64)
        self.output_perm_inv = np.argsort(self.output_perm)

    def encode(self, u):
        """u [..., (num_tx,) k] float 0/1 -> [..., (num_tx,) n]."""
        shape = u.shape
        u = np.asarray(u, np.float32).reshape(-1, self.num_tx, self.k)
        if self.k_padding > 0:
            u = np.concatenate([u, np.zeros(u.shape[:-1] + (self.k_padding,), np.float32)], axis=-1)
        u_crc = opolar.crc_encode(u.reshape(-1, self.tb_size), self.tb_crc)
        u_cb = u_crc.reshape(-1, self.cb_size - self.cb_crc_length)
        if self.cb_crc_length == 24:
            u_cb = opolar.crc_encode(u_cb, "CRC24B")
        c_cb = self.code.encode(u_cb)
        c = c_cb.reshape(-1, self.num_tx, self.num_cbs * int(np.max(self.cw_lengths)))
        c = c[..., self.output_perm][..., :self.n]
        if self.use_scrambler:
            seq = np.stack([osc.generate_prng_seq(self.n, ci) for ci in self.c_init], axis=0)
            c = osc.apply_scrambling(c, seq[None], binary=True)
        return c.reshape(shape[:-1] + (self.n,)).astype(np.float32)


class TBDecoder:
    """tb_decoder.py:71-213"""

    def __init__(self, encoder, num_bp_iter=20, cn_update="boxplus-phi"):
        self.enc = encoder
        self.dec = ldpc_bp.LDPC5GDecoder(encoder.code, cn_update=cn_update, num_iter=num_bp_iter, hard_out=True,
                                         return_infobits=True)

    def decode(self, llr):
        e = self.enc
        shape = llr.shape
        llr = np.asarray(llr, np.float32).reshape(-1, e.num_tx, e.n)
        if e.use_scrambler:
            seq = np.stack([osc.generate_prng_seq(e.n, ci) for ci in e.c_init], axis=0)
            llr = osc.apply_scrambling(llr, seq[None], binary=False)
        nfill = e.code.n * e.num_cbs - e.n
        llr = np.concatenate([llr, np.zeros(llr.shape[:-1] + (nfill,), np.float32)], axis=-1)
        llr = llr[..., e.output_perm_inv].reshape(-1, e.code.n)
        u_cb = self.dec.decode5g(llr)
        if e.cb_crc_length == 24:
            u_cb = u_cb[:, :-24]                                          # CB CRC status is not used (:190-193)
        u_tb = u_cb.reshape(-1, e.tb_size + e.tb_crc_length)
        u_hat, ok = opolar.crc_check(u_tb, e.tb_crc)
        u_hat = u_hat.reshape(shape[:-1] + (e.tb_size,))
        if e.k_padding > 0:
            u_hat = u_hat[..., :-e.k_padding]
        return u_hat.astype(np.float32), np.asarray(ok).reshape(shape[:-1])
