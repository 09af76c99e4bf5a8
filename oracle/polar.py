"""Oracle (NumPy, CPU): CRC, 5G Polar encoder with rate matching, SC and CRC-aided SCL decoding.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Restates (paths relative to
/root/reference/src/sionna/phy/fec):

* CRCEncoder / CRCDecoder            crc.py:100-215, 289-321
* generate_5g_ranking                polar/utils.py:13-112  (38.212 Tab. 5.3.1.2-1)
* PolarEncoder.call                  polar/encoding.py:140-209
* Polar5GEncoder rate matching       polar/encoding.py:362-740 (sub-block / channel / input
                                     interleavers, puncturing / shortening / repetition)
* PolarSCDecoder                     polar/decoding.py:122-263
* PolarSCLDecoder (default TF path, use_fast_scl=True): polar/decoding.py:525-723, 919-1045,
                                     1345-1437 - restated on float32 arrays with the 2L-path /
                                     sort / duplicate structure of the reference
* Polar5GDecoder.call                polar/decoding.py:1999-2086

Pinned by the reference's golden vectors (tests/golden/{crc,polar}_golden.npz): CRC parity for
all six polynomials, 5G encoder + rate matching (5 configurations x 100 words), SC and SCL(L=1)
hard outputs (3 codes x 10 words).  The reference holds no vectors for list sizes > 1; those are pinned
by EXECUTING its own decoders - the NumPy twin (tools/gen_polar_scl_golden.py, round 3) and the whole
Polar5GEncoder / Polar5GDecoder chain with the TensorFlow list decoder under tools/ref_exec (round 4,
tests/test_oracle_ref_exec_polar.py: SC / SCL-8 / SCL-4 / hybrid decisions and CRC status bit for bit).
Tie-breaking of the path sort is fixed here to a stable sort (lowest position first).
"""
import os

import numpy as np

_SEQ = os.path.join(os.path.dirname(__file__), "..", "sionna_amd", "phy", "fec", "polar", "codes",
                    "polar_5g_sequence.npy")
F = np.float32

CRC_POLYS = {"CRC24A": (24, [24, 23, 18, 17, 14, 11, 10, 7, 6, 5, 4, 3, 1, 0]),
             "CRC24B": (24, [24, 23, 6, 5, 1, 0]),
             "CRC24C": (24, [24, 23, 21, 20, 17, 15, 13, 12, 8, 4, 2, 1, 0]),
             "CRC16": (16, [16, 12, 5, 0]), "CRC11": (11, [11, 10, 9, 5, 0]), "CRC6": (6, [6, 5, 0])}


# ------------------------------------------------------------------ CRC
def crc_gen_mat(k, crc_degree):
    """crc.py:100-156"""
    length, coeffs = CRC_POLYS[crc_degree]
    pol = np.zeros(length + 1, int)
    pol[[length - c for c in coeffs]] = 1
    g = np.zeros([k, length])
    x = np.zeros(length, dtype=int)
    x[0] = 1
    for i in range(k):
        x = np.concatenate([x, [0]])
        if x[0] == 1:
            x = np.bitwise_xor(x, pol)
        x = x[1:]
        g[k - i - 1, :] = x
    return g


def crc_encode(bits, crc_degree):
    """crc.py:175-215: [...,k] -> [...,k+crc]"""
    bits = np.asarray(bits, F)
    g = crc_gen_mat(bits.shape[-1], crc_degree)
    par = (bits.astype(np.int64) @ g.astype(np.int64)) % 2
    return np.concatenate([bits, par.astype(F)], axis=-1)


def crc_check(x_crc, crc_degree):
    """crc.py:289-321 -> (x_info, crc_valid[...,1])"""
    length = CRC_POLYS[crc_degree][0]
    par = crc_encode(x_crc, crc_degree)[..., -length:]
    return x_crc[..., :-length], np.sum(par, axis=-1, keepdims=True) == 0


# ------------------------------------------------------------------ code construction
def generate_5g_ranking(k, n, sort=True):
    """polar/utils.py:13-112"""
    q = np.load(_SEQ).astype(int)             # q[w] = channel index with reliability rank w
    order = q[q < n]                          # channels < n in ascending reliability
    frozen, info = order[:n - k], order[n - k:]
    if sort:
        frozen, info = np.sort(frozen), np.sort(info)
    return [frozen.astype(int), info.astype(int)]


_SUB_PERM = np.array([0, 1, 2, 4, 3, 5, 6, 7, 8, 16, 9, 17, 10, 18, 11, 19, 12, 20, 13, 21, 14, 22, 15, 23, 24,
                      25, 26, 28, 27, 29, 30, 31])
_P_IL_MAX = [0, 2, 4, 7, 9, 14, 19, 20, 24, 25, 26, 28, 31, 34, 42, 45, 49, 50, 51, 53, 54, 56, 58, 59, 61, 62, 65,
             66, 67, 69, 70, 71, 72, 76, 77, 81, 82, 83, 87, 88, 89, 91, 93, 95, 98, 101, 104, 106, 108, 110, 111, 113,
             115, 118, 119, 120, 122, 123, 126, 127, 129, 132, 134, 138, 139, 140, 1, 3, 5, 8, 10, 15, 21, 27, 29, 32,
             35, 43, 46, 52, 55, 57, 60, 63, 68, 73, 78, 84, 90, 92, 94, 96, 99, 102, 105, 107, 109, 112, 114, 116, 121,
             124, 128, 130, 133, 135, 141, 6, 11, 16, 22, 30, 33, 36, 44, 47, 64, 74, 79, 85, 97, 100, 103, 117, 125,
             131, 136, 142, 12, 17, 23, 37, 48, 75, 80, 86, 137, 143, 13, 18, 38, 144, 39, 145, 40, 146, 41, 147, 148,
             149, 150, 151, 152, 153, 154, 155, 156, 157, 158, 159, 160, 161, 162, 163]


def subblock_interleaving(u):
    """polar/encoding.py:362-397 (38.212 Sec. 5.4.1.1)"""
    k = u.shape[-1]
    assert k % 32 == 0
    y = np.zeros_like(u)
    for n in range(k):
        i = int(np.floor(32 * n / k))
        y[n] = u[int(_SUB_PERM[i] * k / 32 + np.mod(n, k / 32))]
    return y


def channel_interleaver(c):
    """polar/encoding.py:399-447 (triangular interleaver, 38.212 Sec. 5.4.1.3)"""
    n = c.shape[-1]
    t = 0
    while t * (t + 1) / 2 < n:
        t += 1
    v = np.full([t, t], np.nan)
    kk = 0
    for i in range(t):
        for j in range(t - i):
            if kk < n:
                v[i, j] = c[kk]
            kk += 1
    out = np.zeros_like(c)
    kk = 0
    for j in range(t):
        for i in range(t - j):
            if not np.isnan(v[i, j]):
                out[kk] = v[i, j]
                kk += 1
    return out


def input_interleaver(c):
    """polar/encoding.py:449-497 (38.212 Sec. 5.3.1.1)"""
    k = len(c)
    assert k <= 164
    out = np.empty(k, int)
    i = 0
    for p in _P_IL_MAX:
        if p >= 164 - k:
            out[i] = c[p - (164 - k)]
            i += 1
    return out


class Polar5GCode:
    """Static parameters of Polar5GEncoder (polar/encoding.py:282-321, 499-686)."""

    def __init__(self, k, n, channel_type="uplink"):
        k, n = int(k), int(n)
        assert n >= k and channel_type in ("uplink", "downlink")
        self.k_target, self.n_target, self.channel_type = k, n, channel_type
        if n < 18 or k > 1013 or n > 1088:
            raise ValueError("unsupported length")
        if channel_type == "uplink":
            if 12 <= k <= 19:
                self.crc_degree, k_crc = "CRC6", 6
            elif k >= 20:
                self.crc_degree, k_crc = "CRC11", 11
            else:
                raise ValueError("k_target<12 is not supported")
        else:
            if k > 140 or n < 25 or n > 576:
                raise ValueError("unsupported downlink configuration")
            self.crc_degree, k_crc = "CRC24C", 24
        self.k_crc = k_crc
        k_polar = k + k_crc
        if k_polar > n:
            raise ValueError("k_polar > n_target")
        n_min, n_max = 5, 10
        if n <= (9 / 8) * 2 ** (np.ceil(np.log2(n)) - 1) and k_polar / n < 9 / 16:
            n1 = np.ceil(np.log2(n)) - 1
        else:
            n1 = np.ceil(np.log2(n))
        n2 = np.ceil(np.log2(8 * k_polar))
        n_polar = int(2 ** np.max((np.min([n1, n2, n_max]), n_min)))
        pre = []
        if n < n_polar:
            if k_polar / n <= 7 / 16:                                  # puncturing
                n_int = 32 * np.ceil((n_polar - n) / 32)
                pat = subblock_interleaving(np.arange(n_int))
                for i in range(n_polar - n):
                    pre.append(int(pat[i]))
                if n >= 3 * n_polar / 4:
                    t = int(np.ceil(3 / 4 * n_polar - n / 2) - 1)
                else:
                    t = int(np.ceil(9 / 16 * n_polar - n / 4) - 1)
                for i in range(t):
                    pre.append(i)
            else:                                                      # shortening
                pat = subblock_interleaving(np.arange(32 * np.ceil(n_polar / 32)))
                for i in range(n, n_polar):
                    pre.append(pat[i])
        pre = np.unique(pre)
        ranking, _ = generate_5g_ranking(0, n_polar, sort=False)
        cand = np.setdiff1d(ranking, pre, assume_unique=True)
        info = np.sort([cand[-i - 1] for i in range(k_polar)]).astype(int)
        self.k_polar, self.n_polar = k_polar, n_polar
        self.info_pos = info
        self.frozen_pos = np.setdiff1d(np.arange(n_polar), info, assume_unique=True)
        self.ind_input_int = input_interleaver(np.arange(k_polar)) if channel_type == "downlink" else None
        sub = subblock_interleaving(np.arange(n_polar))
        idx = np.zeros(n)
        for i in range(n):
            if n >= n_polar:
                idx[i] = i % n_polar
            elif k_polar / n <= 7 / 16:
                idx[i] = i + n_polar - n
            else:
                idx[i] = i
        if channel_type == "uplink":
            ch = channel_interleaver(np.arange(n))
            self.ind_rate_matching = sub[idx[ch].astype(int)]
        else:
            self.ind_rate_matching = sub[idx.astype(int)]

    def encode(self, bits):
        """Polar5GEncoder.call (polar/encoding.py:697-740)"""
        bits = np.asarray(bits, F)
        lead = bits.shape[:-1]
        u = crc_encode(bits.reshape(-1, self.k_target), self.crc_degree)
        if self.channel_type == "downlink":
            u = u[:, self.ind_input_int]
        c = polar_encode(u, self.info_pos, self.n_polar)
        return c[:, self.ind_rate_matching].reshape(lead + (self.n_target,))


def polar_encode(u, info_pos, n):
    """PolarEncoder.call (polar/encoding.py:140-209): XOR butterfly over log2(n) stages."""
    x = np.zeros((u.shape[0], n), np.uint8)
    x[:, info_pos] = np.asarray(u).astype(np.uint8)
    for s in range(int(np.log2(n))):
        r = np.arange(n // 2)
        dest = r * 2 - np.mod(r, 2 ** s)
        x[:, dest] ^= x[:, dest + 2 ** s]
    return x.astype(F)


# ------------------------------------------------------------------ decoders
LLR_MAX = F(30.)


def _softplus(x):
    return np.logaddexp(F(0), x).astype(F)


def _cn_op(x, y):
    """polar/decoding.py:684-705"""
    x, y = np.clip(x, -LLR_MAX, LLR_MAX), np.clip(y, -LLR_MAX, LLR_MAX)
    return (_softplus(x + y) - np.logaddexp(x, y)).astype(F)


def sc_decode(llr_logits, frozen_pos, n):
    """PolarSCDecoder.call (polar/decoding.py:122-263): logits [B,n] -> u_hat at the info positions."""
    frozen = np.zeros(n, int)
    frozen[frozen_pos] = 1
    llr = (F(-1.) * np.asarray(llr_logits, F)).reshape(-1, n)

    def rec(l, fr):
        m = len(fr)
        if m > 1:
            if fr.sum() == m:
                z = np.zeros_like(l)
                return z, z
            l1, l2 = l[:, :m // 2], l[:, m // 2:]
            u1, u1u = rec(_cn_op(l1, l2), fr[:m // 2])
            u2, u2u = rec(((1 - 2 * u1u) * l1 + l2).astype(F), fr[m // 2:])
            up = np.concatenate([(u1u != u2u).astype(F), u2u], -1)
            return np.concatenate([u1, u2], -1), up
        if fr[0] == 1:
            z = np.zeros_like(l)
            return z, z
        u = (F(0.5) * (1 - np.sign(l))).astype(F)
        u = np.where(u == 0.5, F(1), u)
        return u, u
    u_hat, _ = rec(llr, frozen)
    info = np.setdiff1d(np.arange(n), frozen_pos)
    return u_hat[:, info]


class SCLDecoder:
    """PolarSCLDecoder, default TF path with use_fast_scl (polar/decoding.py:525-723, 919-1045)."""

    def __init__(self, frozen_pos, n, list_size=8, crc_degree=None, use_fast_scl=True, ind_iil_inv=None):
        self.n, self.L = n, list_size
        self.frozen = np.zeros(n, int)
        self.frozen[frozen_pos] = 1
        self.info_pos = np.setdiff1d(np.arange(n), frozen_pos)
        self.k = len(self.info_pos)
        self.stages = int(np.log2(n))
        self.crc_degree, self.fast, self.ind_iil_inv = crc_degree, use_fast_scl, ind_iil_inv

    def _sort(self):
        ind = np.argsort(self.pm, axis=-1, kind="stable")
        self.pm = np.take_along_axis(self.pm, ind, 1)
        self.uhat = np.take_along_axis(self.uhat, ind[:, :, None, None], 1)
        self.llr = np.take_along_axis(self.llr, ind[:, :, None, None], 1)

    def _dup(self):
        L = self.L
        self.uhat[:, L:] = self.uhat[:, :L]
        self.llr[:, L:] = self.llr[:, :L]
        self.pm[:, L:] = self.pm[:, :L]

    def _rec(self, cw):
        m, L = len(cw), self.L
        s = int(np.log2(m))
        if m > 1:
            if self.fast:
                if self.frozen[cw].sum() == m:                                         # rate-0
                    self.pm += _softplus(F(-1.) * np.clip(self.llr[:, :, s, cw], -LLR_MAX, LLR_MAX)).sum(-1, dtype=F)
                    return
                if self.frozen[cw[-1]] == 0 and self.frozen[cw[:-1]].sum() == m - 1:   # repetition
                    l_in = np.clip(self.llr[:, :, s, cw], -LLR_MAX, LLR_MAX)
                    l_pm = np.concatenate([l_in[:, :L], -l_in[:, L:]], 1)
                    self.pm += _softplus(F(-1.) * l_pm).sum(-1, dtype=F)
                    self.uhat[:, L:, s, cw[0]:cw[-1] + 1] = 1
                    self.uhat[:, L:, 0, cw[-1]] = 1
                    self._sort()
                    self._dup()
                    return
            left, right = cw[:m // 2], cw[m // 2:]
            self.llr[:, :, s - 1, left] = _cn_op(self.llr[:, :, s, left], self.llr[:, :, s, right])
            self._rec(left)
            u_l = self.uhat[:, :, s - 1, left]
            self.llr[:, :, s - 1, right] = ((1 - 2 * u_l) * self.llr[:, :, s, left] + self.llr[:, :, s, right]).astype(F)
            self._rec(right)
            u_l, u_r = self.uhat[:, :, s - 1, left], self.uhat[:, :, s - 1, right]
            self.uhat[:, :, s, cw] = np.concatenate([(u_l != u_r).astype(F), u_r], -1)
        else:
            i = cw[0]
            if self.frozen[i] == 0:
                self.uhat[:, L:, 0, i] = 1
            l_in = np.clip(self.llr[:, :, 0, i], -LLR_MAX, LLR_MAX)
            self.pm += _softplus(-((1 - 2 * self.uhat[:, :, 0, i]) * l_in)).astype(F)
            if self.frozen[i] == 0:
                self._sort()
                self._dup()

    def decode(self, llr_logits):
        """-> (u_hat [B,k] at the info positions, crc_status [B] or None)"""
        n, L = self.n, self.L
        llr_ch = (F(-1.) * np.asarray(llr_logits, F)).reshape(-1, n)
        B = llr_ch.shape[0]
        self.uhat = np.zeros([B, 2 * L, self.stages + 1, n], F)
        self.llr = np.zeros([B, 2 * L, self.stages + 1, n], F)
        self.llr[:, :, self.stages, :] = llr_ch[:, None, :]
        self.pm = np.zeros([B, 2 * L], F)
        self.pm[:, 1:L] = LLR_MAX
        self.pm[:, L + 1:] = LLR_MAX
        self._rec(np.arange(n))
        self._sort()
        pm = self.pm.copy()
        crc_valid = None
        if self.crc_degree is not None:
            u_list = self.uhat[:, :, 0, :][:, :, self.info_pos]
            if self.ind_iil_inv is not None:
                u_list = u_list[:, :, self.ind_iil_inv]
            _, crc_valid = crc_check(u_list, self.crc_degree)
            pm = pm + (1. - crc_valid[..., 0].astype(F)) * LLR_MAX * self.k
        cand = np.argmin(pm, axis=-1)
        c_hat = self.uhat[np.arange(B), cand, 0, :]
        status = crc_valid[np.arange(B), cand, 0] if crc_valid is not None else None
        return c_hat[:, self.info_pos], status


def polar5g_decode(code, llr_logits, dec_type="SC", list_size=8, num_iter=20, bp_math="spec", keep_crc=False):
    """Polar5GDecoder.call (polar/decoding.py:1999-2086): logits [...,n_target] -> [...,k_target].
    dec_type "BP": PolarBPDecoder with hard decisions (decoding.py:1896-1912; oracle/polar_bp.py)."""
    llr = np.asarray(llr_logits, F)
    lead = llr.shape[:-1]
    llr = llr.reshape(-1, code.n_target)
    n, npol = code.n_target, code.n_polar
    if code.channel_type == "uplink":
        llr = llr[:, np.argsort(channel_interleaver(np.arange(n)))]
    if n >= npol:
        n_rep = n - npol
        dem = np.concatenate([llr[:, :n_rep] + llr[:, npol:], llr[:, n_rep:npol]], 1)
    elif code.k_polar / n <= 7 / 16:
        dem = np.concatenate([np.zeros([llr.shape[0], npol - n], F), llr], 1)
    else:
        dem = np.concatenate([llr, -F(100.) * np.ones([llr.shape[0], npol - n], F)], 1)
    dec_in = dem[:, np.argsort(subblock_interleaving(np.arange(npol)))]
    iil_inv = np.argsort(code.ind_input_int) if code.channel_type == "downlink" else None
    if dec_type == "SC":
        u_crc = sc_decode(dec_in, code.frozen_pos, npol)
    elif dec_type == "BP":
        from .polar_bp import bp_decode
        u_crc = bp_decode(dec_in, code.frozen_pos, npol, num_iter, True, bp_math)
    else:
        u_crc, _ = SCLDecoder(code.frozen_pos, npol, list_size, code.crc_degree, ind_iil_inv=iil_inv).decode(dec_in)
    if iil_inv is not None:
        u_crc = u_crc[:, iil_inv]
    if keep_crc:                                                       # the decisions with their CRC bits (status checks)
        return u_crc
    return u_crc[:, :-code.k_crc].reshape(lead + (code.k_target,))
