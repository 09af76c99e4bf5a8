"""Oracle (NumPy/SciPy, CPU): 5G-NR LDPC code construction, encoder and rate matching.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  Restates
``fec/ldpc/encoding.py`` of the reference (paths relative to
/root/reference/src/sionna/phy):

* base-graph selection           encoding.py:248-282
* lifting-size selection         encoding.py:354-409
* base-graph table load          encoding.py:284-320  (tables = 38.212 Tab. 5.3.2-2/-3)
* lifting                        encoding.py:322-352
* RU sub-matrices / B^-1         encoding.py:411-522
* gather-index form + encode     encoding.py:524-591
* rate matching + interleaver    encoding.py:196-246, 634-661

Pinned by the reference's 28 golden generator matrices (tests/test_oracle_ldpc.py).
"""
import os
import numpy as np
import scipy.sparse as sp

_TABLES = os.path.join(os.path.dirname(__file__), "..", "sionna_amd", "phy", "fec",
                       "ldpc", "codes", "bg_tables.npz")

# 38.212 Tab. 5.3.2-1 (encoding.py:366-373)
_LIFT_SETS = [[2, 4, 8, 16, 32, 64, 128, 256],
              [3, 6, 12, 24, 48, 96, 192, 384],
              [5, 10, 20, 40, 80, 160, 320],
              [7, 14, 28, 56, 112, 224],
              [9, 18, 36, 72, 144, 288],
              [11, 22, 44, 88, 176, 352],
              [13, 26, 52, 104, 208],
              [15, 30, 60, 120, 240]]


def sel_basegraph(k, r, bg=None):
    """encoding.py:248-282"""
    if bg is None:
        if k <= 292:
            bg = "bg2"
        elif k <= 3824 and r <= 0.67:
            bg = "bg2"
        elif r <= 0.25:
            bg = "bg2"
        else:
            bg = "bg1"
    elif bg not in ("bg1", "bg2"):
        raise ValueError("Basegraph must be bg1, bg2 or None.")
    if bg == "bg1" and k > 8448:
        raise ValueError("K is not supported by BG1 (too large).")
    if bg == "bg2" and k > 3840:
        raise ValueError("K is not supported by BG2 (too large).")
    if bg == "bg1" and r < 1 / 3:
        raise ValueError("Only coderate>1/3 supported for BG1.")
    if bg == "bg2" and r < 1 / 5:
        raise ValueError("Only coderate>1/5 supported for BG2.")
    return bg


def sel_lifting(k, bg):
    """encoding.py:354-409 -> (z, i_ls, k_b)"""
    if bg == "bg1":
        k_b = 22
    elif k > 640:
        k_b = 10
    elif k > 560:
        k_b = 9
    elif k > 192:
        k_b = 8
    else:
        k_b = 6
    min_val, z, i_ls = 100000, 0, 0
    for i, s in enumerate(_LIFT_SETS):
        for s1 in s:
            x = k_b * s1
            if x >= k and x < min_val:
                min_val, z, i_ls = x, s1, i
    k_b = 22 if bg == "bg1" else 10
    return z, i_ls, k_b


def load_basegraph(i_ls, bg):
    """encoding.py:284-320: dense base matrix, -1 = empty."""
    t = np.load(_TABLES)
    shape = (46, 68) if bg == "bg1" else (42, 52)
    bm = np.zeros(shape) - 1
    bm[t[f"{bg}_row"], t[f"{bg}_col"]] = t[f"{bg}_shift"][:, i_ls]
    return bm


def lift_basegraph(bm, z):
    """encoding.py:322-352: column index of the '1' in row i of a block = (i+shift) mod z."""
    r_idx, c_idx = [], []
    im = np.arange(z)
    for r in range(bm.shape[0]):
        for c in range(bm.shape[1]):
            if bm[r, c] != -1:
                r_idx.append(r * z + im)
                c_idx.append(c * z + np.mod(im + bm[r, c], z))
    if len(r_idx) == 0:
        return sp.csr_matrix((z * bm.shape[0], z * bm.shape[1]))
    r_idx = np.concatenate(r_idx)
    c_idx = np.concatenate(c_idx)
    return sp.csr_matrix((np.ones(len(r_idx)), (r_idx, c_idx)),
                         shape=(z * bm.shape[0], z * bm.shape[1]))


def find_hm_b_inv(bm_b, z, bg):
    """encoding.py:436-522 (closed-form inverse of the 4Zx4Z core)."""
    pm_a = int(bm_b[0, 0])
    pm_b_inv = int(-bm_b[1, 0]) if bg == "bg1" else int(-bm_b[2, 0])
    hm = np.zeros([4 * z, 4 * z])
    im = np.eye(z)
    am = np.roll(im, pm_a, axis=1)
    b_inv = np.roll(im, pm_b_inv, axis=1)
    ab_inv = am @ b_inv
    for j in range(4):
        hm[0:z, j * z:(j + 1) * z] = b_inv
    hm[z:2 * z, 0:z] = im + ab_inv
    for j in (1, 2, 3):
        hm[z:2 * z, j * z:(j + 1) * z] = ab_inv
    if bg == "bg1":
        hm[2 * z:3 * z, 0:z] = ab_inv
        hm[2 * z:3 * z, z:2 * z] = ab_inv
        hm[2 * z:3 * z, 2 * z:3 * z] = im + ab_inv
        hm[2 * z:3 * z, 3 * z:4 * z] = im + ab_inv
    else:
        hm[2 * z:3 * z, 0:z] = im + ab_inv
        hm[2 * z:3 * z, z:2 * z] = im + ab_inv
        hm[2 * z:3 * z, 2 * z:3 * z] = ab_inv
        hm[2 * z:3 * z, 3 * z:4 * z] = ab_inv
    for j in (0, 1, 2):
        hm[3 * z:4 * z, j * z:(j + 1) * z] = ab_inv
    hm[3 * z:4 * z, 3 * z:4 * z] = im + ab_inv
    return sp.csr_matrix(hm)


def mat_to_ind(mat):
    """encoding.py:524-557: padded gather indices, pad value = n (points at an appended 0)."""
    mat = sp.csr_matrix(mat)
    m, n = mat.shape
    n_max = int(np.max(mat.getnnz(axis=1)))
    gat = np.zeros([m, n_max], dtype=np.int64) + n
    coo = mat.tocoo()
    order = np.lexsort((coo.col, coo.row))
    rr, cc, vv = coo.row[order], coo.col[order], coo.data[order]
    cnt = np.zeros(m, dtype=np.int64)
    for r, c, v in zip(rr, cc, vv):
        # entries with value 2 (I + I in B^-1 when P_A P_B^-1 = I) are kept with their
        # multiplicity by the reference's dense->csr conversion as a single entry of
        # value 2; sp.sparse.find returns it once, the gather adds the bit once.
        if v != 0:
            gat[r, cnt[r]] = c
            cnt[r] += 1
    return gat


def matmul_gather(ind, vec):
    """encoding.py:559-570 (sum of gathered entries in float32)."""
    vec = np.concatenate([vec, np.zeros([vec.shape[0], 1], vec.dtype)], axis=1)
    return vec[:, ind].sum(axis=-1, dtype=np.float32)


def generate_out_int(n, m):
    """encoding.py:196-246"""
    if n % m != 0:
        raise ValueError("n must be a multiple of num_bits_per_symbol.")
    perm = np.zeros(n, dtype=int)
    for j in range(n // m):
        for i in range(m):
            perm[i + j * m] = i * (n // m) + j
    return perm, np.argsort(perm)


class LDPC5GCode:
    """All static parameters of ``LDPC5GEncoder.__init__`` (encoding.py:61-137)."""

    def __init__(self, k, n, num_bits_per_symbol=None, bg=None):
        k, n = int(k), int(n)
        if k > 8448 or k < 12:
            raise ValueError("Unsupported code length (k).")
        if n > 316 * 384 or n < 0:
            raise ValueError("Unsupported code length (n).")
        self.k, self.n = k, n
        self.coderate = k / n
        if self.coderate > 0.95:
            raise ValueError("Unsupported coderate (r>0.95).")
        if self.coderate < 1 / 5:
            raise ValueError("Unsupported coderate (r<1/5).")
        self.bg = sel_basegraph(k, self.coderate, bg)
        self.z, self.i_ls, self.k_b = sel_lifting(k, self.bg)
        self.bm = load_basegraph(self.i_ls, self.bg)
        self.n_ldpc = self.bm.shape[1] * self.z
        self.k_ldpc = self.k_b * self.z
        self.pcm = lift_basegraph(self.bm, self.z)
        g, mb, k_b, z = 4, self.bm.shape[0], self.k_b, self.z
        self._a_ind = mat_to_ind(lift_basegraph(self.bm[0:g, 0:k_b], z))
        self._binv_ind = mat_to_ind(find_hm_b_inv(self.bm[0:g, k_b:k_b + g], z, self.bg))
        self._c1_ind = mat_to_ind(lift_basegraph(self.bm[g:mb, 0:k_b], z))
        self._c2_ind = mat_to_ind(lift_basegraph(self.bm[g:mb, k_b:k_b + g], z))
        self.num_bits_per_symbol = num_bits_per_symbol
        if num_bits_per_symbol is not None:
            self.out_int, self.out_int_inv = generate_out_int(n, num_bits_per_symbol)

    def encode_full(self, s):
        """encoding.py:572-591: s [B,k_ldpc] (float32 0/1) -> full codeword [B,n_ldpc]."""
        s = s.astype(np.float32)
        p_a = matmul_gather(self._a_ind, s)
        p_a = matmul_gather(self._binv_ind, p_a)
        p_b = matmul_gather(self._c1_ind, s) + matmul_gather(self._c2_ind, p_a)
        c = np.concatenate([s, p_a, p_b], axis=1)
        return (c.astype(np.uint8) & 1).astype(np.float32)

    def encode(self, bits):
        """encoding.py:599-668: [...,k] -> [...,n] including rate matching."""
        bits = np.asarray(bits, np.float32)
        if bits.shape[-1] != self.k:
            raise ValueError("Last dimension must be of length k.")
        lead = bits.shape[:-1]
        u = bits.reshape(-1, self.k)
        bsz = u.shape[0]
        u_fill = np.concatenate([u, np.zeros([bsz, self.k_ldpc - self.k], np.float32)], 1)
        c = self.encode_full(u_fill)
        c_nf = np.concatenate([c[:, :self.k], c[:, self.k_ldpc:]], axis=1)
        c_short = c_nf[:, 2 * self.z:2 * self.z + self.n]
        if self.num_bits_per_symbol is not None:
            c_short = c_short[:, self.out_int]
        return c_short.reshape(lead + (self.n,))
