"""NumPy models of the two QC kernels of csrc/ldpc5g.hip (same tables, same arithmetic,
same order).  They run on the CPU and are compared with the oracle in the ``not gpu``
suite, so that the ALGORITHMS (closed-form RU encoder on rotations; compressed check-node
state of the on-chip min-sum decoder) are validated even where no GPU is available.
They are test code: nothing in the product imports them.
"""
import numpy as np

F = np.float32


def _tables(enc):
    z = enc.z
    rows = enc._bg_rows.astype(int)
    cols = enc._bg_cols.astype(int)
    sh = enc._bg_shifts.astype(int) % z
    mb, nb = (46, 68) if enc._bg == "bg1" else (42, 52)
    by_row = [[] for _ in range(mb)]
    for r, c, s in zip(rows, cols, sh):
        by_row[r].append((c, s))
    for r in range(mb):
        by_row[r].sort()
    return z, mb, nb, by_row


def encode_qc_model(enc, u):
    """Model of ldpc5g_encode_kernel: u [B,k] 0/1 -> c [B,n]."""
    z, mb, nb, by_row = _tables(enc)
    k_b = enc._k_b
    B = u.shape[0]
    cw = np.zeros((B, nb * z), np.uint8)
    cw[:, :enc.k] = u.astype(np.uint8) & 1
    zz = np.arange(z)
    blk = lambda c: cw[:, c * z:(c + 1) * z]
    rot = lambda x, s: x[:, (zz + s) % z]                       # (P_s x)[z] = x[(z+s) mod Z]
    lam = []
    for r in range(4):
        acc = np.zeros((B, z), np.uint8)
        for c, s in by_row[r]:
            if c < k_b:
                acc ^= rot(blk(c), s)
        lam.append(acc)
    find = lambda r, c: [s for cc, s in by_row[r] if cc == c][0]
    s_a = find(0, k_b)
    s_b = find(1 if enc._bg == "bg1" else 2, k_b)
    p0 = rot(lam[0] ^ lam[1] ^ lam[2] ^ lam[3], -s_b)
    ap0 = rot(p0, s_a)
    p1 = lam[0] ^ ap0
    p3 = lam[3] ^ ap0
    p2 = (lam[2] ^ p3) if enc._bg == "bg1" else (lam[1] ^ p1)
    for j, p in enumerate((p0, p1, p2, p3)):
        cw[:, (k_b + j) * z:(k_b + j + 1) * z] = p
    for r in range(4, mb):
        acc = np.zeros((B, z), np.uint8)
        for c, s in by_row[r]:
            if c < k_b + 4:
                acc ^= rot(blk(c), s)
        cw[:, (k_b + r) * z:(k_b + r + 1) * z] = acc
    # rate matching (short_to_full(out_to_short(o)))
    n, k, k_ldpc = enc.n, enc.k, enc.k_ldpc
    o = np.arange(n)
    m = enc.num_bits_per_symbol
    t = o if m is None else (o % m) * (n // m) + o // m
    uu = t + 2 * z
    full = np.where(uu < k, uu, uu + (k_ldpc - k))
    return cw[:, full].astype(np.float32)


def decode_onchip_model(dec, llr, num_iter, offset=0.0):
    """Model of ldpc5g_decode_kernel (one codeword at a time, vectorised over lifted copies).

    dec: sionna_amd LDPC5GDecoder (host object, gives pruning); llr [B,n] logits.
    Returns x_hat internal LLRs clipped [B, N_vn] (callers map to outputs).
    """
    enc = dec.encoder
    z, mb, nb, by_row = _tables(enc)
    n_vn, n_cn = dec.num_vns, dec.num_cns
    llr_max = F(dec.llr_max)
    by_col = [[] for _ in range(nb)]
    for r in range(mb):
        for pos, (c, s) in enumerate(by_row[r]):
            by_col[c].append((r, s, pos))
    # rate recovery (recover_llr)
    B = llr.shape[0]
    k, n, k_ldpc = enc.k, enc.n, enc.k_ldpc
    v = np.arange(n_vn)
    u = np.where(v < k, v, v - (k_ldpc - k))
    t = u - 2 * z
    valid = (t >= 0) & (t < n) & ~((v >= k) & (v < k_ldpc))
    m = enc.num_bits_per_symbol
    tt = np.clip(t, 0, n - 1)
    o = tt if m is None else (tt // (n // m)) + (tt % (n // m)) * m
    rec = np.where(valid[None, :], llr[:, o], F(0))
    rec[:, (v >= k) & (v < k_ldpc)] = -llr_max
    out = np.zeros((B, n_vn), F)
    zz = np.arange(z)
    for b in range(B):
        l = (F(-1) * np.clip(rec[b], -llr_max, llr_max)).astype(F)
        xt = l.copy()
        m1 = np.zeros(n_cn, F); m2 = np.zeros(n_cn, F)
        idxs = np.zeros(n_cn, np.int64); sgn = np.zeros(n_cn, np.int64)
        for _ in range(num_iter):
            for r in range(mb):
                cn = r * z + zz
                act = cn < n_cn
                if not act.any():
                    continue
                cna = cn[act]; za = zz[act]
                d = len(by_row[r])
                min1 = np.full(len(cna), np.inf, F); min2 = np.full(len(cna), np.inf, F)
                idx = np.zeros(len(cna), np.int64); cnt = np.zeros(len(cna), np.int64)
                neg = np.zeros(len(cna), np.int64)
                for i, (c, s) in enumerate(by_row[r]):
                    c2v = np.where(idxs[cna] == i, m2[cna], m1[cna])
                    c2v = np.where((sgn[cna] >> i) & 1, -c2v, c2v)
                    v2c = np.clip(F(-1) * c2v + xt[c * z + (za + s) % z], -llr_max, llr_max).astype(F)
                    neg |= (v2c < 0).astype(np.int64) << i
                    a = np.abs(v2c)
                    lt = a < min1
                    eq = (a == min1) & ~lt
                    lt2 = (a < min2) & ~lt & ~eq
                    min2 = np.where(lt, min1, np.where(lt2, a, min2))
                    idx = np.where(lt, i, idx)
                    cnt = np.where(lt, 1, np.where(eq, cnt + 1, cnt))
                    min1 = np.where(lt, a, min1)
                with np.errstate(invalid="ignore"):
                    min_e = np.where(cnt == 1, (min2 - min1) + min1, min1).astype(F)
                a1 = np.minimum(np.maximum(min1 - F(offset), F(0)), llr_max)
                a2 = np.minimum(np.maximum(min_e - F(offset), F(0)), llr_max)
                par = np.array([bin(x).count("1") & 1 for x in neg])
                allm = (1 << d) - 1
                s_new = np.where(par == 1, ~neg & allm, neg)
                m1[cna], m2[cna], idxs[cna], sgn[cna] = a1, a2, idx, s_new
            new_xt = xt.copy()
            for c in range(nb):
                vn = c * z + zz
                act = vn < n_vn
                if not act.any():
                    continue
                vna = vn[act]; za = zz[act]
                x = np.zeros(len(vna), F)
                for (r, s, pos) in by_col[c]:
                    cn = r * z + (za - s) % z
                    ok = cn < n_cn
                    cnc = np.where(ok, cn, 0)
                    c2v = np.where(idxs[cnc] == pos, m2[cnc], m1[cnc])
                    c2v = np.where((sgn[cnc] >> pos) & 1, -c2v, c2v)
                    x = np.where(ok, x + c2v, x).astype(F)
                new_xt[vna] = x + l[vna]
            xt = new_xt
        out[b] = np.clip(xt, -llr_max, llr_max)
    return out
