#!/usr/bin/env python3
"""Derive compact golden vectors from the reference's own test fixtures.

The reference cannot be imported here (needs TensorFlow), but its golden FILES can be
read.  This script turns them into small fixtures under tests/golden/ that travel to
the GPU box (where /root/reference does not exist):

* ldpc_enc_golden.npz  - for each of the 28 golden generator matrices
  test/codes/ldpc/k{K}_n{N}_G.npy (sparse 1-based (row,col) pairs, recipe of
  test/unit/fec/test_ldpc_encoding.py:122-148): 8 seeded random info words u and
  c = u*G mod 2 computed from the REFERENCE matrix (bit-packed).
* crc_golden.npz / polar_golden.npz - straight re-packs of test/codes/{crc,polar}/*.npy.
* example_pcms.npz - the 4 small parity-check matrices of
  src/sionna/phy/fec/ldpc/codes/example_codes.npy used by the generic-decoder tests
  (fec/utils.py:478-523), stored as sparse index pairs.

Usage: python tools/gen_golden.py [/root/reference]
"""
import os, re, sys
import numpy as np

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(out_dir, exist_ok=True)

# ---- LDPC encoder
d = os.path.join(ref, "test/codes/ldpc")
out = {}
params = []
for f in sorted(os.listdir(d)):
    m = re.match(r"k(.*)_n(.*)_G.npy", f)
    if not m:
        continue
    k, n = int(m.group(1)), int(m.group(2))
    gm_sp = np.array(np.load(os.path.join(d, f), allow_pickle=True)).astype(np.int64)
    rows, cols = gm_sp[0] - 1, gm_sp[1] - 1           # gm[c-1, r-1] = 1 in the reference test
    rng = np.random.default_rng(1000 + k + n)
    u = rng.integers(0, 2, size=(8, k)).astype(np.uint8)
    c = np.zeros((8, n), np.int64)
    for b in range(8):
        np.add.at(c[b], cols, u[b, rows])
    c = (c & 1).astype(np.uint8)
    out[f"u_{k}_{n}"] = np.packbits(u, axis=1)
    out[f"c_{k}_{n}"] = np.packbits(c, axis=1)
    params.append((k, n))
out["params"] = np.array(params, np.int32)
np.savez_compressed(os.path.join(out_dir, "ldpc_enc_golden.npz"), **out)
print("ldpc:", len(params), "codes")

# ---- CRC / Polar: re-pack
for sub in ("crc", "polar"):
    d = os.path.join(ref, "test/codes", sub)
    pk = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(".npy"):
            a = np.load(os.path.join(d, f), allow_pickle=True)
            pk[f[:-4]] = np.asarray(a)
    np.savez_compressed(os.path.join(out_dir, f"{sub}_golden.npz"), **pk)
    print(sub, {k: (v.shape, str(v.dtype)) for k, v in pk.items()})

# ---- example parity-check matrices
ex = np.load(os.path.join(ref, "src/sionna/phy/fec/ldpc/codes/example_codes.npy"), allow_pickle=True)
pk = {}
for i in range(len(ex)):
    pcm = np.array(ex[i])
    r, c = np.nonzero(pcm)
    pk[f"shape_{i}"] = np.array(pcm.shape, np.int32)
    pk[f"rc_{i}"] = np.stack([r, c]).astype(np.int32)
np.savez_compressed(os.path.join(out_dir, "example_pcms.npz"), **pk)
print("example pcms:", [tuple(pk[f"shape_{i}"]) for i in range(len(ex))])

# ---- 5G NR transport-block vectors (test/unit/nr/tb_refs/*.npz): bit-packed re-pack
d = os.path.join(ref, "test/unit/nr/tb_refs")
pk, meta = {}, []
for i, f in enumerate(sorted(os.listdir(d))):
    t = np.load(os.path.join(d, f))
    for key in ("u_ref", "c_ref", "c_ref_no_scr"):
        pk[f"{key}_{i}"] = np.packbits(t[key].astype(np.uint8), axis=1)
    meta.append([t["u_ref"].shape[1], t["c_ref"].shape[1], int(t["n_id"]), int(t["n_rnti"]),
                 int(round(float(t["coderate"]) * 1000)), int(t["num_bits_per_symbol"]), int(t["num_layers"])])
pk["meta"] = np.array(meta, np.int64)     # tb_size, num_coded_bits, n_id, n_rnti, coderate*1000, m, layers
np.savez_compressed(os.path.join(out_dir, "tb_golden.npz"), **pk)
print("tb refs:", meta)
