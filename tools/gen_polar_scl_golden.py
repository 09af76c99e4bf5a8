"""Generates tests/golden/polar_scl_np_golden.npz by EXECUTING the reference's own NumPy SC-list decoder
(PolarSCLDecoder._decode_np_batch and helpers, /root/reference/src/sionna/phy/fec/polar/decoding.py:1047-1290).

TensorFlow is not installed in this image, so `import sionna.phy` is impossible; the NumPy twin itself needs nothing
but NumPy.  The script therefore compiles the reference source file unmodified in a namespace where `tensorflow`
and the sionna imports are inert stand-ins, creates a PolarSCLDecoder WITHOUT running its TensorFlow __init__
(object.__new__) and sets exactly the attributes the twin reads (:1047-1290): _n, _n_stages, _list_size, _llr_max
(30., :435), _frozen_ind, _use_fast_scl, _cw_ind.  Inputs are the "true" LLRs `-1 * logits` in float32, as `call`
hands them to `tf.py_function` (:1374-1388).

Stored per case: frozen positions, the float32 logits, the bit-packed stage-0 candidate lists of all 2L decoders in
the twin's final sorted order, and their float64 path metrics.  Run here (needs /root/reference); the fixture travels.
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference/src/sionna/phy/fec/polar/decoding.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class _Inert:
    """Stands for anything TensorFlow-ish that the module touches at import time (decorators, dtypes)."""
    def __getattr__(self, name): return _Inert()
    def __call__(self, *a, **k): return a[0] if len(a) == 1 and callable(a[0]) and not k else _Inert()


def load_reference_class():
    stubs = {"tensorflow": _Inert(), "sionna": types.ModuleType("sionna"), "sionna.phy": types.ModuleType("sionna.phy"),
             "sionna.phy.fec": types.ModuleType("sionna.phy.fec"), "sionna.phy.fec.crc": types.ModuleType("crc"),
             "sionna.phy.fec.polar": types.ModuleType("polar"), "sionna.phy.fec.polar.encoding": types.ModuleType("enc")}
    stubs["sionna.phy"].Block = object
    stubs["sionna.phy.fec.crc"].CRCDecoder = stubs["sionna.phy.fec.crc"].CRCEncoder = object
    stubs["sionna.phy.fec.polar.encoding"].Polar5GEncoder = object
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    try:
        ns = {"__name__": "reference_polar_decoding"}
        with open(REF) as f:
            exec(compile(f.read(), REF, "exec"), ns)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return ns["PolarSCLDecoder"]


def reference_twin(cls, frozen_pos, n, list_size, use_fast_scl):
    dec = object.__new__(cls)
    dec._n, dec._n_stages, dec._list_size, dec._llr_max = n, int(np.log2(n)), list_size, 30.
    dec._frozen_ind = np.zeros(n)
    dec._frozen_ind[frozen_pos] = 1
    dec._use_fast_scl = use_fast_scl
    dec._cw_ind = np.arange(n)
    return dec


CASES = [  # (n, k, list_size, use_fast_scl, batch, ebno_db)   k includes the CRC bits
    (32, 16, 2, True, 64, 1.0), (64, 32, 4, True, 64, 1.5), (64, 40, 8, False, 48, 2.0), (128, 43, 8, True, 48, 1.0),
    (256, 128, 8, True, 32, 1.5), (256, 200, 4, True, 32, 3.5), (512, 256, 8, True, 24, 2.0), (1024, 523, 8, True, 24, 2.0),
    (1024, 523, 8, False, 8, 2.5), (128, 64, 16, True, 16, 1.0), (64, 12, 1, True, 32, 0.0),
]


def main():
    from oracle import polar as op
    cls = load_reference_class()
    rng = np.random.default_rng(2024)
    out = {}
    for ci, (n, k, L, fast, B, ebno) in enumerate(CASES):
        frozen, info = op.generate_5g_ranking(k, n)
        u = rng.integers(0, 2, (B, k))
        c = op.polar_encode(u, info, n)
        no = 1.0 / (10 ** (ebno / 10) * (k / n) * 1.0) / 2          # BPSK-equivalent noise variance per real dimension
        y = (1 - 2 * c) + np.sqrt(no) * rng.normal(size=c.shape)
        logits = (-2 * y / no).astype(np.float32)                    # logit = log p(1)/p(0)
        dec = reference_twin(cls, frozen, n, L, fast)
        llr_ch = (np.float32(-1.) * logits)                           # :1374
        msg_uhat, msg_pm = dec._decode_np_batch(llr_ch)
        lists = msg_uhat[:, :, 0, :].astype(np.uint8)                 # [B, 2L, n]
        out[f"c{ci}_meta"] = np.array([n, k, L, int(fast)], np.int32)
        out[f"c{ci}_frozen"] = frozen.astype(np.int16)
        out[f"c{ci}_logits"] = logits
        out[f"c{ci}_lists"] = np.packbits(lists, axis=-1)
        out[f"c{ci}_pm"] = msg_pm.astype(np.float64)
        ber = np.mean(lists[:, 0, :][:, info] != u)
        print(f"case {ci}: n={n} k={k} L={L} fast={fast} B={B}: BER of the best path {ber:.4f}")
    path = os.path.join(ROOT, "tests", "golden", "polar_scl_np_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
