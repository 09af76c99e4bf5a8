#!/usr/bin/env python3
"""Generates tests/golden/idd_ref_golden.npz by EXECUTING the reference's own iterative-detection-and-decoding chain under
the NumPy stand-in for TensorFlow (tools/ref_exec): the ``IddModel`` of
tutorials/phy/Introduction_to_Iterative_Detection_and_Decoding.ipynb (cells 9-13) with perfect-CSI Rayleigh fading -
``ofdm.LinearDetector("lmmse", "bit", "maxlog")`` (ofdm/detection.py:740-847 on OFDMDetector :21-317 and
mimo/detection.py:24-143) -> ``LDPC5GDecoder(return_infobits=False, return_state=True)`` (soft output + decoder state) ->
``ofdm.MMSEPICDetector`` with the decoder's LLRs as prior (ofdm/detection.py:1062-1173 on OFDMDetectorWithPrior :320-510,
mimo/detection.py:1314-1643) -> ``LDPC5GDecoder(..., msg_v2c=state)`` -, plus ``KBestDetector`` / ``EPDetector`` outputs on
the same received grid.  ResourceGrid / KroneckerPilotPattern / StreamManagement / ResourceGridMapper are the reference's
too (pilot symbols from a NumPy generator instead of TF's - perfect CSI, they do not enter).  16 receive antennas, 4
single-antenna users, fft 48, 16-QAM, 5G LDPC (1152, 2304) with the output interleaver, min-sum 12 iterations.

Also prints what a Monte-Carlo run of this reference-executed chain gives (``--mc SECONDS``): the evidence that the
BLER tables SAVED in that notebook for IDD are not what the reference's current code produces (profiles/r04_idd_ref_exec.txt).
Run here (needs /root/reference); the fixture travels."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "idd_ref_golden.npz")


class _StableNp:
    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def argsort(a, *args, **kw):
        kw.setdefault("kind", "stable")
        return np.argsort(a, *args, **kw)


def load():
    from tools.ref_exec import tf_numpy
    from tools.ref_exec.loader import reference
    ref = reference()
    ref.load_utils()
    tf = ref.tf
    tf.linalg.matrix_transpose = lambda a, **k: tf_numpy._t(np.swapaxes(np.asarray(a), -1, -2))
    tf.linalg.matvec = lambda a, b, adjoint_a=False, **k: tf_numpy._t(np.einsum(
        "...ij,...j->...i", np.conj(np.swapaxes(np.asarray(a), -1, -2)) if adjoint_a else np.asarray(a), np.asarray(b)))
    mp = ref.load("sionna.phy.mapping")

    class PilotSource:                                            # QAMSource stand-in: QPSK pilots from NumPy
        def __init__(self, *a, **k):
            self.rng = np.random.default_rng(0)

        def __call__(self, shape):
            return tf_numpy._t(((1 - 2 * self.rng.integers(0, 2, shape)) + 1j * (1 - 2 * self.rng.integers(0, 2, shape))).astype(np.complex64) / np.sqrt(2))
    mp.QAMSource = PilotSource
    mimo, ofdm = sys.modules["sionna.phy.mimo"], sys.modules["sionna.phy.ofdm"]
    for pkg, names in ((mimo, ("utils", "equalization", "detection", "stream_management")), (ofdm, ("pilot_pattern", "resource_grid"))):
        for n in names:
            m = ref.load(f"{pkg.__name__}.{n}")
            for k, v in vars(m).items():
                if not k.startswith("_"):
                    setattr(pkg, k, v)
    od = ref.load("sionna.phy.ofdm.detection")
    ref.load("sionna.phy.fec.ldpc.codes", package_dir=True)
    enc_m = ref.load("sionna.phy.fec.ldpc.encoding")
    dec_m = ref.load("sionna.phy.fec.ldpc.decoding")
    dec_m.np = _StableNp()                                        # see tools/gen_ldpc_bp_golden.py
    return mp, mimo, ofdm, od, enc_m, dec_m


class Chain:
    def __init__(self):
        mp, mimo, ofdm, od, enc_m, dec_m = load()
        self.n_ue, self.n_rx, self.m = 4, 16, 4
        self.rg = ofdm.ResourceGrid(num_ofdm_symbols=14, pilot_ofdm_symbol_indices=[2, 11], fft_size=48, num_tx=self.n_ue,
                                    pilot_pattern="kronecker", subcarrier_spacing=30e3)
        self.sm = mimo.StreamManagement(np.ones([1, self.n_ue]), 1)
        self.N = 48 * 12 * self.m
        self.K = self.N // 2
        const = mp.Constellation("qam", self.m)
        self.enc = enc_m.LDPC5GEncoder(self.K, self.N, num_bits_per_symbol=self.m)
        self.mapper, self.rgmap = mp.Mapper(constellation=const), ofdm.ResourceGridMapper(self.rg)
        kw = dict(constellation_type="qam", num_bits_per_symbol=self.m, hard_out=False)
        self.lmmse = od.LinearDetector("lmmse", "bit", "maxlog", self.rg, self.sm, **kw)
        self.kbest = od.KBestDetector("bit", self.n_ue, 64, self.rg, self.sm, **kw)
        self.ep = od.EPDetector("bit", self.rg, self.sm, self.m, l=10, hard_out=False)
        self.pic = od.MMSEPICDetector(output="bit", resource_grid=self.rg, stream_management=self.sm, demapping_method="maxlog",
                                      constellation=const, num_iter=1, hard_out=False)
        D = dec_m.LDPC5GDecoder
        self.siso = D(self.enc, return_infobits=False, num_iter=12, return_state=True, hard_out=False, cn_update="minsum")
        self.final = D(self.enc, return_infobits=True, return_state=True, hard_out=True, num_iter=12, cn_update="minsum")
        self.one_shot = D(self.enc, return_infobits=True, hard_out=True, num_iter=12, cn_update="minsum")

    def draw(self, rng, B, ebno_db):
        b = rng.integers(0, 2, (B, self.n_ue, 1, self.K)).astype(np.float32)
        x_rg = np.asarray(self.rgmap(self.mapper(self.enc(b))))
        h = ((rng.normal(size=(B, 1, self.n_rx, self.n_ue, 1)) + 1j * rng.normal(size=(B, 1, self.n_rx, self.n_ue, 1))) / np.sqrt(2))
        # OFDMChannel(normalize_channel=True): unit mean energy per (batch, rx, tx) link over antennas and the grid
        h = (h / np.sqrt(np.mean(np.abs(h) ** 2, axis=(2, 4), keepdims=True))).astype(np.complex64)
        no = np.float32(1 / (10 ** (ebno_db / 10) * 0.5 * self.m))
        hf = np.broadcast_to(h[..., None, None], h.shape + (14, 48)).copy()
        y = np.einsum("brmtsof,btsof->brmof", hf, x_rg)
        y = (y + np.sqrt(no / 2) * (rng.normal(size=y.shape) + 1j * rng.normal(size=y.shape))).astype(np.complex64)
        return b, h, hf, y, np.full([B], no, np.float32)

    def idd2(self, hf, y, no):
        ev = np.zeros(hf.shape, np.float32)
        llr0 = np.asarray(self.lmmse(y, hf, ev, no))
        llr_dec, msg = self.siso(llr0)
        llr1 = np.asarray(self.pic(y, hf, np.asarray(llr_dec), ev, no))
        bh, _ = self.final(llr1, msg_v2c=np.asarray(msg))
        return llr0, np.asarray(llr_dec), np.asarray(msg), llr1, np.asarray(bh)


def main():
    c = Chain()
    if "--mc" in sys.argv:
        secs = float(sys.argv[sys.argv.index("--mc") + 1])
        ebno = float(sys.argv[sys.argv.index("--ebno") + 1]) if "--ebno" in sys.argv else -7.0
        rng = np.random.default_rng(2024)
        e1 = e2 = nb = 0
        t0 = time.time()
        while time.time() - t0 < secs:
            b, h, hf, y, no = c.draw(rng, 32, ebno)
            llr0, _, _, _, bh = c.idd2(hf, y, no)
            e1 += int(np.sum((np.asarray(c.one_shot(llr0)) != b).reshape(-1, c.K).any(-1)))
            e2 += int(np.sum((bh != b).reshape(-1, c.K).any(-1)))
            nb += b.shape[0] * c.n_ue
            print(f"{ebno} dB, {nb} blocks: one-shot LMMSE BLER {e1 / nb:.4f}, IDD-2 BLER {e2 / nb:.4f}  [{time.time() - t0:.0f} s]", flush=True)
        return
    rng = np.random.default_rng(11)
    b, h, hf, y, no = c.draw(rng, 3, -7.5)
    llr0, llr_dec, msg, llr1, bh = c.idd2(hf, y, no)
    ev = np.zeros(hf.shape, np.float32)
    out = dict(b=b.astype(np.uint8), h=h, y=y, no=no, llr_lmmse=llr0, llr_dec=llr_dec, llr_pic=llr1, b_hat=bh.astype(np.uint8),
               state_head=msg[:4096].copy(), state_shape=np.array(msg.shape),
               llr_kbest=np.asarray(c.kbest(y, hf, ev, no)), llr_ep=np.asarray(c.ep(y, hf, ev, no)))
    import hashlib
    out["state_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(msg).tobytes()).digest(), np.uint8)
    np.savez_compressed(OUT, **out)
    print("IDD-2 block errors", int(np.sum((bh != b).reshape(-1, c.K).any(-1))), "of", b.shape[0] * 4)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
