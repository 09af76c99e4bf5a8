"""GPU parity of the Polar belief-propagation decoder (csrc/polar_bp.hip, ``PolarBPDecoder`` and
``Polar5GDecoder(dec_type="BP")``) against oracle/polar_bp.py in the defined float32 arithmetic (array_equal on soft
outputs) and against the reference's own PolarBPDecoder executed under the NumPy stand-in
(tests/golden/polar_bp_ref_golden.npz; the oracle is pinned to it bit for bit in tests/test_oracle_ref_exec_polar_bp.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import polar as op, polar_bp as obp

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "polar_bp_ref_golden.npz"))
PLAIN = [tuple(int(v) for v in r) for r in GOLD["plain"]]
FIVEG = [tuple(int(v) for v in r) for r in GOLD["fiveg"]]
unpack = lambda a, w: np.unpackbits(a, axis=1)[:, :w].astype(np.float32)


@pytest.fixture(scope="module")
def phy():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p


def _np(t):
    return t.detach().cpu().numpy()


def _report(name, got, want):
    d = np.abs(got - want)
    print(f"{name}: exact {np.mean(got == want):.6f}  max|d| {d.max():.3e}  sign flips {int(np.sum((got > 0) != (want > 0)))}")


# n = 2, 8: degenerate graphs; 32 ... 256: several codewords per workgroup (batch not a multiple of the packing);
# 512 / 1024: one codeword per workgroup; 2048: message columns in the device workspace
@pytest.mark.parametrize("k,n,it,B", [(1, 2, 3, 5), (4, 8, 20, 37), (16, 32, 20, 131), (40, 64, 7, 65), (64, 128, 20, 33),
                                      (100, 256, 2, 19), (256, 512, 20, 24), (512, 1024, 20, 40), (300, 1024, 1, 3),
                                      (1024, 2048, 5, 6)])
def test_bp_soft_output_bit_exact(phy, k, n, it, B):
    frozen = np.sort(np.random.default_rng(n + k).permutation(n)[:n - k]) if (n < 32 or n > 1024) else op.generate_5g_ranking(k, n)[0]
    rng = np.random.default_rng(7 * n + it)
    llr = (rng.normal(size=(B, n)) * 6 + 3).astype(np.float32)
    llr[0, : n // 2] = 0.                       # erasures
    llr[-1] *= 10.                              # far into the +-19.3 clip
    soft = _np(phy.fec.polar.PolarBPDecoder(frozen, n, num_iter=it, hard_out=False)(llr))
    want = obp.bp_decode(llr, frozen, n, it, hard_out=False, math="spec")
    _report(f"n={n} it={it}", soft, want)
    assert soft.shape == (B, k) and np.array_equal(soft, want)
    hard = _np(phy.fec.polar.PolarBPDecoder(frozen, n, num_iter=it, hard_out=True)(llr))
    assert np.array_equal(hard, obp.bp_decode(llr, frozen, n, it, hard_out=True, math="spec"))


@pytest.mark.parametrize("i", range(len(PLAIN)))
def test_bp_matches_reference_execution(phy, i):
    """the reference's PolarBPDecoder on NumPy's exp / log against the kernel on the defined exp / log: soft outputs within
    1e-4 of the clip value, decisions equal unless the reference's soft output sits at the threshold"""
    k, n, it, B = PLAIN[i]
    g = {key.split("/", 1)[1]: GOLD[key] for key in GOLD.files if key.startswith(f"p{i}/")}
    soft = _np(phy.fec.polar.PolarBPDecoder(g["frozen_pos"], n, num_iter=it, hard_out=False)(g["logits"]))
    assert np.max(np.abs(soft - g["soft"])) <= 1e-4 * 19.3
    hard = _np(phy.fec.polar.PolarBPDecoder(g["frozen_pos"], n, num_iter=it)(g["logits"]))
    flips = hard != unpack(g["hard"], k)
    assert np.all(np.abs(g["soft"][flips]) < 1e-3)


@pytest.mark.parametrize("i", range(len(FIVEG)))
def test_polar5g_bp_chain(phy, i):
    k, n, down, it, B = FIVEG[i]
    ct = "downlink" if down else "uplink"
    g = {key.split("/", 1)[1]: GOLD[key] for key in GOLD.files if key.startswith(f"g{i}/")}
    enc = phy.fec.polar.Polar5GEncoder(k, n, channel_type=ct)
    dec = phy.fec.polar.Polar5GDecoder(enc, dec_type="BP", num_iter=it, return_crc_status=True)
    assert dec.dec_type == "BP" and isinstance(dec.polar_dec, phy.fec.polar.PolarBPDecoder)
    u_hat, crc = dec(g["logits"])
    code = op.Polar5GCode(k, n, ct)
    want = op.polar5g_decode(code, g["logits"], "BP", num_iter=it, bp_math="spec")
    assert np.array_equal(_np(u_hat), want)
    u_crc = op.polar5g_decode(code, g["logits"], "BP", num_iter=it, bp_math="spec", keep_crc=True)
    assert np.array_equal(_np(crc).reshape(-1).astype(bool), op.crc_check(u_crc, code.crc_degree)[1].reshape(-1))
    # and the reference's own chain (executed): equal wherever its decisions are not at the threshold
    ref = unpack(g["u_hat"], k)
    same = np.all(_np(u_hat) == ref, axis=1)
    assert same.mean() >= 0.9, f"{int((~same).sum())} of {B} blocks differ from the reference-executed decisions"
    # leading dimensions, no CRC status
    dec2 = phy.fec.polar.Polar5GDecoder(enc, dec_type="BP", num_iter=it)
    out = dec2(g["logits"][: (B // 2) * 2].reshape(2, B // 2, n))
    assert out.shape == (2, B // 2, k) and np.array_equal(_np(out).reshape(-1, k), want[: (B // 2) * 2])


def test_bp_decodes_and_validates(phy):
    """BP-20 recovers clean codewords; constructor checks of the reference (decoding.py:1498-1525, 1575-1581)"""
    k, n = 128, 256
    frozen, _ = op.generate_5g_ranking(k, n)
    enc = phy.fec.polar.PolarEncoder(frozen, n)
    u = phy.mapping.BinarySource()([300, k])
    llr = (2. * _np(enc(u)) - 1.) * 8.
    dec = phy.fec.polar.PolarBPDecoder(frozen, n)
    assert np.array_equal(_np(dec(llr)), _np(u)) and (dec.n, dec.k, dec.num_iter, dec.hard_out, dec.llr_max) == (n, k, 20, True, 19.3)
    dec.num_iter = 3
    assert dec.num_iter == 3 and _np(dec(llr)).shape == (300, k)
    assert dec(np.zeros((0, n), np.float32)).shape == (0, k)
    with pytest.raises(ValueError):
        dec(np.zeros((2, n + 1), np.float32))
    for bad in (dict(num_iter=0), dict(num_iter=-2)):
        with pytest.raises(ValueError):
            phy.fec.polar.PolarBPDecoder(frozen, n, **bad)
    for bad in (dict(num_iter=2.5), dict(hard_out=1)):
        with pytest.raises(TypeError):
            phy.fec.polar.PolarBPDecoder(frozen, n, **bad)
    with pytest.raises(ValueError):
        phy.fec.polar.PolarBPDecoder(frozen[:10], 100)


@pytest.mark.parametrize("k,n", [(1, 32), (10, 32), (32, 32), (100, 256), (123, 1024), (1024, 1024)])
def test_bp_identity(phy, k, n):
    """the reference's own test (test/unit/fec/test_polar_decoding.py:742-769): noiseless BPSK is recovered, k = n included,
    and an input without batch dimension"""
    frozen, _ = op.generate_5g_ranking(k, n)
    frozen = np.asarray(frozen, int)
    enc = phy.fec.polar.PolarEncoder(frozen, n)
    dec = phy.fec.polar.PolarBPDecoder(frozen, n)
    u = phy.mapping.BinarySource()([10, k])
    assert np.array_equal(_np(dec(20. * (2. * enc(u) - 1.))), _np(u))
    u1 = phy.mapping.BinarySource()([k])
    out = dec(20. * (2. * enc(u1) - 1.))
    assert out.shape == (k,) and np.array_equal(_np(out), _np(u1))


@pytest.mark.parametrize("hard_out", [False, True])
def test_bp_numerics_and_float64_twin(phy, hard_out):
    """test_polar_decoding.py:819-842: 200 iterations on very large LLRs stay finite; :884-993: agreement with the
    reference test's float64 NumPy decoder (in-place message arrays) at its own tolerance"""
    k, n = 120, 256
    frozen, _ = op.generate_5g_ranking(k, n)
    rng = np.random.default_rng(5)
    big = rng.normal(2000., np.sqrt(4000.), (100, n)).astype(np.float32)
    out = _np(phy.fec.polar.PolarBPDecoder(frozen, n, hard_out=hard_out, num_iter=200)(big))
    assert np.all(np.isfinite(out))
    # float64 twin: the same schedule (the in-place arrays of the reference's test hold exactly the newest column values)
    k, n = 64, 128
    frozen, info = op.generate_5g_ranking(k, n)
    llr = rng.normal(2 / 0.3, np.sqrt(4 / 0.3), (100, n)).astype(np.float32)
    for it in (5, 10, 20, 40):
        S = 7
        L, R = np.zeros((100, S + 1, n)), np.zeros((100, S + 1, n))
        L[:, S] = -1. * llr
        R[:, 0, frozen] = 19.3

        def bx(x, y):
            x, y = np.clip(x, -19.3, 19.3), np.clip(y, -19.3, 19.3)
            return np.log(1 + np.exp(x + y)) - np.log(np.exp(x) + np.exp(y))
        for _ in range(it):
            for s in range(S):
                i1, i2 = obp.stage_indices(n, s)
                l1, l2, r1, r2 = L[:, s + 1, i1], L[:, s + 1, i2], R[:, s, i1], R[:, s, i2]
                R[:, s + 1, i1], R[:, s + 1, i2] = bx(r1, l2 + r2), bx(r1, l1) + r2
            for s in range(S - 1, -1, -1):
                i1, i2 = obp.stage_indices(n, s)
                l1, l2, r1, r2 = L[:, s + 1, i1], L[:, s + 1, i2], R[:, s, i1], R[:, s, i2]
                L[:, s, i1], L[:, s, i2] = bx(l1, l2 + r2), bx(r1, l1) + l2
        soft_ref = L[:, 0, info]
        got = _np(phy.fec.polar.PolarBPDecoder(frozen, n, hard_out=hard_out, num_iter=it)(llr))
        if hard_out:
            assert np.mean(got == 0.5 * (1 - np.sign(soft_ref))) > 0.9995
        else:
            assert np.allclose(-got, soft_ref, rtol=5e-2, atol=5e-3)
