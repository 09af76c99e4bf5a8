"""GPU parity tests: the HIP path (through the C-ABI, via the sionna.phy-style blocks) against
the CPU oracle on the same seeded inputs, against the committed golden fixtures, and - at
BASELINE.json's full sizes - through size-independent properties.

Bars (BASELINE.json north_star): bit-exact for encoder / interleaver / hard decisions /
counters / random bit stream and for the whole min-sum decoder (defined summation order);
<= 1e-5 relative (+ small absolute floor, stated per test) for LLRs of the demapper and
the boxplus decoders.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ffi_option(key, value="1"):
    """development switch through the C entry samd_debug_set_option (the library never reads the environment after load)"""
    from sionna_amd import _ffi
    return _ffi.option(key, value)

from oracle.ldpc5g import LDPC5GCode
from oracle import ldpc_bp as obp, mapping as omap, utils as outil, cbind

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def phy():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p


def _np(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ RNG / sources / AWGN
def test_binary_source_bit_exact(phy):
    phy.config.seed = 1234
    src = phy.mapping.BinarySource()
    for call, shape in enumerate([(7,), (3, 1001), (64, 2816)]):
        got = _np(src(shape))
        ref = outil.random_bits(1234, call, int(np.prod(shape))).reshape(shape)
        assert np.array_equal(got, ref)
    assert abs(float(got.mean()) - 0.5) < 0.01


def test_awgn_matches_stream_spec(phy):
    phy.config.seed = 77
    x = (np.arange(10001) % 7 - 3 + 1j * (np.arange(10001) % 5 - 2)).astype(np.complex64)
    y = _np(phy.channel.AWGN()(x, 0.37))
    ref = outil.awgn(x, 0.37, 77, 0)
    assert np.allclose(y, ref, rtol=1e-5, atol=2e-6)
    no = np.linspace(0.01, 2, x.size).astype(np.float32)
    y2 = _np(phy.channel.AWGN()(x, no))
    assert np.allclose(y2, outil.awgn(x, no, 77, 1), rtol=1e-5, atol=2e-6)
    # statistics of the noise itself
    w = _np(phy.utils.complex_normal([200000], 2.0))
    assert abs(np.var(w) - 2.0) < 0.03 and abs(np.mean(w)) < 0.01


# ------------------------------------------------------------------ mapper / demapper
@pytest.mark.parametrize("m", [2, 4, 6, 8])
def test_mapper_bit_exact(phy, m):
    rng = np.random.default_rng(m)
    bits = rng.integers(0, 2, (5, 24 * m)).astype(np.float32)
    x = _np(phy.mapping.Mapper("qam", m)(bits))
    assert np.array_equal(x, omap.mapper(bits, omap.qam(m)))


@pytest.mark.parametrize("m", [2, 4, 6, 8])
@pytest.mark.parametrize("method", ["app", "maxlog"])
@pytest.mark.parametrize("separable", [True, False])
def test_demapper_vs_oracle(phy, m, method, separable):
    """Both demapper kernels (per-axis square-QAM path and generic 2^m-point path) against the
    float64 oracle (reference formula mapping.py:664-691/927-967; the reference's own test uses
    atol 1e-5 against scipy logsumexp, test_mapping.py:175-199)."""
    rng = np.random.default_rng(10 + m)
    pts = omap.qam(m)
    y = (pts[rng.integers(0, 2 ** m, (4, 300))]
         + (rng.normal(size=(4, 300)) + 1j * rng.normal(size=(4, 300))) * 0.3).astype(np.complex64)
    for no in (np.float32(0.2), rng.uniform(0.01, 100, size=(4, 300)).astype(np.float32)):
        llr = _np(phy.mapping.Demapper(method, "qam", m, separable=separable)(y, no))
        ref64 = omap.demapper(y.astype(np.complex128), np.asarray(no, np.float64), pts.astype(np.complex128), method)
        # north-star bar: 1e-5 relative; the float32 reference itself carries ~1e-6 * |exponent|
        assert np.allclose(llr, ref64, rtol=1e-5, atol=1e-4 * max(1.0, float(np.max(np.abs(ref64))) * 1e-2))
        hard = _np(phy.mapping.Demapper(method, "qam", m, hard_out=True, separable=separable)(y, no))
        sure = np.abs(ref64) > 1e-3
        assert np.array_equal(hard[sure], (ref64 > 0).astype(np.float32)[sure])


@pytest.mark.parametrize("m", [1, 2, 4, 6])
@pytest.mark.parametrize("method", ["app", "maxlog"])
def test_demapper_with_prior(phy, m, method):
    """Demapper.call(y, no, prior) (mapping.py:664-691, 944-967)."""
    rng = np.random.default_rng(m)
    pts = omap.qam(m) if m > 1 else omap.pam(1).astype(np.complex64) if hasattr(omap, "pam") else np.array([1, -1], np.complex64)
    ctype = "qam" if m > 1 else "pam"
    y = (rng.normal(size=(7, 50)) + 1j * rng.normal(size=(7, 50))).astype(np.complex64)
    no = rng.uniform(0.05, 1.0, (7, 50)).astype(np.float32)
    dm = phy.mapping.Demapper(method, ctype, m)
    pts = np.asarray(dm.constellation.points)
    for prior in (rng.normal(size=m).astype(np.float32) * 2, rng.normal(size=(7, 50, m)).astype(np.float32) * 3):
        got = _np(dm(y, no, prior))
        ref = omap.demapper(y, no, pts, method, prior=prior)
        assert np.allclose(got, ref, rtol=1e-4, atol=2e-4), np.max(np.abs(got - ref))
    # a zero prior changes nothing; a strong prior dominates the decision
    assert np.allclose(_np(dm(y, no, np.zeros(m, np.float32))), _np(dm(y, no)), rtol=1e-4, atol=2e-4)
    strong = np.where(rng.integers(0, 2, m) > 0, 5000.0, -5000.0).astype(np.float32)
    hard = _np(phy.mapping.Demapper(method, ctype, m, hard_out=True)(y, no, strong))
    assert np.array_equal(hard.reshape(7, 50, m), np.broadcast_to(strong > 0, (7, 50, m)).astype(np.float32))


def test_demapper_custom_constellation(phy):
    # a rotated / non-square constellation has no per-axis structure -> generic kernel
    rng = np.random.default_rng(2)
    pts = (omap.qam(4) * np.exp(1j * 0.3)).astype(np.complex64)
    const = phy.mapping.Constellation("custom", 4, points=pts)
    assert const.pam_levels() is None
    y = (pts[rng.integers(0, 16, (3, 100))] + 0.2 * (rng.normal(size=(3, 100)) + 1j * rng.normal(size=(3, 100)))).astype(np.complex64)
    llr = _np(phy.mapping.Demapper("app", constellation=const)(y, 0.1))
    ref = omap.demapper(y.astype(np.complex128), 0.1, pts.astype(np.complex128), "app")
    assert np.allclose(llr, ref, rtol=1e-5, atol=1e-3)
    x = _np(phy.mapping.Mapper(constellation=const)(rng.integers(0, 2, (2, 40)).astype(np.float32)))
    assert x.shape == (2, 10)


def test_demapper_measures_against_raw_points_like_the_reference(phy):
    """Reference-EXECUTED pin (tests/golden/phy_ref_golden.npz, tools/gen_phy_ref_golden.py): for a "custom" constellation
    with normalize=center=True the reference's Mapper transmits ``constellation()`` (mapping.py:514) while its Demapper and
    SymbolDemapper measure distances to the stored, un-normalised ``constellation.points`` (mapping.py:667-668, 777)."""
    g = np.load(os.path.join(GOLD, "phy_ref_golden.npz"))
    const = phy.mapping.Constellation("custom", 3, points=g["custom3_in"], normalize=True, center=True)
    assert np.allclose(np.asarray(const()), g["custom3_points"], rtol=1e-6, atol=1e-7)
    got = _np(phy.mapping.Demapper("app", constellation=const)(g["custom3_y"], 0.3))
    assert np.allclose(got, g["custom3_app"], rtol=1e-5, atol=1e-4), np.abs(got - g["custom3_app"]).max()
    sd = _np(phy.mapping.SymbolDemapper(constellation=const)(g["custom3_y"], 0.3))
    ref = omap.symbol_demapper(g["custom3_y"], 0.3, g["custom3_in"])
    assert np.allclose(sd, ref, rtol=1e-4, atol=1e-4)
    # reference-executed QAM demapper outputs, straight against the kernels (both the per-axis and the generic one)
    for m in (2, 4, 6, 8):
        for meth in ("app", "maxlog"):
            for no in (0.5, 0.05):
                ref = g[f"qam{m}_{meth}_no{no}"]
                y = g[f"qam{m}_y_no{no}"]
                for sep in (True, False):
                    got = _np(phy.mapping.Demapper(meth, "qam", m, separable=sep)(y, no))
                    scale = max(np.abs(ref).max(), np.abs(y).max() ** 2 / no)
                    assert np.abs(got - ref).max() <= 1e-5 * scale, (m, meth, no, sep, np.abs(got - ref).max())
            got = _np(phy.mapping.Demapper(meth, "qam", m)(g[f"qam{m}_y_t"], g[f"qam{m}_no_t"], g[f"qam{m}_prior"]))
            assert np.allclose(got, g[f"qam{m}_{meth}_prior"], rtol=1e-5, atol=2e-4)


# ------------------------------------------------------------------ encoder
G = np.load(os.path.join(GOLD, "ldpc_enc_golden.npz"))


@pytest.mark.parametrize("k,n", [tuple(int(v) for v in p) for p in G["params"]])
def test_encoder_golden(phy, k, n):
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    u = np.unpackbits(G[f"u_{k}_{n}"], axis=1)[:, :k].astype(np.float32)
    c_ref = np.unpackbits(G[f"c_{k}_{n}"], axis=1)[:, :n].astype(np.float32)
    assert np.array_equal(_np(enc(u)), c_ref)


@pytest.mark.parametrize("k,n,bg,m", [(1024, 2048, "bg1", None), (2816, 8448, "bg1", 6), (200, 600, None, 2),
                                      (3840, 4800, "bg2", 4)])
def test_encoder_vs_oracle_interleaved(phy, k, n, bg, m):
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    u = np.random.default_rng(k).integers(0, 2, (2, 5, k)).astype(np.float32)   # multi-dim batch
    assert np.array_equal(_np(enc(u)), LDPC5GCode(k, n, m, bg).encode(u))


@pytest.mark.parametrize("k,n,bg,m", [(2816, 8448, "bg1", 6), (1408, 2816, "bg1", None), (704, 1500, "bg1", 2),
                                      (8448, 12672, "bg1", 4), (5632, 16896, "bg1", 6), (1280, 3840, "bg2", 4),
                                      (2560, 5120, "bg2", None)])
def test_encoder_bit_packed_vs_oracle_and_byte_kernel(phy, k, n, bg, m):
    """Lifting sizes that are multiples of 32 (C2: Z = 128) run the bit-packed encoder (one wave per codeword, rotated
    blocks as v_alignbit of two words, output through the folded interleaver / puncturing table): same bits as the
    oracle and as the byte-per-bit kernel (SAMD_ENC_BYTES=1), odd batch sizes included (four codewords per workgroup)."""
    import os
    code = LDPC5GCode(k, n, m, bg)
    assert code.z % 32 == 0, code.z
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    rng = np.random.default_rng(n)
    for batch in (1, 7, 130):
        u = rng.integers(0, 2, (batch, k)).astype(np.float32)
        got = _np(enc(u))
        assert np.array_equal(got, code.encode(u)), (k, n, batch)
        with _ffi_option("SAMD_ENC_BYTES"):                      # a fresh handle is built under the switch
            assert np.array_equal(_np(enc(u)), got)


# ------------------------------------------------------------------ generic BP decoder
def _example_pcm(i):
    ex = np.load(os.path.join(GOLD, "example_pcms.npz"))
    pcm = np.zeros(tuple(ex[f"shape_{i}"]), np.float32)
    pcm[ex[f"rc_{i}"][0], ex[f"rc_{i}"][1]] = 1
    return pcm


def _close(a, b, what, frac=0.999, max_abs=None):
    """1e-5 relative with an absolute floor of 1e-4 (LLRs are bounded by llr_max=20)."""
    ok = np.isclose(a, b, rtol=1e-5, atol=1e-4)
    assert ok.mean() > frac, f"{what}: {1 - ok.mean():.2e} of the outputs outside tolerance, max {np.max(np.abs(a - b))}"
    if max_abs is not None:
        assert np.max(np.abs(a - b)) <= max_abs, f"{what}: max deviation {np.max(np.abs(a - b))}"


@pytest.mark.parametrize("pcm_id", [0, 1, 3, 4])
@pytest.mark.parametrize("cn", ["minsum", "offset-minsum", "boxplus-phi", "boxplus"])
def test_generic_decoder_vs_oracle(phy, pcm_id, cn):
    pcm = _example_pcm(pcm_id)
    rng = np.random.default_rng(pcm_id)
    llr = rng.normal(loc=-1.2, scale=2.5, size=(37, pcm.shape[1])).astype(np.float32)
    llr[0, :3] = 0.0
    llr[1] = np.round(llr[1])                       # ties / duplicate minima
    for it in (0, 1, 5):
        dec = phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update=cn, hard_out=False, num_iter=it, return_state=True)
        x, st = dec(llr)
        ref = obp.LDPCBPDecoder(pcm, cn_update=cn, hard_out=False, num_iter=it, return_state=True)
        xr, sr = ref.decode(llr)
        if cn in ("minsum", "offset-minsum", "boxplus-phi"):
            # boxplus-phi: phi on the DEFINED float32 exp / log (bp_math.h == oracle/ldpc_bp.c) -> bit for bit as well
            assert np.array_equal(_np(x), xr) and np.array_equal(_np(st), sr), f"{cn} it={it}"
        else:
            _close(_np(x), xr, f"{cn} it={it} x_hat")
            _close(_np(st), sr, f"{cn} it={it} state")
    if cn == "boxplus-phi":
        # the same rule on the hardware transcendentals: the round-2 tolerance bars
        xf, stf = phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update="boxplus-phi-fast", hard_out=False, num_iter=5, return_state=True)(llr)
        _close(_np(xf), xr, "boxplus-phi-fast x_hat")
        _close(_np(stf), sr, "boxplus-phi-fast state")
    hard = _np(phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update=cn, num_iter=5)(llr))
    sure = np.abs(xr) > 1e-3
    assert np.array_equal(hard[sure], (xr > 0).astype(np.float32)[sure])        # logits>0 <=> bit 1
    assert set(np.unique(hard)) <= {0.0, 1.0}


def test_generic_decoder_state_passing(phy):
    # decoding.py:569-573/636: 2 x 3 iterations with state == 6 iterations
    pcm = _example_pcm(3)
    llr = np.random.default_rng(3).normal(loc=-1, scale=2, size=(9, pcm.shape[1])).astype(np.float32)
    d3 = phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update="minsum", hard_out=False, num_iter=3, return_state=True)
    d6 = phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update="minsum", hard_out=False, num_iter=6, return_state=True)
    _, st = d3(llr)
    x2, st2 = d3(llr, msg_v2c=st)
    x6, st6 = d6(llr)
    assert np.array_equal(_np(x2), _np(x6)) and np.array_equal(_np(st2), _np(st6))


def test_generic_decoder_invariants(phy):
    pcm = _example_pcm(3)
    for cn in ("minsum", "offset-minsum", "boxplus-phi", "boxplus"):
        dec = phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update=cn, hard_out=False, num_iter=5, return_state=True)
        x, st = dec(np.zeros((4, pcm.shape[1]), np.float32))
        assert float(x.abs().max()) == 0 and float(st.abs().max()) == 0        # all-erasure
        x, st = dec(np.random.default_rng(0).normal(scale=50, size=(8, pcm.shape[1])).astype(np.float32))
        assert float(x.abs().max()) <= 20 and float(st.abs().max()) <= 20


def test_big_degree_fallback(phy):
    # dense parity-check matrix: CN degree 48, VN degree ~12 -> re-read kernels
    rng = np.random.default_rng(5)
    pcm = (rng.random((16, 64)) < 0.75).astype(np.float32)
    pcm[:, 0] = 1
    llr = rng.normal(loc=-0.5, scale=3, size=(10, 64)).astype(np.float32)
    for cn in ("minsum", "boxplus-phi"):
        x = _np(phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update=cn, hard_out=False, num_iter=3)(llr))
        xr = obp.LDPCBPDecoder(pcm, cn_update=cn, hard_out=False, num_iter=3).decode(llr)
        if cn == "minsum":
            assert np.array_equal(x, xr)
        else:
            _close(x, xr, "bigdeg phi")


def test_symbol_sources(phy):
    phy.config.seed = 5
    x, ind, b = phy.mapping.QAMSource(4, return_indices=True, return_bits=True)([3, 50])
    pts = omap.qam(4)
    assert tuple(x.shape) == (3, 50) and tuple(b.shape) == (3, 50, 4)
    assert np.array_equal(_np(ind), (_np(b) * [8, 4, 2, 1]).sum(-1).astype(np.int32))
    assert np.allclose(_np(x), pts[_np(ind)])
    assert np.array_equal(_np(b), outil.random_bits(5, 0, 600).reshape(3, 50, 4))
    y = _np(phy.mapping.QAMSource(2, seed=7)([1000]))
    assert y.shape == (1000,) and np.allclose(np.abs(y), 1.0, atol=1e-6) and abs(y.mean()) < 0.1
    z = _np(phy.mapping.PAMSource(2)([4, 8]))
    assert np.allclose(z.imag, 0) and set(np.round(np.unique(z.real) * np.sqrt(5), 3)) <= {-3.0, -1.0, 1.0, 3.0}
    custom = phy.mapping.Constellation("custom", 1, points=np.array([1j, -1j]))
    assert set(np.unique(_np(phy.mapping.SymbolSource(constellation=custom)([64])))) <= {1j, -1j}


# ------------------------------------------------------------------ CN schedules (layered decoding)
def test_scheduling_independent_checks(phy):
    """Reference test_ldpc_decoding.py:121-160."""
    pcm = np.array([[1, 1, 1, 0, 0, 0], [0, 0, 0, 1, 1, 1]], np.float32)
    x = np.arange(6, dtype=np.float32)
    outs = []
    for cns in ("flooding", np.stack([[0], [1]]), np.stack([[0], [0]])):
        dec = phy.fec.ldpc.LDPCBPDecoder(pcm, num_iter=10, hard_out=False, cn_update="minsum", cn_schedule=cns,
                                         llr_max=100000)
        outs.append(_np(dec(x)))
        ref = obp.LDPCBPDecoder(pcm, num_iter=10, hard_out=False, cn_update="minsum", cn_schedule=cns, llr_max=100000)
        assert np.array_equal(outs[-1], ref.decode(x))
    assert np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], outs[2])
    for bad in (np.zeros(3, np.int32), np.array([[0, 2]]), np.array([[-1, 0]]), "layered"):
        with pytest.raises(ValueError):
            phy.fec.ldpc.LDPCBPDecoder(pcm, cn_schedule=bad)


@pytest.mark.parametrize("pcm_id", [0, 3, 4])
@pytest.mark.parametrize("cn", ["minsum", "offset-minsum", "boxplus-phi", "boxplus"])
def test_custom_schedule_vs_oracle(phy, pcm_id, cn):
    """Arbitrary array schedules (groups of unequal structure, CNs skipped or repeated across
    rows) against the oracle's literal restatement of _bp_iter (full VN update every
    sub-iteration); state in/out included."""
    pcm = _example_pcm(pcm_id)
    m = pcm.shape[0]
    rng = np.random.default_rng(pcm_id + 10)
    llr = rng.normal(loc=-1.2, scale=2.5, size=(21, pcm.shape[1])).astype(np.float32)
    llr[1] = np.round(llr[1])
    w = max(1, m // 3)
    perm = rng.permutation(m)
    sched = np.stack([perm[:w], perm[-w:], np.sort(perm[w:2 * w]) if m >= 2 * w else perm[:w]], axis=0)
    for it in (0, 1, 4):
        dec = phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update=cn, hard_out=False, num_iter=it, return_state=True,
                                         cn_schedule=sched)
        ref = obp.LDPCBPDecoder(pcm, cn_update=cn, hard_out=False, num_iter=it, return_state=True, cn_schedule=sched)
        x, st = dec(llr)
        xr, sr = ref.decode(llr)
        x2, st2 = dec(llr, msg_v2c=st)                                  # IDD: continue from the state
        xr2, sr2 = ref.decode(llr, msg_v2c=sr)
        if cn in ("minsum", "offset-minsum", "boxplus-phi"):
            assert np.array_equal(_np(x), xr) and np.array_equal(_np(st), sr), f"{cn} it={it}"
            assert np.array_equal(_np(x2), xr2) and np.array_equal(_np(st2), sr2), f"{cn} it={it} (state in)"
        else:
            _close(_np(x), xr, f"{cn} it={it} x_hat")
            _close(_np(st), sr, f"{cn} it={it} state")
            _close(_np(x2), xr2, f"{cn} it={it} x_hat (state in)")


@pytest.mark.parametrize("k,n", [(12, 25), (20, 65), (45, 63), (12, 59), (500, 1000)])
def test_scheduling_pruning_5g(phy, k, n):
    """Reference test_ldpc_decoding.py:735-757 + bit-exactness against the oracle."""
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    code = LDPC5GCode(k, n)
    x = np.arange(n, dtype=np.float32)[None]
    out = []
    for p in (False, True):
        kw = dict(cn_schedule="layered", num_iter=5, return_infobits=False, hard_out=False, llr_max=10000,
                  cn_update="minsum", prune_pcm=p)
        y = -_np(phy.fec.ldpc.LDPC5GDecoder(enc, **kw)(-x))
        assert np.array_equal(y, -obp.LDPC5GDecoder(code, **kw).decode5g(-x))
        out.append(y)
    assert np.allclose(out[0], out[1])


def test_layered_5g_vs_oracle_and_convergence(phy):
    """Layered 5G decoding: bit-exact (min-sum) / tolerance (boxplus) against the oracle, and the
    reference's rule of thumb (test_ldpc_decoding.py:689-733): 8 layered ~ 16 flooding iterations."""
    k, n = 200, 400
    code = LDPC5GCode(k, n)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    u, c, llr = _noisy_llr(code, 40, 11, sigma=0.75)
    kw = dict(cn_update="minsum", cn_schedule="layered", num_iter=4, hard_out=False)
    assert np.array_equal(_np(phy.fec.ldpc.LDPC5GDecoder(enc, **kw)(llr)), obp.LDPC5GDecoder(code, **kw).decode5g(llr))
    # boxplus rules: well-conditioned regime for the 1e-5 bar (tanh/atanh and phi amplify one ulp
    # to ~0.5 once messages saturate, see test_5g_boxplus_vs_oracle); saturating regime: signs agree
    u, c, llr_lo = _noisy_llr(code, 40, 13, sigma=1.3)
    for cn in ("boxplus", "boxplus-phi"):
        kw = dict(cn_update=cn, cn_schedule="layered", num_iter=1, hard_out=False)
        _close(_np(phy.fec.ldpc.LDPC5GDecoder(enc, **kw)(llr_lo)), obp.LDPC5GDecoder(code, **kw).decode5g(llr_lo),
               f"layered {cn}", frac=0.995)
        kw["num_iter"] = 4
        got, ref = _np(phy.fec.ldpc.LDPC5GDecoder(enc, **kw)(llr)), obp.LDPC5GDecoder(code, **kw).decode5g(llr)
        sure = np.abs(ref) > 1.0
        assert np.array_equal(got[sure] > 0, ref[sure] > 0) and np.mean(np.abs(got - ref) < 0.05) > 0.98
    u, c, llr = _noisy_llr(code, 4000, 12, sigma=0.79)
    bler = {}
    for cns, it in (("layered", 8), ("flooding", 16), ("flooding", 8)):
        b_hat = _np(phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="boxplus", cn_schedule=cns, num_iter=it)(llr))
        bler[(cns, it)] = np.mean(np.any(b_hat != u, axis=1))
    assert 0 < bler[("flooding", 16)] < 0.5
    assert np.isclose(bler[("layered", 8)], bler[("flooding", 16)], rtol=0.7)
    assert bler[("flooding", 8)] > bler[("layered", 8)]


@pytest.mark.parametrize("k,n,bg,m", [(2816, 8448, "bg1", 6), (1280, 3840, "bg2", 4), (2816, 5632, "bg1", None),
                                      (1408, 4224, "bg1", None), (1920, 5760, "bg2", 2),           # Z = 128, 128, 128, 64, 192
                                      (768, 1536, None, 2), (1024, 2048, "bg1", None), (480, 1440, None, 2), (200, 600, None, 2),   # Z = 80, 48, 60, 26
                                      (3000, 6000, "bg1", None)])                                  # Z = 144 (partly filled chunks)
def test_layered_on_chip_bit_exact(phy, k, n, bg, m):
    """cn_schedule="layered" on the on-chip layered engine (csrc/ldpc5g_onchip_ly.hip: c2v and variable-node totals in
    LDS, one kernel for the whole decode) - min-sum and offset-min-sum soft outputs bit for bit against the oracle's
    literal form (check-node update of the layer, then EVERY variable node) and against the HBM-resident scheduled engine;
    codeword output, hard output and a grid of few workgroups (several codewords per workgroup) included."""
    import os
    code = LDPC5GCode(k, n, m, bg)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    u, c, llr = _noisy_llr(code, 6, k + 3 * n, sigma=0.7)
    llr[0, :9] = 0
    llr[1] = np.round(llr[1])
    for cn in ("minsum", "offset-minsum", "boxplus-phi"):
        for it, infobits in ((1, True), (3, False)):
            kw = dict(cn_update=cn, cn_schedule="layered", num_iter=it, hard_out=False, return_infobits=infobits)
            dec = phy.fec.ldpc.LDPC5GDecoder(enc, **kw)
            from sionna_amd import _ffi
            assert _ffi.lib().samd_ldpc5g_decode_layered_supported(enc._handle(dec._nb_pruned_nodes), dec._cn_mode) == 1
            got = _np(dec(llr))
            ref = obp.LDPC5GDecoder(code, **kw).decode5g(llr)
            assert np.array_equal(got, ref), f"{cn} it={it}: {np.mean(got != ref):.3e} differ, max {np.max(np.abs(got - ref))}"
            with _ffi_option("SAMD_NO_ONCHIP_LAYERED"):
                assert np.array_equal(_np(dec(llr)), ref)                  # the scheduled HBM-resident engine
    big = np.tile(llr, (11, 1))
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", cn_schedule="layered", num_iter=4)
    full = _np(dec(big))
    with _ffi_option("SAMD_ONCHIP_GRID", 5):
        assert np.array_equal(_np(dec(big)), full)
    assert np.array_equal(full[:6], full[6:12])


# ------------------------------------------------------------------ 5G decoder (both engines)
CODES5G = [(64, 128, None, None), (200, 600, None, 2), (1024, 2048, "bg1", None), (500, 1000, None, 4),
           (2816, 8448, "bg1", 6),
           (4224, 12672, "bg1", None), (7040, 14080, None, 4),      # on-chip with the channel LLRs in L2 (workspace)
           (8448, 16896, None, 2)]                                  # ... and the VN totals as well


def _noisy_llr(code, batch, seed, sigma=0.8):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 2, (batch, code.k)).astype(np.float32)
    c = code.encode(u)
    y = (2 * c - 1) + sigma * rng.normal(size=c.shape)
    return u, c, (2 * y / sigma ** 2).astype(np.float32)


@pytest.mark.parametrize("k,n,bg,m", CODES5G)
@pytest.mark.parametrize("cn", ["minsum", "offset-minsum"])
def test_5g_minsum_bit_exact_both_engines(phy, k, n, bg, m, cn):
    code = LDPC5GCode(k, n, m, bg)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    u, c, llr = _noisy_llr(code, 13, k + n)
    llr[0, :7] = 0
    llr[1] = np.round(llr[1])
    for it, infobits in ((0, False), (1, False), (7, True), (20, False)):
        odec = obp.LDPC5GDecoder(code, cn_update=cn, hard_out=False, return_infobits=infobits, num_iter=it)
        l5 = odec.rate_recover(llr)
        xr = cbind.bp_decode(odec, l5, num_iter=it, hard_out=0)
        ref = xr[:, :k] if infobits else None
        if not infobits:      # map like decoding.py:1506-1531
            x_nf = np.concatenate([xr[:, :k], xr[:, code.k_ldpc:]], axis=1)
            ref = x_nf[:, 2 * code.z:2 * code.z + n]
            if m is not None:
                ref = ref[:, code.out_int]
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, hard_out=False, return_infobits=infobits, num_iter=it)
        got_onchip = _np(dec(llr))
        assert dec._onchip_ok, "on-chip engine should accept this code"
        assert np.array_equal(got_onchip, ref), f"on-chip {cn} it={it}"
        with _ffi_option("SAMD_ONCHIP_COMPRESSED"):             # compressed check-node state engine (every code size)
            got_compressed = _np(dec(llr))
        assert dec._onchip_ok and np.array_equal(got_compressed, ref), f"on-chip (compressed state) {cn} it={it}"
        dec._onchip_ok = False                                  # force the HBM-resident engine
        got_generic = _np(dec(llr))
        assert np.array_equal(got_generic, ref), f"generic {cn} it={it}"
    hard = _np(phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn)(llr))
    odec = obp.LDPC5GDecoder(code, cn_update=cn, hard_out=True, num_iter=20)
    assert np.array_equal(hard, cbind.bp_decode(odec, odec.rate_recover(llr))[:, :k])


# random (k, n) pairs (seeded draw, fillers, odd lifting sizes, partial last base rows, rates 0.2 ... 0.9)
RANDOM_CODES = [(7202, 11115), (1544, 2929), (711, 1585), (3155, 4187), (5962, 17392), (7650, 11644), (867, 988),
                (2548, 3019), (2415, 3322), (5386, 9607), (5459, 10623), (6984, 15975), (1362, 3799), (2376, 4182),
                (810, 1219), (3663, 17527), (3956, 8682), (3804, 11295), (4055, 8034), (5041, 12295),
                (1840, 3054), (92, 207), (139, 286), (2003, 7522), (188, 241), (1582, 2468), (49, 105), (2940, 3216),
                (1191, 3751), (3068, 9642), (156, 340), (204, 716), (66, 87), (36, 73), (24, 44), (38, 168),
                (2125, 6061), (2115, 5342), (1994, 3815), (974, 1068), (75, 113), (37, 68), (58, 76), (43, 81)]


@pytest.mark.parametrize("k,n", RANDOM_CODES)
def test_5g_random_codes_all_engines(phy, k, n):
    """Whatever engine the library picks for a code (explicit messages, compressed state with part of it in L2,
    HBM-resident): min-sum equals the oracle bit for bit, boxplus-phi on chip equals the HBM engine."""
    code = LDPC5GCode(k, n)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    u, c, llr = _noisy_llr(code, 5, k ^ n, sigma=0.7)
    assert np.array_equal(_np(enc(u)), c)
    odec = obp.LDPC5GDecoder(code, cn_update="minsum", hard_out=False, num_iter=6)
    ref = cbind.bp_decode(odec, odec.rate_recover(llr), num_iter=6, hard_out=0)[:, :k]
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", hard_out=False, num_iter=6)
    assert np.array_equal(_np(dec(llr)), ref) and dec._onchip_ok
    with _ffi_option("SAMD_ONCHIP_COMPRESSED"):
        assert np.array_equal(_np(dec(llr)), ref)
    # explicit messages with the last rows in L2, also beyond the size range where the library selects it
    # (the handle caches its tables, so a fresh encoder / handle is built under the flag)
    with _ffi_option("SAMD_FORCE_SPILL"):
        enc2 = phy.fec.ldpc.LDPC5GEncoder(k, n)
        dec2 = phy.fec.ldpc.LDPC5GDecoder(enc2, cn_update="minsum", hard_out=False, num_iter=6)
        assert np.array_equal(_np(dec2(llr)), ref)
    dec._onchip_ok = False
    assert np.array_equal(_np(dec(llr)), ref)
    # the kernel generated for the code (round 6: any even lifting size whose messages fit LDS), small batch forced onto it
    from sionna_amd import _ffi
    with _ffi_option("SAMD_LDPC_JIT", "2"):
        enc3 = phy.fec.ldpc.LDPC5GEncoder(k, n)
        dec3 = phy.fec.ldpc.LDPC5GDecoder(enc3, cn_update="minsum", hard_out=False, num_iter=6)
        h3 = enc3._handle(dec3._nb_pruned_nodes)
        if _ffi.lib().samd_ldpc5g_jit_supported(h3):
            assert np.array_equal(_np(dec3(llr)), ref)
            assert _ffi.lib().samd_ldpc5g_jit_launches(h3) > 0, "generated kernel did not run"
            dec4 = phy.fec.ldpc.LDPC5GDecoder(enc3, cn_update="offset-minsum", hard_out=True, return_infobits=False, num_iter=3)
            od4 = obp.LDPC5GDecoder(code, cn_update="offset-minsum", hard_out=True, return_infobits=False, num_iter=3)
            assert np.array_equal(_np(dec4(llr)), od4.decode5g(llr))
        else:
            # no generated kernel only for an odd lifting size, or when the messages exceed LDS (then the generic engine is not
            # the explicit-message one either: samd_ldpc5g_decode_engine = 2)
            assert enc3.z % 2 == 1 or _ffi.lib().samd_ldpc5g_decode_engine(h3, _ffi.CN_MODES["minsum"]) != 2, (k, n, enc3.z)
    for cn, infobits in (("boxplus-phi", True), ("boxplus", False)):
        decp = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, hard_out=False, return_infobits=infobits, num_iter=4)
        a = _np(decp(llr))
        if decp._onchip_ok:
            decp._onchip_ok = False
            assert np.array_equal(a, _np(decp(llr))), cn
    # codeword output of the min-sum engines (marginals of the fused degree-1 columns)
    odec = obp.LDPC5GDecoder(code, cn_update="offset-minsum", hard_out=True, return_infobits=False, num_iter=3)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="offset-minsum", hard_out=True, return_infobits=False, num_iter=3)
    assert np.array_equal(_np(dec(llr)), odec.decode5g(llr))


@pytest.mark.parametrize("k,n,bg,m", CODES5G)
@pytest.mark.parametrize("cn", ["boxplus-phi", "boxplus"])
def test_5g_boxplus_onchip_equals_hbm_engine(phy, k, n, bg, m, cn):
    """The on-chip boxplus engine (one float per edge in LDS) uses the arithmetic and the summation order of the
    HBM-resident engine: the two return the same bits (which test_5g_boxplus_vs_oracle holds to the oracle)."""
    code = LDPC5GCode(k, n, m, bg)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    u, c, llr = _noisy_llr(code, 11, 2 * k + n, sigma=0.6)
    llr[0, :9] = 0
    llr[1] = np.round(llr[1])
    llr[2] = 0                                                   # all-erasure word stays all-zero
    ran_onchip = False
    for it, infobits in ((0, False), (1, True), (3, False), (20, True)):
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, hard_out=False, return_infobits=infobits, num_iter=it)
        a = _np(dec(llr))
        if not dec._onchip_ok:                                   # messages do not fit in LDS: HBM engine only
            continue
        ran_onchip = True
        dec._onchip_ok = False
        b = _np(dec(llr))
        assert np.array_equal(a, b), f"{cn} it={it} infobits={infobits}: max diff {np.max(np.abs(a - b))}"
        assert np.all(a[2] == 0)
    if k <= 4096:
        assert ran_onchip


@pytest.mark.parametrize("k,n,bg,m", CODES5G)
def test_5g_boxplus_phi_bit_exact_vs_oracle(phy, k, n, bg, m):
    """boxplus-phi, the reference's DEFAULT rule (decoding.py:1045-1166): phi(x) = log(e^x+1) - log(e^x-1) is evaluated
    literally in float32 on a DEFINED exp / log - the Cephes / Eigen algorithm TensorFlow-CPU's kernels are built on, one
    fixed sequence of IEEE operations (oracle/ldpc_bp.c == csrc/bp_math.h).  Every engine (on-chip explicit messages,
    last rows in L2, HBM-resident) therefore returns the oracle's soft outputs bit for bit, well conditioned and
    saturating alike (round 2 held this rule to "95 % of the outputs within 1e-5")."""
    code = LDPC5GCode(k, n, m, bg)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    for sigma, its in ((0.9, (1,)), (0.55, (1, 10, 20))):
        u, c, llr = _noisy_llr(code, 8, k, sigma=sigma)
        llr[0, :5] = 0
        llr[1] = np.round(llr[1])
        for it in its:
            odec = obp.LDPC5GDecoder(code, cn_update="boxplus-phi", hard_out=False, num_iter=it)
            ref = cbind.bp_decode(odec, odec.rate_recover(llr), num_iter=it, hard_out=0)[:, :k]
            dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="boxplus-phi", hard_out=False, num_iter=it)
            got = _np(dec(llr))
            assert np.array_equal(got, ref), f"sigma={sigma} it={it}: max diff {np.max(np.abs(got - ref))}"
            if dec._onchip_ok:                                     # ... and the HBM-resident engine on the same input
                dec._onchip_ok = False
                assert np.array_equal(_np(dec(llr)), ref), f"HBM engine sigma={sigma} it={it}"
    assert np.array_equal(_np(phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="boxplus-phi", return_infobits=False)(llr)), c)


@pytest.mark.parametrize("k,n,bg,m", CODES5G)
@pytest.mark.parametrize("cn", ["boxplus-phi-fast", "boxplus"])
def test_5g_boxplus_vs_oracle(phy, k, n, bg, m, cn):
    """The rules on the GPU's transcendental unit / libm: the tanh rule (tanhf, atanhf) and "boxplus-phi-fast" (v_exp_f32,
    v_log_f32: ~1 ulp with unspecified last bits; phi amplifies them on saturating messages, DESIGN.md "phi
    conditioning").  Bars:
      * well-conditioned regime (low SNR, 1 iteration): the north-star bar 1e-5 relative (+1e-4 absolute floor) on
        >= 99.9 % of the outputs;
      * saturating regime: hard decisions identical, >= 90 % of the soft outputs within the bar, none further than 0.5
        from the oracle (measured: profiles/r02_phi_scale.json)."""
    code = LDPC5GCode(k, n, m, bg)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    ocn = "boxplus-phi" if cn == "boxplus-phi-fast" else cn
    u, c, llr = _noisy_llr(code, 8, k, sigma=0.9)
    odec = obp.LDPC5GDecoder(code, cn_update=ocn, hard_out=False, return_infobits=True, num_iter=1)
    got = _np(phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, hard_out=False, num_iter=1)(llr))
    _close(got, odec.decode5g(llr), f"{cn} well-conditioned")
    u, c, llr = _noisy_llr(code, 8, k, sigma=0.55)
    for it in (1, 10):
        odec = obp.LDPC5GDecoder(code, cn_update=ocn, hard_out=False, return_infobits=True, num_iter=it)
        ref = cbind.bp_decode(odec, odec.rate_recover(llr), num_iter=it, hard_out=0)[:, :k]
        got = _np(phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, hard_out=False, num_iter=it)(llr))
        assert np.mean(np.isclose(got, ref, rtol=1e-5, atol=1e-4)) >= 0.9, f"{cn} it={it}"
        assert np.max(np.abs(got - ref)) <= 0.5, f"{cn} it={it}: {np.max(np.abs(got - ref))}"
        assert np.array_equal(got > 0, ref > 0)
    assert np.array_equal(_np(phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, return_infobits=False)(llr)), c)


@pytest.mark.parametrize("ebno", [4.0, 4.5])
def test_c2_boxplus_phi_at_scale_bit_exact(phy, ebno):
    """The reference's DEFAULT rule at BASELINE config C2 scale (n=8448, k=2816, 64-QAM, 20 iterations, 2048 codewords
    in the waterfall) against the C oracle on the same LLRs: soft outputs array_equal (round 2: 95.1 % within 1e-5 at
    4.0 dB, because exp / log came from two different libms; now both sides follow one defined arithmetic).  The
    hardware-transcendental variant keeps the round-2 bars."""
    k, n, m, B = 2816, 8448, 6, 2048
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    code = LDPC5GCode(k, n, m, "bg1")
    phy.config.seed = int(ebno * 100)
    no = phy.utils.ebnodb2no(ebno, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
    got = _np(phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="boxplus-phi", num_iter=20, hard_out=False)(llr))
    odec = obp.LDPC5GDecoder(code, cn_update="boxplus-phi", num_iter=20, hard_out=False)
    ref = cbind.bp_decode(odec, odec.rate_recover(_np(llr)))[:, :k]
    assert np.array_equal(got, ref), f"{np.mean(got != ref):.3e} of the soft outputs differ, max {np.max(np.abs(got - ref))}"
    fast = _np(phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="boxplus-phi-fast", num_iter=20, hard_out=False)(llr))
    hg, hr, ub = fast > 0, ref > 0, _np(u) > 0
    decoded = np.all(hr == ub, axis=1)
    assert decoded.any() and np.array_equal(hg[decoded], hr[decoded])
    assert np.mean(np.all(hg == hr, axis=1)) >= 0.995
    assert abs(int(np.any(hg != ub, axis=1).sum()) - int(np.any(hr != ub, axis=1).sum())) <= 2
    assert np.mean(np.isclose(fast, ref, rtol=1e-5, atol=1e-4)) >= (0.9 if ebno < 4.25 else 0.99)


@pytest.mark.parametrize("ebno", [4.0])
def test_c2_minsum_soft_outputs_at_scale_bit_exact(phy, ebno):
    """The headline kernel held to the oracle DIRECTLY at scale (round-2 verdict, test gap): C2, min-sum, 20 iterations,
    2048 codewords in the waterfall through a grid of 64 workgroups (SAMD_ONCHIP_GRID), so every workgroup decodes 32
    codewords in sequence - the grid-stride loop and the reuse of its workspace row - and the soft outputs are
    array_equal to oracle/ldpc_bp.c; offset min-sum on the same batch."""
    import os
    k, n, m, B = 2816, 8448, 6, 2048
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    code = LDPC5GCode(k, n, m, "bg1")
    phy.config.seed = 4242
    no = phy.utils.ebnodb2no(ebno, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
    for cn in ("minsum", "offset-minsum"):
        odec = obp.LDPC5GDecoder(code, cn_update=cn, num_iter=20, hard_out=False)
        ref = cbind.bp_decode(odec, odec.rate_recover(_np(llr)))[:, :k]
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=20, hard_out=False)
        assert dec._onchip_ok
        for grid in (None, "64"):
            if grid:
                with _ffi_option("SAMD_ONCHIP_GRID", grid):
                    got = _np(dec(llr))
            else:
                got = _np(dec(llr))
            assert np.array_equal(got, ref), f"{cn} grid={grid}: {np.mean(got != ref):.3e} differ"
    frac_err = np.mean(np.any((ref > 0) != (_np(u) > 0), axis=1))
    assert 0.0 < frac_err < 1.0                                     # the waterfall: some words fail, some decode


def test_5g_large_z_rate_third_on_chip(phy):
    # Z=384, rate 1/3: only (M1, M2) stay in LDS, LLRs / VN totals / sign words live in the L2 workspace
    k, n = 8448, 25344
    code = LDPC5GCode(k, n)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    u, c, llr = _noisy_llr(code, 3, 1, sigma=0.5)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", hard_out=False, num_iter=5)
    odec = obp.LDPC5GDecoder(code, cn_update="minsum", hard_out=False, num_iter=5)
    got = _np(dec(llr))
    assert dec._onchip_ok and dec._ws is not None
    assert np.array_equal(got, cbind.bp_decode(odec, odec.rate_recover(llr))[:, :k])


# ------------------------------------------------------------------ metrics + sim_ber + chain
def test_count_errors(phy):
    rng = np.random.default_rng(0)
    b = rng.integers(0, 2, (1000, 257)).astype(np.float32)
    bh = b.copy()
    flip = rng.random(b.shape) < 0.001
    bh[flip] = 1 - bh[flip]
    assert int(phy.utils.count_errors(torch.tensor(b).cuda(), torch.tensor(bh).cuda())) == int(flip.sum())
    assert int(phy.utils.count_block_errors(torch.tensor(b).cuda(), torch.tensor(bh).cuda())) == int(flip.any(1).sum())


def _chain(phy, k, n, m, bg, cn, num_iter):
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=num_iter)
    src, mapper, awgn = phy.mapping.BinarySource(), phy.mapping.Mapper("qam", m), phy.channel.AWGN()
    demap = phy.mapping.Demapper("app", "qam", m)

    def mc_fun(batch_size, ebno_db):
        no = phy.utils.ebnodb2no(ebno_db, m, k / n)
        u = src([batch_size, k])
        llr = demap(awgn(mapper(enc(u)), no), no)
        return u, dec(llr)
    return mc_fun


def test_c1_chain_matches_oracle_on_same_noise(phy):
    """Config C1 (QPSK, BG1 k=1024 n=2048, BP-10 boxplus-phi, B=32): whole chain on the GPU vs
    the oracle chain on the SAME Philox noise: identical bit errors up to LLR ties."""
    k, n, m = 1024, 2048, 2
    code = LDPC5GCode(k, n, m, "bg1")
    pts = omap.qam(m)
    for seed, ebno in ((1, 0.0), (2, 1.5), (3, 3.0)):
        phy.config.seed = seed
        u, u_hat = _chain(phy, k, n, m, "bg1", "boxplus-phi", 10)(32, ebno)
        no = outil.ebnodb2no(ebno, m, k / n)
        uo = outil.random_bits(seed, 0, 32 * k).reshape(32, k)
        assert np.array_equal(_np(u), uo)
        y = outil.awgn(omap.mapper(code.encode(uo), pts), no, seed, 1)
        llr = omap.demapper(y, no, pts, "app")
        ref = obp.LDPC5GDecoder(code, num_iter=10).decode5g(llr)
        mism = np.mean(_np(u_hat) != ref)
        assert mism < 2e-3, f"seed {seed}: {mism}"
        assert abs(np.mean(_np(u_hat) != uo) - np.mean(ref != uo)) < 2e-3


def test_sim_ber_runs_and_stops(phy):
    phy.config.seed = 5
    ber, bler = phy.utils.sim_ber(_chain(phy, 200, 400, 2, None, "minsum", 10), np.array([0., 3., 6., 9.]),
                                  batch_size=200, max_mc_iter=3, num_target_block_errors=50, verbose=False)
    ber = ber.numpy()
    assert ber[0] > 0.05 and ber[-1] == 0 and np.all(np.diff(ber) <= 1e-9)


# ------------------------------------------------------------------ full-size properties (C2)
def test_c2_full_batch_properties(phy):
    """BASELINE config C2 at a large batch: size-independent properties."""
    k, n, m = 2816, 8448, 6
    B = 4096
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    phy.config.seed = 9
    src = phy.mapping.BinarySource()
    u1, u2 = src([B, k]), src([B, k])
    c1, c2 = enc(u1), enc(u2)
    # linearity over GF(2)
    c12 = enc(((u1 + u2) % 2))
    assert torch.equal(c12, (c1 + c2) % 2)
    # noiseless round trip through mapper / demapper / both decoder engines
    mapper, demap = phy.mapping.Mapper("qam", m), phy.mapping.Demapper("app", "qam", m)
    llr = demap(mapper(c1), 0.05)
    assert torch.equal(hard := (llr > 0).float(), c1), "demapper hard decisions must reproduce the codeword"
    for cn in ("minsum", "boxplus-phi"):
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=20)
        assert torch.equal(dec(llr).as_subclass(torch.Tensor), u1.as_subclass(torch.Tensor))
    # erasures: puncture 25 % of the LLRs, decoder still recovers (rate 1/3)
    mask = (torch.rand_like(llr) < 0.25)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=20)
    u_hat = dec(torch.where(mask, torch.zeros_like(llr), llr))
    assert float((u_hat != u1).float().mean()) < 1e-4
    # on-chip == generic on the full batch
    dec2 = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=5, hard_out=False)
    noisy = llr * 0.05 + torch.randn_like(llr)
    a = dec2(noisy)
    dec2._onchip_ok = False
    b = dec2(noisy)
    assert torch.equal(a.as_subclass(torch.Tensor), b.as_subclass(torch.Tensor))


# ------------------------------------------------------------------ extension points (custom.py) on the device
def test_decoder_callbacks_and_custom_updates_on_device(phy):
    """Message callbacks / callable node updates run the torch engine on device tensors: with pass-through callbacks
    the result equals the HIP engine's (min-sum: to float32 rounding - segment sums may associate differently),
    exported rule functions select the HIP engine, statistics callbacks see every iteration."""
    from sionna_amd.phy.fec.ldpc import (cn_update_minsum, cn_update_phi, DecoderStatisticsCallback, EXITCallback,
                                         WeightedBPCallback)
    k, n = 400, 800
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    code = LDPC5GCode(k, n)
    u, c, llr = _noisy_llr(code, 64, 5, sigma=0.55)
    plain = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", hard_out=False, num_iter=12)
    ref = _np(plain(llr))
    by_fn = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn_update_minsum, hard_out=False, num_iter=12)
    assert not by_fn._custom and np.array_equal(_np(by_fn(llr)), ref)
    stats = DecoderStatisticsCallback(12)
    seen = []
    cb = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", hard_out=False, num_iter=12, c2v_callbacks=[stats],
                                    v2c_callbacks=[lambda m, it, x_hat: (seen.append((it, m.flat_values.is_cuda)), m)[1]])
    assert cb._custom
    got = _np(cb(llr))
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-4)
    assert seen == [(i, True) for i in range(0, 13)] and np.all(stats.num_samples == 64)
    # (this rate-matched code keeps punctured degree-1 parity nodes with LLR 0: their check nodes never count as
    # satisfied, so the convergence statistic is exercised on a regular code below)
    reg = phy.fec.utils.load_parity_check_examples(3)[0]
    st2 = DecoderStatisticsCallback(10)
    y0 = -(3.0 + 1.5 * torch.randn(256, reg.shape[1], device="cuda"))          # all-zero codeword: logits < 0
    phy.fec.ldpc.LDPCBPDecoder(reg, cn_update="minsum", num_iter=10, c2v_callbacks=[st2])(y0)
    assert np.all(st2.num_samples == 256) and np.all(np.diff(st2.success_rate) >= 0) and st2.success_rate[-1] > 0.5, st2.success_rate
    assert 0 < st2.avg_number_iterations < 10
    # state out / in on the custom path, hard output, generic decoder with a scaled min-sum written by the user
    cbs = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", hard_out=True, num_iter=3, return_state=True,
                                     c2v_callbacks=[lambda m, it: m])
    hb, st = cbs(llr)
    hb2, _ = cbs(llr, msg_v2c=st)
    assert tuple(st.shape) == (cbs.num_edges, 64) and np.array_equal(_np(hb2), (ref > 0).astype(np.float32))
    pcm = phy.fec.utils.load_parity_check_examples(1)[0]
    scaled = lambda msg, llr_clipping=None: cn_update_minsum(msg, llr_clipping) * 0.75
    dec = phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update=scaled, hard_out=False, num_iter=5)
    y = torch.randn(32, pcm.shape[1], device="cuda") * 2 - 3
    out = dec(y)
    assert tuple(out.shape) == (32, pcm.shape[1]) and bool(torch.isfinite(out.as_subclass(torch.Tensor)).all())
    # weighted BP: gradient reaches the trainable edge weights through the device engine
    w = WeightedBPCallback(cb.num_edges)
    wdec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn_update_phi, hard_out=False, num_iter=3, v2c_callbacks=[w])
    assert wdec._custom
    soft = wdec(torch.as_tensor(llr).cuda()).as_subclass(torch.Tensor)
    loss = torch.nn.functional.binary_cross_entropy_with_logits(soft, torch.as_tensor(u).cuda())
    loss.backward()
    assert w.weights.grad is not None and float(w.weights.grad.abs().sum()) > 0
    mi = EXITCallback(3)
    phy.fec.ldpc.LDPC5GDecoder(enc, num_iter=3, v2c_callbacks=[mi])(-4.0 - 2 * torch.randn(16, n, device="cuda"))
    assert np.all(np.isfinite(mi.mi))           # mi[0] = the initial messages (decoding.py:583-594)


# ---------------------------------------------------------------------------------------------------------------------
# The HIP decoders / encoder held DIRECTLY to outputs of the reference's own decoding.py / encoding.py executed under the
# NumPy stand-in for TensorFlow (tests/golden/ldpc_bp_ref_golden.npz, tools/gen_ldpc_bp_golden.py) - not via the oracle.
# The boxplus rules there ran on NumPy's float32 exp / log / tanh: an arithmetic INDEPENDENT of csrc/bp_math.h.
# ---------------------------------------------------------------------------------------------------------------------
_REFX = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "ldpc_bp_ref_golden.npz")))
# ... and the BASELINE.json codes themselves through the executed reference (tools/gen_ldpc_bp_golden.py --baseline)
_REFX.update(np.load(os.path.join(os.path.dirname(__file__), "golden", "ldpc_bp_ref_golden_baseline.npz")))


def _refx_sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


@pytest.mark.parametrize("i", range(5))
def test_generic_decoder_matches_reference_execution(phy, i):
    """LDPCBPDecoder._bp_iter loops of the reference on the five example parity-check matrices: min-sum family soft outputs
    and decoder state bit for bit (state by SHA-256), hard decisions and a resumed decode (msg_v2c) too; boxplus (tanh)
    within 1e-5 (+1e-4 floor) on >= 99.9 % of the outputs, boxplus-phi the same for one iteration and >= 90 % / all signs /
    maximum 0.5 after five (the reference side ran NumPy's exp / log: an arithmetic independent of csrc/bp_math.h)."""
    g = _REFX
    pcm, llr = _example_pcm(i), g[f"bp_ex{i}_llr"]
    for rule in ("minsum", "offset-minsum", "boxplus", "boxplus-phi"):
        for it in ((1, 5) if i != 4 else (5,)):
            x, st = phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update=rule, hard_out=False, num_iter=it, return_state=True)(llr)
            ref = g[f"bp_ex{i}_{rule}_it{it}_x"]
            if rule in ("minsum", "offset-minsum"):
                assert np.array_equal(_np(x), ref), (rule, it)
                assert np.array_equal(_refx_sha(_np(st)), g[f"bp_ex{i}_{rule}_it{it}_state_sha"]), (rule, it)
            elif it == 1:
                _close(_np(x), ref, f"{rule} it={it} vs reference execution")
            elif rule == "boxplus":
                # tanhf / atanhf of the device library against NumPy's after five iterations: measured 95.6 ... 100 %
                # within the bar, maximum 0.023 (profiles/r04_refexec_gpu_bars.txt)
                got = _np(x)
                assert np.mean(np.isclose(got, ref, rtol=1e-5, atol=1e-4)) >= 0.93 and np.max(np.abs(got - ref)) <= 0.1, (rule, it)
            else:
                # five iterations of phi on two different exp / log: BP amplifies last-bit differences where
                # phi(sum - phi_self) cancels (DESIGN.md "phi conditioning"); the defined form itself sits at 0.915 ... 1.0
                # within the bar against the reference's NumPy-libm run on these matrices, maximum 0.22
                got = _np(x)
                assert np.mean(np.isclose(got, ref, rtol=1e-5, atol=1e-4)) >= 0.9 and np.max(np.abs(got - ref)) <= 0.5, (rule, it)
                assert np.array_equal((got > 0)[np.abs(ref) > 1e-2], (ref > 0)[np.abs(ref) > 1e-2])
    d = phy.fec.ldpc.LDPCBPDecoder(pcm, cn_update="minsum", hard_out=True, num_iter=3, return_state=True)
    x1, st1 = d(llr)
    x2, st2 = d(llr, msg_v2c=st1)
    assert np.array_equal(_np(x1).astype(np.uint8), g[f"bp_ex{i}_minsum_hard3"])
    assert np.array_equal(_np(x2).astype(np.uint8), g[f"bp_ex{i}_minsum_hard3_resumed"])
    want = g[f"bp_ex{i}_minsum_resumed_state"]
    assert np.array_equal(_np(st2) if i < 2 else _refx_sha(_np(st2)), want)


@pytest.mark.parametrize("tag", ["c1", "bg2s", "bg2m", "bg1r", "c2", "c4"])
def test_5g_chain_matches_reference_execution(phy, tag):
    """LDPC5GEncoder -> LDPC5GDecoder of the reference (C1 = BG1 k=1024 n=2048 BP-10; BG2 small / with the output
    interleaver; BG1 with rate matching): codewords bit for bit; min-sum family soft outputs (return_infobits=False),
    decoder state and decisions bit for bit, flooding and layered; boxplus rules: decisions identical on the words the
    reference decodes, all signs equal, soft outputs within 1e-5 (+1e-4) on >= 93 % for the boxplus rules (words that do not
    converge amplify last bits; the reference side ran NumPy's exp / log / tanh)."""
    g = _REFX
    k, n, bg, z, m, iters = (int(v) for v in g[f"g5_{tag}_meta"])
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=(m or None), bg=f"bg{bg}")
    assert enc.z == z
    u, llr = g[f"g5_{tag}_u"], g[f"g5_{tag}_llr"]
    assert np.array_equal(_np(enc(u.astype(np.float32))), g[f"g5_{tag}_c"])
    D = phy.fec.ldpc.LDPC5GDecoder
    for rule in ("minsum", "offset-minsum", "boxplus", "boxplus-phi"):
        x, st = D(enc, cn_update=rule, hard_out=False, return_infobits=False, num_iter=iters, return_state=True)(llr)
        uh = _np(D(enc, cn_update=rule, hard_out=True, return_infobits=True, num_iter=iters)(llr)).astype(np.uint8)
        ref, uref = g[f"g5_{tag}_{rule}_x"], g[f"g5_{tag}_{rule}_uhat"]
        if rule in ("minsum", "offset-minsum"):
            assert np.array_equal(_np(x), ref), rule
            assert np.array_equal(_refx_sha(_np(st)), g[f"g5_{tag}_{rule}_state_sha"]), rule
            assert np.array_equal(uh, uref), rule
        else:
            # measured for the defined phi against this fixture (CPU, oracle/ldpc_bp.c): 95.7 ... 99.98 % within the bar,
            # all signs equal, maximum 2.08 (one saturation step of phi on a word that does not converge)
            conv = np.all(uref == u, axis=1)
            got = _np(x)
            assert conv.sum() >= 2 and np.array_equal(uh[conv], uref[conv]), rule
            # (tanh rule on the device library's tanhf / atanhf: measured 95.9 ... 98.0 % on the GPU)
            assert np.mean(np.isclose(got, ref, rtol=1e-5, atol=1e-4)) >= 0.93, rule
            assert np.max(np.abs(got - ref)) <= 2.5 and np.array_equal((got > 0)[np.abs(ref) > 1e-2], (ref > 0)[np.abs(ref) > 1e-2]), rule
    if True:
        # BASELINE's own codes (C1, C2, C4) and the other fixtures: the kernel GENERATED for the code (csrc/ldpc5g_jit.cpp;
        # SAMD_LDPC_JIT=2 = also for this batch of 4; round 6: every even lifting size) against the executed reference, both
        # output forms, both rules
        from sionna_amd import _ffi
        with _ffi_option("SAMD_LDPC_JIT", "2"):
            enc_j = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=(m or None), bg=f"bg{bg}")
            for rule in ("minsum", "offset-minsum"):
                dj = D(enc_j, cn_update=rule, hard_out=False, return_infobits=False, num_iter=iters)
                xj = _np(dj(llr))
                assert _ffi.lib().samd_ldpc5g_jit_launches(enc_j._handle(dj._nb_pruned_nodes)) > 0, "specialised kernel did not run"
                assert np.array_equal(xj, g[f"g5_{tag}_{rule}_x"]), rule
                uj = _np(D(enc_j, cn_update=rule, hard_out=True, return_infobits=True, num_iter=iters)(llr)).astype(np.uint8)
                assert np.array_equal(uj, g[f"g5_{tag}_{rule}_uhat"]), rule
    if tag in ("c1", "bg2s", "c2"):
        it = max(2, iters // 2)
        x = D(enc, cn_update="minsum", hard_out=False, return_infobits=False, num_iter=it, cn_schedule="layered")(llr)
        assert np.array_equal(_np(x), g[f"g5_{tag}_layered_minsum_x"])
        x = D(enc, cn_update="boxplus-phi", hard_out=False, return_infobits=False, num_iter=it, cn_schedule="layered")(llr)
        assert np.mean(np.isclose(_np(x), g[f"g5_{tag}_layered_phi_x"], rtol=1e-5, atol=1e-4)) >= 0.93


@pytest.mark.parametrize("m", [1, 2, 4, 6])
def test_symbol_logits2llrs_block(phy, m):
    """phy.mapping.SymbolLogits2LLRs (samd_symbol_logits2llrs_f32) against the reference's own block executed under the
    NumPy stand-in (tests/golden/phy_ref_golden.npz) and the float64 oracle: app / maxlog, priors per row and shared."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "phy_ref_golden.npz"))
    z, pr, pv = g[f"l2l{m}_z"], g[f"l2l{m}_prior"], g[f"l2l{m}_prior_vec"]
    for meth in ("app", "maxlog"):
        blk = phy.mapping.SymbolLogits2LLRs(meth, m)
        for key, prior in (("", None), ("_prior", pr), ("_prior_vec", pv)):
            ref = g[f"l2l{m}_{meth}{key}"]
            got = _np(blk(z) if prior is None else blk(z, prior))
            assert got.shape == ref.shape and np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (meth, key)
            assert np.abs(got - omap.symbol_logits2llrs(z, m, meth, prior)).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    hard = _np(phy.mapping.SymbolLogits2LLRs("app", m, hard_out=True)(z, pr))
    assert np.array_equal(hard.astype(np.uint8), g[f"l2l{m}_hard"])
    assert tuple(phy.mapping.SymbolLogits2LLRs("app", m)(np.zeros((0, 1 << m), np.float32)).shape) == (0, m)
