"""CPU tests that close pinning holes of the oracle:

(i)   oracle/ldpc_bp.c (the fast checker behind every "bit-exact on 5G codes" GPU assertion, and the CPU baseline) is
      compared DIRECTLY with oracle/ldpc_bp.py (the literal restatement pinned in tests/test_oracle_ldpc.py);
(ii)  oracle/mapping.demapper against the formula of the reference's own NumPy test, scipy ``logsumexp`` over the
      point sets C_{i,1} / C_{i,0} (/root/reference/test/unit/mapping/test_mapping.py:175-199, atol 1e-5);
(iii) ebnodb2no with a resource grid: host function == oracle == the formula of utils/misc.py:225-251 by hand."""
import numpy as np
import pytest
from scipy.special import logsumexp

from oracle.ldpc5g import LDPC5GCode
from oracle import ldpc_bp as bp, cbind, mapping as omap, utils as outil, ofdm as o


def _noisy_llr(code, m, batch, ebno_db, seed):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 2, (batch, code.k)).astype(np.float32)
    c = code.encode(u)
    no = float(outil.ebnodb2no(ebno_db, m, code.k / code.n))
    pts = omap.qam(m)
    x = omap.mapper(c, pts)
    y = (x + np.sqrt(no / 2) * (rng.normal(size=x.shape) + 1j * rng.normal(size=x.shape))).astype(np.complex64)
    return u, omap.demapper(y, np.float32(no), pts, "app")


@pytest.mark.parametrize("k,n,m,bg,batch,iters,ebno", [
    (1024, 2048, 2, "bg1", 32, 10, 1.5),           # BASELINE config C1
    (2816, 8448, 6, "bg1", 4, 20, 4.5),            # BASELINE config C2 (n=8448, BP-20, 64-QAM)
    (200, 400, 2, None, 16, 6, 2.0),               # BG2, partial last row of the pruned graph
])
def test_c_oracle_equals_numpy_oracle(k, n, m, bg, batch, iters, ebno):
    code = LDPC5GCode(k, n, m, bg)
    u, llr = _noisy_llr(code, m, batch, ebno, k)
    for cn in ("minsum", "offset-minsum", "boxplus-phi", "boxplus"):
        dec = bp.LDPC5GDecoder(code, cn_update=cn, num_iter=iters, hard_out=False)
        llr_full = dec.rate_recover(llr)
        ref = dec.decode(llr_full)                                        # NumPy, literal restatement, [B, N_vn]
        got = cbind.bp_decode(dec, llr_full)                              # C, OpenMP
        assert got.shape == ref.shape
        if cn in ("minsum", "offset-minsum", "boxplus-phi"):
            # same defined order - and, for boxplus-phi since round 3, the same DEFINED float32 exp / log (Cephes /
            # Eigen restatement, oracle/ldpc_bp.c) in both oracles: bit for bit
            assert np.array_equal(got, ref), cn
        else:
            # transcendental rules: glibc vs NumPy SIMD libm differ in the last bits and phi amplifies that on
            # saturating messages (DESIGN.md "phi conditioning"): after ONE iteration the two agree to 1e-4, after all
            # iterations the decisions agree on every decoded word and the soft values on most positions
            one_c, one_py = cbind.bp_decode(dec, llr_full, num_iter=1), dec.decode(llr_full, num_iter=1)
            assert np.isclose(one_c, one_py, rtol=1e-4, atol=1e-3).mean() > 0.995, cn
            conv = np.all((ref[:, :k] > 0) == u, axis=1)
            assert conv.any() and np.array_equal((got[:, :k] > 0)[conv], (ref[:, :k] > 0)[conv]), cn
            assert np.mean(np.abs(got - ref) <= 1e-2 * (1 + np.abs(ref))) > 0.9, cn
        hard = cbind.bp_decode(dec, llr_full, hard_out=True)
        assert np.array_equal(hard, (got > 0).astype(np.float32))
    # zero iterations = hard decision on the channel LLRs, one iteration on a generic PCM as well
    dec0 = bp.LDPC5GDecoder(code, cn_update="minsum", num_iter=0, hard_out=False)
    assert np.array_equal(cbind.bp_decode(dec0, dec0.rate_recover(llr)), dec0.decode(dec0.rate_recover(llr)))


@pytest.mark.parametrize("m", [2, 4, 6])
def test_demapper_oracle_against_reference_test_formula(m):
    """The loop of test_mapping.py:175-199 (per-symbol noise variance in [0.01, 100]) on the oracle's demapper."""
    rng = np.random.default_rng(m)
    pts = omap.qam(m)
    c0, c1 = omap._bit_sets(m)
    bits = rng.integers(0, 2, (20, 10 * m)).astype(np.float32)
    x = omap.mapper(bits, pts)
    x = (x + 0.2 * (rng.normal(size=x.shape) + 1j * rng.normal(size=x.shape))).astype(np.complex64)
    no = rng.uniform(0.01, 100, x.shape).astype(np.float32)
    app, maxlog = omap.demapper(x, no, pts, "app"), omap.demapper(x, no, pts, "maxlog")
    for l in range(x.shape[0]):
        for i, y in enumerate(x[l]):
            e = -np.abs(y - pts) ** 2 / no[l, i]
            want_app = logsumexp(np.take(e, c1), axis=0) - logsumexp(np.take(e, c0), axis=0)
            want_max = np.max(np.take(e, c1), axis=0) - np.max(np.take(e, c0), axis=0)
            assert np.allclose(want_app, app[l, i * m:(i + 1) * m], atol=1e-5)
            assert np.allclose(want_max, maxlog[l, i * m:(i + 1) * m], atol=1e-5)


def test_ebnodb2no_with_resource_grid():
    import sionna_amd.phy as phy
    # (the Kronecker pilots are drawn on the device, so the CPU test uses a pilot-free grid; the C4 GPU chain test
    # asserts the same equality for the config-4 grid with pilots)
    kw = dict(num_tx=1, num_streams_per_tx=2, cyclic_prefix_length=6, num_guard_carriers=[5, 6], dc_null=True)
    rg, org = phy.ofdm.ResourceGrid(14, 76, 15e3, **kw), o.ResourceGrid(14, 76, 15e3, **kw)
    for db, m, r in ((-3.0, 2, 0.5), (0.0, 4, 1 / 3), (7.5, 6, 0.75)):
        got, ref = float(phy.utils.ebnodb2no(db, m, r, rg)), float(outil.ebnodb2no(db, m, r, org))
        # utils/misc.py:225-251 by hand: Es = 1/num_streams_per_tx, scaled by (all REs incl. CP) / (data REs)
        es = (1 / 2) * (14 * (1 + 6 / 76) * 64) / rg.num_data_symbols
        want = 1 / (10 ** (db / 10) * r * m / es)
        assert got == ref and abs(got - want) < 2e-6 * want
    assert rg.num_data_symbols == org.num_data_symbols == 14 * 64 and rg.num_effective_subcarriers == 64


def test_phi_spec_exp_log_accuracy_and_endpoints():
    """The defined float32 exp / log of the boxplus-phi rule (oracle/ldpc_bp.c): within ~1 ulp of float64 on a dense
    sample of the clipped domain (the full sweep of all 2.3e8 floats - exp 1.005 ulp, log 0.90 ulp - is quoted in the
    source), the endpoints the reference's own test needs come out exactly ("all-erasure -> zeros",
    test_ldpc_decoding.py:279-291), and phi follows -log(tanh(x/2)) where that is well conditioned."""
    rng = np.random.default_rng(0)
    x = np.concatenate([np.exp(rng.uniform(np.log(8.5e-8), np.log(16.635532), 400000)), np.linspace(8.5e-8, 16.635532, 100001)]).astype(np.float32)
    ulp = lambda v: np.spacing(np.abs(v).astype(np.float32)).astype(np.float64)
    e = cbind.spec_exp_f32(x)
    t = np.exp(x.astype(np.float64))
    assert np.max(np.abs(e - t) / ulp(t)) < 1.1
    for a in (e + np.float32(1), e - np.float32(1)):
        a = a[a > 0]
        l, tl = cbind.spec_log_f32(a), np.log(a.astype(np.float64))
        assert np.max(np.abs(l - tl) / ulp(tl)) < 1.0
    ends = cbind.phi_f32(np.array([16.635532, 20.0, 8.5e-8, 0.0, 16.6], np.float32))
    assert ends[0] == 0 and ends[1] == 0 and ends[2] == np.float32(16.635532) and ends[3] == np.float32(16.635532) and ends[4] == 0
    xs = np.linspace(0.05, 6, 2000).astype(np.float32)
    assert np.allclose(cbind.phi_f32(xs), -np.log(np.tanh(xs.astype(np.float64) / 2)), rtol=2e-5, atol=1e-6)


def test_phi_round5_exp_log_accuracy():
    """The cheaper defined exp / log that phi evaluates since round 5 (oracle/ldpc_bp.c phi_expf / phi_logf = csrc/bp_math.h
    phi_exp2_f32 / phi_log2_f32): exp within 1.1 ulp like the Cephes form (same reduction and polynomial); the table-driven
    log within 1.2e-7 ABSOLUTE + 0.6 ulp (its error does not shrink with the result near ln 1 = 0 - phi only ever takes
    differences of logarithms whose larger one is >= ln 2); and phi itself agrees with the Cephes-based phi of rounds 3-4
    to 2e-6 relative over the clipped domain."""
    rng = np.random.default_rng(1)
    x = np.concatenate([np.exp(rng.uniform(np.log(8.5e-8), np.log(16.635532), 400000)), np.linspace(8.5e-8, 16.635532, 100001)]).astype(np.float32)
    ulp = lambda v: np.spacing(np.abs(v).astype(np.float32)).astype(np.float64)
    e = cbind.phi_exp_f32(x)
    t = np.exp(x.astype(np.float64))
    assert np.max(np.abs(e - t) / ulp(t)) < 1.1
    for a in (e + np.float32(1), e - np.float32(1)):
        a = a[a > 0]
        l, tl = cbind.phi_log_f32(a), np.log(a.astype(np.float64))
        assert np.max(np.abs(l - tl) - 0.6 * ulp(tl)) < 1.2e-7
    # against the literal form on float64: where phi is well conditioned
    xs = np.linspace(0.05, 6, 4000).astype(np.float32)
    assert np.allclose(cbind.phi_f32(xs), -np.log(np.tanh(xs.astype(np.float64) / 2)), rtol=2e-5, atol=1e-6)
    # against the Cephes-based definition of rounds 3-4 (spec_exp / spec_log, still the Polar BP decoder's arithmetic)
    xc = np.clip(x, np.float32(8.5e-8), np.float32(16.0))
    ec = cbind.spec_exp_f32(xc)
    old = cbind.spec_log_f32(ec + np.float32(1)) - cbind.spec_log_f32(ec - np.float32(1))
    new = cbind.phi_f32(xc)
    assert np.mean(new == old) > 0.85 and np.mean(np.abs(new - old) <= 1e-5 * np.abs(old) + 1e-7) > 0.99


def test_phi_defined_range_over_its_whole_domain():
    """Two facts the generated boxplus-phi kernel (csrc/jit/ldpc5g_jit_templates.h, jit_cn_phi_rolled) builds on, checked on EVERY
    float of the clamped domain [8.5e-8, 16.635532] (231.6 M arguments, ~8 s of the C oracle): the defined phi is never negative
    (nor -0) - its sign bit is free to carry the edge's sign between the two passes - and never exceeds phi(8.5e-8) = 16.635532,
    so min(phi, llr_max) can only act for llr_max below that value."""
    lo, hi = int(np.float32(8.5e-8).view(np.uint32)), int(np.float32(16.635532).view(np.uint32))
    top = np.float32(16.635532)
    step = 1 << 24
    for s0 in range(lo, hi + 1, step):
        u = np.arange(s0, min(s0 + step, hi + 1), dtype=np.uint32)
        y = cbind.phi_f32(u.view(np.float32))
        assert not np.any(y.view(np.uint32) >> 31), "a negative (or -0) phi value"
        assert y.max() <= top
    assert cbind.phi_f32(np.array([8.5e-8], np.float32))[0] == top


@pytest.mark.parametrize("k,n,m,bg,batch,iters,ebno", [(1024, 2048, 2, "bg1", 19, 10, 1.5), (2816, 8448, 6, "bg1", 9, 20, 4.5)])
def test_simd_c_baseline_equals_scalar_c_oracle(k, n, m, bg, batch, iters, ebno):
    """bench.py's CPU baseline (8 codewords per vector, oracle_ldpc_bp_decode_simd) returns the scalar C oracle's soft
    outputs bit for bit for the three rules it covers, batches that are not multiples of 8 included."""
    code = LDPC5GCode(k, n, m, bg)
    u, llr = _noisy_llr(code, m, batch, ebno, k + 1)
    for cn in ("minsum", "offset-minsum", "boxplus-phi"):
        dec = bp.LDPC5GDecoder(code, cn_update=cn, num_iter=iters, hard_out=False)
        llr_full = dec.rate_recover(llr)
        assert np.array_equal(cbind.bp_decode(dec, llr_full, simd=True), cbind.bp_decode(dec, llr_full)), cn
        assert np.array_equal(cbind.bp_decode(dec, llr_full, simd=True, hard_out=True), cbind.bp_decode(dec, llr_full, hard_out=True))


def test_f64_ofdm_oracle_is_the_float32_oracle_in_double():
    """oracle/f64_ofdm.py (specification of precision="double" for the OFDM link blocks) draws the same Philox realisation as
    the float32 oracle - which is pinned to the executed reference - and evaluates the same formulas: both agree to float32
    rounding on the same inputs."""
    from oracle import f64_ofdm as o64, ofdm as o32, utils as outil
    w64, w32 = o64.complex_normal(3, 1, 999, 1.7), outil.complex_normal(3, 1, 999, 1.7)
    assert w64.dtype == np.complex128 and np.allclose(w64, w32, rtol=0, atol=2e-6)
    a64, t64 = o64.tdl_cir(7, 0, 4, 14, 14e3, [0, 3e-8, 2e-7], [0.5, 0.3, 0.2], 10.0, 200.0, 2, 2, 20, los_power=0.4)
    a32, t32 = o32.tdl_cir(7, 0, 4, 14, 14e3, [0, 3e-8, 2e-7], [0.5, 0.3, 0.2], 10.0, 200.0, 2, 2, 20, los_power=0.4)
    assert np.allclose(a64, a32, rtol=1e-3, atol=2e-4) and np.allclose(t64, t32)
    fr = o64.subcarrier_frequencies(76, 15e3)
    assert np.array_equal(fr.astype(np.float32), o32.subcarrier_frequencies(76, 15e3))
    for norm in (False, True):
        h64 = o64.cir_to_ofdm_channel(fr, a64, t64, norm)
        assert np.allclose(h64, o32.cir_to_ofdm_channel(fr.astype(np.float32), a32, t32, norm), rtol=1e-3, atol=1e-3)
    rng = np.random.default_rng(0)
    x = (rng.normal(size=(4, 1, 2, 14, 76)) + 1j * rng.normal(size=(4, 1, 2, 14, 76)))
    assert np.allclose(o64.apply_ofdm_channel(x, h64), o32.apply_ofdm_channel(x.astype(np.complex64), h64.astype(np.complex64)), rtol=1e-4, atol=1e-4)
    xt = o64.ofdm_modulate(x, 6)
    assert np.allclose(xt, o32.ofdm_modulate(x.astype(np.complex64), 6), rtol=1e-5, atol=1e-5)
    assert np.allclose(o64.ofdm_demodulate(xt, 76, -2, 6), o32.ofdm_demodulate(xt.astype(np.complex64), 76, -2, 6), rtol=1e-4, atol=1e-4)
    assert np.allclose(o64.ofdm_demodulate(xt, 76, 0, 6), x, rtol=1e-12, atol=1e-12)
    rg = o32.ResourceGrid(14, 76, 15e3, num_tx=1, num_streams_per_tx=2, cyclic_prefix_length=6, num_guard_carriers=[5, 6],
                          dc_null=True, pilot_pattern="kronecker", pilot_ofdm_symbol_indices=[2, 11])
    d = (rng.normal(size=(3, 1, 2, rg.num_data_symbols)) + 1j * rng.normal(size=(3, 1, 2, rg.num_data_symbols)))
    assert np.allclose(o64.rg_map(rg, d), o32.rg_map(rg, d.astype(np.complex64)), rtol=1e-6, atol=1e-6)
    y = (rng.normal(size=(3, 1, 4, 14, 76)) + 1j * rng.normal(size=(3, 1, 4, 14, 76)))
    h64, e64 = o64.ls_estimate(rg, y, 0.2)
    h32, e32 = o32.ls_estimate(rg, y.astype(np.complex64), 0.2)
    assert np.allclose(h64, h32, rtol=1e-5, atol=1e-5) and np.allclose(e64, e32, rtol=1e-5, atol=1e-6)
