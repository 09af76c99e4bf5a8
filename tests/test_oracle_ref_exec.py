"""Pins oracle/ldpc_bp.py, oracle/ldpc_bp.c and oracle/ldpc5g.py to outputs of the reference's OWN code.

tests/golden/ldpc_bp_ref_golden.npz was produced by tools/gen_ldpc_bp_golden.py, which imports the reference's
``fec/ldpc/decoding.py`` and ``encoding.py`` unmodified under a NumPy stand-in for TensorFlow (tools/ref_exec) and runs
``vn_update_sum``, ``cn_update_*``, ``LDPCBPDecoder._bp_iter`` loops and the ``LDPC5GEncoder -> LDPC5GDecoder`` chain.
Everything that involves only IEEE-exact float32 operations (VN update, min-sum family, whole min-sum decodes, hard
decisions, decoder state) must match BIT FOR BIT; exp/log/tanh-based rules are bit-identical once the oracle uses the
same (NumPy) transcendental and within 1e-5 per node update with its own defined exp/log.
The GPU == oracle tests (tests/test_gpu_parity.py) carry these pins to the HIP kernels."""
import hashlib
import os

import numpy as np
import pytest

from oracle import cbind, ldpc_bp as obp
from oracle.ldpc5g import LDPC5GCode

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ldpc_bp_ref_golden.npz")
# tools/gen_ldpc_bp_golden.py --baseline: the BASELINE.json codes themselves (C2: BG1 k=2816 n=8448 with the 64-QAM
# interleaver, 20 iterations; C4's code k=768 n=1536) through the executed reference
GOLD_BASELINE = os.path.join(os.path.dirname(__file__), "golden", "ldpc_bp_ref_golden_baseline.npz")
EX = os.path.join(os.path.dirname(__file__), "golden", "example_pcms.npz")
RULES = ("minsum", "offset-minsum", "boxplus", "boxplus-phi")


@pytest.fixture(scope="module")
def g():
    merged = dict(np.load(GOLD))
    merged.update(np.load(GOLD_BASELINE))
    return merged


def example_pcm(i):
    ex = np.load(EX)
    pcm = np.zeros(tuple(ex[f"shape_{i}"]))
    rc = ex[f"rc_{i}"]
    pcm[rc[0], rc[1]] = 1
    return pcm


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def numpy_phi(x):
    """decoding.py:1092-1120 with NumPy's float32 exp/log - the arithmetic the fixture generator ran."""
    x = np.clip(x, np.float32(8.5e-8), np.float32(16.635532))
    return np.log(np.exp(x) + np.float32(1)) - np.log(np.exp(x) - np.float32(1))


@pytest.fixture()
def phi_as_numpy(monkeypatch):
    monkeypatch.setattr(obp, "_phi", numpy_phi)


@pytest.mark.parametrize("i", range(5))
def test_edge_order_is_the_reference_stable_order(g, i):
    d = obp.LDPCBPDecoder(example_pcm(i), "minsum")
    assert np.array_equal(np.stack([d.cn_idx, d.vn_idx]), g[f"node_ex{i}_edges"])


@pytest.mark.parametrize("i", range(5))
def test_vn_update_and_minsum_family_bit_exact(g, i):
    d = obp.LDPCBPDecoder(example_pcm(i), "minsum")
    p = f"node_ex{i}_"
    clip = np.float32(20.)
    assert np.array_equal(obp.cn_update_minsum(d._cn_rag, g[p + "cn_in"], clip), g[p + "minsum"])
    assert np.array_equal(obp.cn_update_offset_minsum(d._cn_rag, g[p + "cn_in"], clip), g[p + "offset"])
    assert np.array_equal(obp.cn_update_offset_minsum(d._cn_rag, g[p + "cn_in"], None, offset=0.3), g[p + "offset03_noclip"])
    xe, xtot = obp.vn_update_sum(d._vn_rag, g[p + "vn_c2v"], g[p + "vn_llr"], clip)
    assert np.array_equal(xe, g[p + "vn_xe"]) and np.array_equal(xtot, g[p + "vn_xtot"])
    xe, xtot = obp.vn_update_sum(d._vn_rag, g[p + "vn_c2v"], g[p + "vn_llr"], None)
    assert np.array_equal(xe, g[p + "vn_xe_noclip"]) and np.array_equal(xtot, g[p + "vn_xtot_noclip"])


@pytest.mark.parametrize("i", range(5))
def test_tanh_and_phi_structure_bit_exact_under_numpy_transcendentals(g, i, phi_as_numpy):
    d = obp.LDPCBPDecoder(example_pcm(i), "boxplus")
    p = f"node_ex{i}_"
    clip = np.float32(20.)
    assert np.array_equal(obp.cn_update_tanh(d._cn_rag, g[p + "cn_in"], clip), g[p + "tanh"])
    assert np.array_equal(obp.cn_update_phi(d._cn_rag, g[p + "cn_in"], clip), g[p + "phi"])


@pytest.mark.parametrize("i", range(5))
def test_phi_with_the_defined_exp_log_within_1e5(g, i):
    """One node update with the oracle's own exp/log (oracle/ldpc_bp.c:31-88) against NumPy's float32 exp/log.  Both are
    <= 1 ulp routines; phi(sum phi - phi_self) amplifies a last-bit difference wherever the subtraction cancels (one
    dominant edge) or the argument sits at the 8.5e-8 clip, which the deliberately nasty inputs (exact zeros, 1e-6,
    +-llr_max) provoke.  Hence: >= 97 % of the messages within the north star's 1e-5 relative bar, all within 5e-4
    absolute (llr_max is 20)."""
    d = obp.LDPCBPDecoder(example_pcm(i), "boxplus-phi")
    p = f"node_ex{i}_"
    got = obp.cn_update_phi(d._cn_rag, g[p + "cn_in"], np.float32(20.))
    ref = g[p + "phi"]
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)
    assert np.mean(rel <= 1e-5) >= 0.97, np.mean(rel <= 1e-5)
    assert np.abs(got - ref).max() <= 5e-4, np.abs(got - ref).max()
    big = np.abs(ref) > 1e-3
    assert np.array_equal(np.sign(got[big]), np.sign(ref[big]))


def _decoder(pcm, rule, it, **kw):
    return obp.LDPCBPDecoder(pcm, rule, num_iter=it, hard_out=kw.pop("hard_out", False), return_state=True, **kw)


@pytest.mark.parametrize("i", range(5))
def test_bp_loops_on_example_pcms(g, i, phi_as_numpy):
    pcm, llr = example_pcm(i), g[f"bp_ex{i}_llr"]
    for rule in RULES:
        for it in ((1, 5) if i != 4 else (5,)):
            x, st = _decoder(pcm, rule, it).decode(llr)
            assert np.array_equal(x, g[f"bp_ex{i}_{rule}_it{it}_x"]), (rule, it)
            assert np.array_equal(sha(st), g[f"bp_ex{i}_{rule}_it{it}_state_sha"]), (rule, it)
    d = _decoder(pcm, "minsum", 3, hard_out=True)
    x1, st1 = d.decode(llr)
    x2, st2 = d.decode(llr, msg_v2c=st1)
    assert np.array_equal(x1.astype(np.uint8), g[f"bp_ex{i}_minsum_hard3"])
    assert np.array_equal(x2.astype(np.uint8), g[f"bp_ex{i}_minsum_hard3_resumed"])
    want = g[f"bp_ex{i}_minsum_resumed_state"]
    assert np.array_equal(st2 if i < 2 else sha(st2), want)


@pytest.mark.parametrize("i", range(5))
def test_c_oracle_minsum_matches_reference_execution(g, i):
    """The C restatement (what the GPU tests compare with) against the reference-executed loops, min-sum family."""
    pcm, llr = example_pcm(i), g[f"bp_ex{i}_llr"]
    for rule in ("minsum", "offset-minsum"):
        for it in ((1, 5) if i != 4 else (5,)):
            d = obp.LDPCBPDecoder(pcm, rule, num_iter=it, hard_out=False)
            assert np.array_equal(cbind.bp_decode(d, llr), g[f"bp_ex{i}_{rule}_it{it}_x"]), (rule, it)


CASES_5G = ("c1", "bg2s", "bg2m", "bg1r", "c2", "c4")


def _code(g, tag):
    k, n, bg, z, m, iters = (int(v) for v in g[f"g5_{tag}_meta"])
    code = LDPC5GCode(k, n, m or None, f"bg{bg}")
    assert code.z == z
    return code, iters


@pytest.mark.parametrize("tag", CASES_5G)
def test_5g_encoder_matches_reference_execution(g, tag):
    code, _ = _code(g, tag)
    assert np.array_equal(code.encode(g[f"g5_{tag}_u"]), g[f"g5_{tag}_c"])


@pytest.mark.parametrize("tag", CASES_5G)
def test_5g_decoder_matches_reference_execution(g, tag, phi_as_numpy):
    code, iters = _code(g, tag)
    llr = g[f"g5_{tag}_llr"]
    for rule in RULES:
        d = obp.LDPC5GDecoder(code, rule, hard_out=False, return_infobits=False, num_iter=iters, return_state=True)
        x, st = d.decode5g(llr)
        assert np.array_equal(x, g[f"g5_{tag}_{rule}_x"]), rule
        assert np.array_equal(sha(st), g[f"g5_{tag}_{rule}_state_sha"]), rule
        d = obp.LDPC5GDecoder(code, rule, hard_out=True, return_infobits=True, num_iter=iters)
        assert np.array_equal(d.decode5g(llr).astype(np.uint8), g[f"g5_{tag}_{rule}_uhat"]), rule
    if tag in ("c1", "bg2s", "c2"):
        it = max(2, iters // 2)
        d = obp.LDPC5GDecoder(code, "minsum", hard_out=False, return_infobits=False, num_iter=it, cn_schedule="layered")
        assert np.array_equal(d.decode5g(llr), g[f"g5_{tag}_layered_minsum_x"])
        d = obp.LDPC5GDecoder(code, "boxplus-phi", hard_out=False, return_infobits=False, num_iter=it, cn_schedule="layered")
        assert np.array_equal(d.decode5g(llr), g[f"g5_{tag}_layered_phi_x"])


@pytest.mark.parametrize("tag", CASES_5G)
def test_5g_c_oracle_minsum_and_defined_phi(g, tag):
    """C oracle on the pruned 5G graph: min-sum family bit-exact to the reference execution; boxplus-phi with the defined
    exp/log: identical hard decisions on converged words and soft outputs within 1e-5 for >= 99 % of them (BP amplifies
    last-bit differences on words that do not converge, which is why this is not a max-norm bar)."""
    code, iters = _code(g, tag)
    llr = g[f"g5_{tag}_llr"]
    for rule in ("minsum", "offset-minsum"):
        d = obp.LDPC5GDecoder(code, rule, hard_out=False, return_infobits=False, num_iter=iters)
        full = cbind.bp_decode(d, d.rate_recover(llr))
        x_nf = np.concatenate([full[:, :code.k], full[:, code.k_ldpc:]], axis=1)[:, 2 * code.z:2 * code.z + code.n]
        if code.num_bits_per_symbol is not None:
            x_nf = x_nf[:, code.out_int]
        assert np.array_equal(x_nf, g[f"g5_{tag}_{rule}_x"]), rule
    d = obp.LDPC5GDecoder(code, "boxplus-phi", hard_out=False, return_infobits=True, num_iter=iters)
    got = cbind.bp_decode(d, d.rate_recover(llr))[:, :code.k]
    u = g[f"g5_{tag}_u"]
    uhat_ref = g[f"g5_{tag}_boxplus-phi_uhat"]
    conv = np.all(uhat_ref == u, axis=1)                           # words the reference decoded correctly
    assert conv.sum() >= 2
    assert np.array_equal((got[conv] > 0).astype(np.uint8), u[conv])
