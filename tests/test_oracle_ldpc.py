"""Pin the CPU oracle (oracle/ldpc5g.py, oracle/ldpc_bp.py) against the reference's anchors.

Anchors, all usable without TensorFlow:
* 28 golden generator matrices  -> tests/golden/ldpc_enc_golden.npz (tools/gen_golden.py)
* per-node leave-one-out formulas of test/unit/fec/test_ldpc_decoding.py:400-655
  (re-derived here in float64, tolerance rtol=atol=1e-3 like the reference)
* decoder invariants of the same file (:55-91, :279-304, :363-380, :1024-1040)
"""
import os
import numpy as np
import pytest

from oracle.ldpc5g import LDPC5GCode, generate_out_int
from oracle import ldpc_bp as bp
from oracle import utils as outil

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ldpc_enc_golden.npz"))


@pytest.mark.parametrize("k,n", [tuple(int(v) for v in p) for p in G["params"]])
def test_encoder_golden(k, n):
    code = LDPC5GCode(k, n)
    u = np.unpackbits(G[f"u_{k}_{n}"], axis=1)[:, :k].astype(np.float32)
    c_ref = np.unpackbits(G[f"c_{k}_{n}"], axis=1)[:, :n].astype(np.float32)
    assert np.array_equal(code.encode(u), c_ref)


def test_encoder_structure():
    # test_ldpc_encoding.py:31-70: all-zero -> all-zero; systematic part u[2Z:] == c[:k-2Z]
    code = LDPC5GCode(1024, 2048, bg="bg1")
    assert (code.z, code.n_ldpc, code.k_ldpc) == (48, 3264, 1056)
    assert np.all(code.encode(np.zeros((2, 1024), np.float32)) == 0)
    u = np.random.default_rng(0).integers(0, 2, (4, 1024)).astype(np.float32)
    c = code.encode(u)
    assert np.array_equal(u[:, 2 * code.z:], c[:, :1024 - 2 * code.z])
    # H c^T = 0 for the un-rate-matched codeword
    s = np.concatenate([u, np.zeros((4, code.k_ldpc - 1024), np.float32)], 1)
    cw = code.encode_full(s)
    assert np.all((code.pcm @ cw.T) % 2 == 0)


def test_survey_table_parameters():
    # derived code parameters
    c2 = LDPC5GCode(2816, 8448, num_bits_per_symbol=6, bg="bg1")
    d2 = bp.LDPC5GDecoder(c2)
    assert (c2.z, c2.i_ls, c2.n_ldpc, c2.k_ldpc) == (128, 0, 8704, 2816)
    assert (d2.num_vns, d2.num_cns, d2.num_edges, d2.nb_pruned) == (8704, 5888, 40448, 0)
    c1 = LDPC5GCode(1024, 2048, bg="bg1")
    d1 = bp.LDPC5GDecoder(c1)
    assert (d1.num_vns, d1.num_cns, d1.num_edges, d1.nb_pruned) == (2176, 1120, 9920, 1088)


def test_out_interleaver():
    for m in (1, 2, 4, 6, 8):
        p, pi = generate_out_int(48 * m, m)
        assert np.array_equal(p[pi], np.arange(48 * m))


# ------------------------------------------------------------------ node updates
DEGS = [3, 4, 5, 6, 7]


def _ragged(rng, degs=DEGS, bs=100):
    ids = np.repeat(np.arange(len(degs)), degs)
    return bp._Ragged(ids, len(degs)), rng.normal(size=(len(ids), bs)).astype(np.float32)


def _loo(r, msg, f):
    """Apply leave-one-out function f(others [d-1,B]) per edge, float64."""
    out = np.zeros(msg.shape, np.float64)
    for n in range(r.num_nodes):
        e = np.nonzero(r.ids == n)[0]
        for i in e:
            others = msg[[j for j in e if j != i]].astype(np.float64)
            out[i] = f(others)
    return out


def _clip(v, c):
    return v if c is None else np.clip(v, -c, c)


@pytest.mark.parametrize("clipv", [5, 20, 100, None])
@pytest.mark.parametrize("offset", [0, 0.5, 1.0])
def test_cn_offset_minsum(clipv, offset):
    r, msg = _ragged(np.random.default_rng(1))
    ref = _loo(r, msg, lambda o: np.prod(np.sign(o), 0) * np.maximum(np.min(np.abs(o), 0) - offset, 0))
    out = bp.cn_update_offset_minsum(r, msg, clipv, offset=offset)
    assert out.dtype == np.float32
    assert np.allclose(out, _clip(ref, clipv), rtol=1e-3, atol=1e-3)


def test_cn_minsum_double_min():
    # test_ldpc_decoding.py:505-511 and the worked example of SURVEY A.4
    r = bp._Ragged([0, 0, 0, 0], 1)
    out = bp.cn_update_minsum(r, np.array([[2.1], [2.1], [3], [4]], np.float32))
    assert np.allclose(out[:, 0], 2.1)
    r = bp._Ragged([0, 0, 0], 1)
    out = bp.cn_update_minsum(r, np.array([[1], [-2], [3]], np.float32))
    assert np.array_equal(out[:, 0], np.array([-2, 1, -1], np.float32))


@pytest.mark.parametrize("clipv", [5, 20, 100, None])
def test_cn_tanh(clipv):
    r, msg = _ragged(np.random.default_rng(2))
    ref = _loo(r, msg, lambda o: 2 * np.arctanh(np.prod(np.tanh(o / 2), 0)))
    assert np.allclose(bp.cn_update_tanh(r, msg, clipv), _clip(ref, clipv), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("clipv", [5, 20, 100, None])
def test_cn_phi(clipv):
    r, msg = _ragged(np.random.default_rng(3))
    phi = lambda x: -np.log(np.tanh(x / 2))
    ref = _loo(r, msg, lambda o: np.prod(np.sign(o), 0) * phi(np.sum(phi(np.abs(o)), 0)))
    assert np.allclose(bp.cn_update_phi(r, msg, clipv), _clip(ref, clipv), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("clipv", [5, 20, 100, None])
@pytest.mark.parametrize("no", [0, 0.1, 1.0])
def test_vn_sum(clipv, no):
    rng = np.random.default_rng(4)
    r, msg = _ragged(rng)
    llr = (no * rng.normal(size=(r.num_nodes, msg.shape[1]))).astype(np.float32)
    xe, xt = bp.vn_update_sum(r, msg, llr, clipv)
    tot = np.stack([msg[r.ids == n].astype(np.float64).sum(0) for n in range(r.num_nodes)]) + llr
    assert np.allclose(xt, _clip(tot, clipv), rtol=1e-3, atol=1e-3)
    assert np.allclose(xe, _clip(tot[r.ids] - msg, clipv), rtol=1e-3, atol=1e-3)


# ------------------------------------------------------------------ decoder invariants
def _example_pcm(i):
    ex = np.load(os.path.join(os.path.dirname(__file__), "golden", "example_pcms.npz"))
    pcm = np.zeros(tuple(ex[f"shape_{i}"]), np.float32)
    pcm[ex[f"rc_{i}"][0], ex[f"rc_{i}"][1]] = 1
    return pcm


CN_TYPES = ["boxplus", "boxplus-phi", "minsum", "offset-minsum"]


@pytest.mark.parametrize("cn", CN_TYPES)
def test_all_erasure_and_bounds(cn):
    pcm = _example_pcm(3)
    dec = bp.LDPCBPDecoder(pcm, cn_update=cn, hard_out=False, num_iter=5, return_state=True)
    x, st = dec.decode(np.zeros((4, pcm.shape[1]), np.float32))
    assert np.all(x == 0) and np.all(st == 0)               # :279-291
    llr = np.random.default_rng(5).normal(scale=30, size=(20, pcm.shape[1])).astype(np.float32)
    x, st = dec.decode(llr)
    assert np.all(np.abs(x) <= 20) and np.all(np.abs(st) <= 20)   # :363-380
    hard = bp.LDPCBPDecoder(pcm, cn_update=cn, hard_out=True, num_iter=5).decode(llr)
    assert set(np.unique(hard)) <= {0.0, 1.0}


def test_identity_routing():
    # :55-91: identity node updates expose the message routing
    pcm = _example_pcm(2)
    dec = bp.LDPCBPDecoder(pcm, cn_update="identity", vn_update="identity", hard_out=False,
                           num_iter=1, llr_max=100000)
    n = pcm.shape[1]
    llr = np.arange(n, dtype=np.float32)[None, :] + 1
    y = -dec.decode(-llr)          # feed internal LLRs = vn index + 1
    deg = pcm.sum(0)
    assert np.allclose(y[0] / (deg + 1), llr[0])


@pytest.mark.parametrize("k,n", [(12, 20), (100, 257), (123, 597), (1234, 1512), (64, 128)])
def test_zero_iter_identity_5g(k, n):
    # :1024-1040: num_iter=0 returns the rate-matched input
    code = LDPC5GCode(k, n)
    dec = bp.LDPC5GDecoder(code, hard_out=False, return_infobits=False, num_iter=0)
    llr = np.random.default_rng(6).normal(size=(3, n)).astype(np.float32)
    assert np.allclose(dec.decode5g(llr), llr)


@pytest.mark.parametrize("cn", CN_TYPES)
@pytest.mark.parametrize("m", [None, 2, 4])
def test_e2e_5g(cn, m):
    # noisy BPSK-like LLRs, decoder recovers the info bits and re-emits the codeword
    code = LDPC5GCode(200, 600, num_bits_per_symbol=m)
    rng = np.random.default_rng(7)
    u = rng.integers(0, 2, (8, 200)).astype(np.float32)
    c = code.encode(u)
    no = 0.4
    y = (2 * c - 1) + np.sqrt(no) * rng.normal(size=c.shape)
    llr = (4 * y / (2 * no)).astype(np.float32)       # logit convention log p1/p0
    assert np.array_equal(bp.LDPC5GDecoder(code, cn_update=cn).decode5g(llr), u)
    c_hat = bp.LDPC5GDecoder(code, cn_update=cn, return_infobits=False).decode5g(llr)
    assert np.array_equal(c_hat, c)


def test_pruning_equivalence():
    # :758-783: pruned vs unpruned graph give (nearly) the same soft output
    code = LDPC5GCode(500, 1000)
    rng = np.random.default_rng(8)
    llr = rng.normal(loc=-1.0, scale=1.5, size=(10, 1000)).astype(np.float32)
    a = bp.LDPC5GDecoder(code, hard_out=False, prune_pcm=True, num_iter=10).decode5g(llr)
    b = bp.LDPC5GDecoder(code, hard_out=False, prune_pcm=False, num_iter=10).decode5g(llr)
    assert np.mean(np.abs(a - b)) < 5e-2


# ------------------------------------------------------------------ CN schedules (layered decoding)
def test_scheduling_independent_checks():
    """Reference test_ldpc_decoding.py:121-160: two disjoint checks -> layered == flooding
    bit for bit; updating only CN 0 differs."""
    pcm = np.array([[1, 1, 1, 0, 0, 0], [0, 0, 0, 1, 1, 1]])
    x = np.arange(6, dtype=np.float32)
    outs = []
    for cns in ("flooding", np.stack([[0], [1]]), np.stack([[0], [0]])):
        dec = bp.LDPCBPDecoder(pcm, num_iter=10, hard_out=False, cn_update="minsum", cn_schedule=cns, llr_max=100000)
        outs.append(dec.decode(x))
    assert np.array_equal(outs[0], outs[1])
    assert not np.array_equal(outs[0], outs[2])
    for bad in (np.zeros(3, np.int32), np.array([[0, 2]]), np.array([[-1, 0]])):
        with pytest.raises(ValueError):
            bp.LDPCBPDecoder(pcm, cn_schedule=bad)
    with pytest.raises(ValueError):
        bp.LDPCBPDecoder(pcm, cn_schedule="layered")


@pytest.mark.parametrize("k,n", [(12, 25), (20, 65), (45, 63), (12, 59), (500, 1000)])
def test_scheduling_pruning_5g(k, n):
    """Reference test_ldpc_decoding.py:735-757: pruning must not disturb the layered schedule."""
    code = LDPC5GCode(k, n)
    x = np.arange(n, dtype=np.float32)[None]
    out = []
    for p in (False, True):
        dec = bp.LDPC5GDecoder(code, cn_schedule="layered", num_iter=5, return_infobits=False, hard_out=False,
                                llr_max=10000, cn_update="minsum", prune_pcm=p)
        out.append(-dec.decode5g(-x))
    assert np.allclose(out[0], out[1])


def test_layered_needs_about_half_the_iterations():
    """Rule of thumb asserted by reference test_ldpc_decoding.py:689-733 (8 layered ~ 16 flooding)."""
    k, n, B = 200, 400, 600
    code = LDPC5GCode(k, n)
    rng = np.random.default_rng(3)
    b = rng.integers(0, 2, (B, k)).astype(np.float32)
    c = code.encode(b)
    no = outil.ebnodb2no(1.5, 2, k / n)
    y = (2 * c - 1) + np.sqrt(no) * rng.normal(size=c.shape).astype(np.float32)
    llr = (2 * y / no).astype(np.float32)
    bler = []
    for cns, it in (("layered", 8), ("flooding", 16)):
        dec = bp.LDPC5GDecoder(code, num_iter=it, cn_update="boxplus", cn_schedule=cns)
        bler.append(np.mean(np.any(dec.decode5g(llr) != b, axis=1)))
    assert 0 < bler[1] < 0.5
    assert np.isclose(bler[0], bler[1], rtol=0.7)
    # and layered beats flooding at the same iteration count
    dec = bp.LDPC5GDecoder(code, num_iter=8, cn_update="boxplus", cn_schedule="flooding")
    assert np.mean(np.any(dec.decode5g(llr) != b, axis=1)) > bler[0]
