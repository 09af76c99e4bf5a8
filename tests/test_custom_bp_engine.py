"""Host-logic tests (CPU tensors) of the extension-point engine sionna_amd/phy/fec/ldpc/custom.py: the torch
restatement of _bp_iter with RaggedMessages, the exported node updates and the callbacks, against oracle/ldpc_bp.py.
(The decoder blocks only ever hand it device tensors; the GPU tests run it through LDPCBPDecoder / LDPC5GDecoder.)"""
import numpy as np
import pytest
import torch

from oracle import ldpc_bp as obp
from sionna_amd.phy.fec.ldpc import custom


def _pcm(m, n, seed):
    rng = np.random.default_rng(seed)
    pcm = (rng.random((m, n)) < 0.25).astype(np.int64)
    for c in range(n):
        if pcm[:, c].sum() == 0:
            pcm[rng.integers(0, m), c] = 1
    for r in range(m):
        while pcm[r].sum() < 2:
            pcm[r, rng.integers(0, n)] = 1
    return pcm


def _engine(dec, cn, vn="sum", c2v=(), v2c=(), sched=None):
    return custom.CustomBPEngine(dec.cn_idx, dec.vn_idx, dec.num_cns, dec.num_vns, custom.BUILTIN_CN[cn],
                                 custom.BUILTIN_VN[vn], c2v, v2c, sched)


@pytest.mark.parametrize("cn", ["minsum", "offset-minsum", "boxplus-phi", "boxplus"])
@pytest.mark.parametrize("sched", [None, "rows"])
def test_engine_matches_oracle(cn, sched):
    pcm = _pcm(12, 30, 1)
    rng = np.random.default_rng(2)
    llr = (rng.normal(size=(7, 30)) * 3).astype(np.float32)
    schedule = None if sched is None else np.arange(12).reshape(4, 3)
    for it in (0, 1, 4):
        ref = obp.LDPCBPDecoder(pcm, cn_update=cn, hard_out=False, num_iter=it, return_state=True, cn_schedule="flooding" if schedule is None else schedule)
        xr, sr = ref.decode(llr)
        x, v2c = _engine(ref, cn, sched=schedule).decode(torch.from_numpy(llr), it, 20.0)
        tol = dict(rtol=1e-5, atol=1e-5) if cn in ("minsum", "offset-minsum") else dict(rtol=2e-4, atol=2e-4)
        assert np.allclose((-x).numpy(), xr, **tol), (cn, it)
        assert np.allclose((-v2c).numpy(), sr, **tol), (cn, it)
    # state in: 2 + 2 iterations == 4 iterations
    eng = _engine(ref, cn, sched=schedule)
    x2, s2 = eng.decode(torch.from_numpy(llr), 2, 20.0)
    x4a, _ = eng.decode(torch.from_numpy(llr), 2, 20.0, msg_v2c=-s2)
    x4, _ = eng.decode(torch.from_numpy(llr), 4, 20.0)
    if sched is None:            # with a schedule the c2v buffer restarts from zero, like in the reference (:581)
        assert torch.allclose(x4a, x4, rtol=1e-5, atol=1e-5)


def test_callbacks_and_weighted_bp_gradient():
    pcm = _pcm(10, 24, 3)
    ref = obp.LDPCBPDecoder(pcm, cn_update="boxplus", hard_out=False, num_iter=5)
    rng = np.random.default_rng(4)
    llr = torch.from_numpy((-(4.0 + rng.normal(size=(64, 24)) * 2.5)).astype(np.float32))   # all-zero codeword, logits < 0
    seen = []
    stats, exit_cb = custom.DecoderStatisticsCallback(5), custom.EXITCallback(5)
    eng = _engine(ref, "boxplus", c2v=[stats, lambda m, it: (seen.append(("c2v", it, m.shape)), m)[1]],
                  v2c=[exit_cb, lambda m, it, x_hat: (seen.append(("v2c", it, tuple(x_hat.shape))), m)[1]])
    x, _ = eng.decode(llr, 5, 20.0)
    assert [s[1] for s in seen if s[0] == "c2v"] == [0, 1, 2, 3, 4] and [s[1] for s in seen if s[0] == "v2c"] == [0, 1, 2, 3, 4, 5]
    # decoding.py:583-594: the v2c callbacks run once before the first iteration (it = 0, third argument = channel LLRs)
    assert seen[0] == ("v2c", 0, (24, 64)) and seen[1][2] == (10, None, 64) and seen[2][2] == (24, 64)
    assert np.all(stats.num_samples == 64) and np.all(np.diff(stats.success_rate) >= 0) and stats.success_rate[-1] > 0.5
    assert 0 < stats.avg_number_iterations <= 5
    mi = exit_cb.mi
    assert np.all(np.isfinite(mi)) and mi[-1] > mi[1] > 0 and mi[0] > 0          # mi[0]: the channel messages
    # callbacks that return the messages unchanged do not change the result
    x0, _ = _engine(ref, "boxplus").decode(llr, 5, 20.0)
    assert torch.equal(x, x0)
    # weighted BP: unit weights are the identity, the loss is differentiable w.r.t. the weights
    class _W(custom.WeightedBPCallback):
        def __init__(self, num_edges):
            self._edge_weights = torch.nn.Parameter(torch.ones(num_edges))
    w = _W(ref.num_edges)
    xw, _ = _engine(ref, "boxplus", c2v=[w], v2c=[w]).decode(llr, 5, 20.0)
    assert torch.allclose(xw, x0, rtol=1e-6, atol=1e-6)
    loss = torch.nn.functional.softplus(-xw).mean()
    loss.backward()
    assert w.weights.grad is not None and torch.isfinite(w.weights.grad).all() and w.weights.grad.abs().sum() > 0


def test_custom_node_functions_and_identity():
    pcm = _pcm(8, 20, 5)
    ref = obp.LDPCBPDecoder(pcm, cn_update="identity", vn_update="identity", hard_out=False, num_iter=3)
    llr = (np.random.default_rng(6).normal(size=(5, 20)) * 2).astype(np.float32)
    x, _ = _engine(ref, "identity", "identity").decode(torch.from_numpy(llr), 3, 20.0)
    assert np.allclose((-x).numpy(), ref.decode(llr), rtol=1e-5, atol=1e-5)
    # a user-written check-node rule on RaggedMessages: scaled min-sum
    def scaled_minsum(msg, llr_clipping=None):
        return custom.cn_update_minsum(msg, llr_clipping) * 0.8
    ms = obp.LDPCBPDecoder(pcm, cn_update="minsum", hard_out=False, num_iter=1)
    eng = custom.CustomBPEngine(ms.cn_idx, ms.vn_idx, ms.num_cns, ms.num_vns, scaled_minsum, custom.vn_update_sum, [], [])
    x1, _ = eng.decode(torch.from_numpy(llr), 1, 20.0)
    assert x1.shape == (5, 20) and torch.isfinite(x1).all()
    r = custom.RaggedMessages(torch.arange(6.0).reshape(6, 1), torch.tensor([0, 2, 6]))
    assert r.shape == (2, None, 1) and r.reduce_sum().reshape(-1).tolist() == [1.0, 14.0]
    assert r.reduce_min().reshape(-1).tolist() == [0.0, 2.0] and (2 * r).flat_values[5, 0] == 10.0
