"""Integration test on the GPU: iterative detection and decoding (MMSE-PIC detector with bit priors <->
LDPC5GDecoder with soft output and IDD state passing), the use case behind SURVEY.md 8(f) rank 2
(reference notebook Introduction_to_Iterative_Detection_and_Decoding.ipynb).  Statistical property:
IDD iterations lower the BER of one-shot LMMSE detection + decoding at equal decoder effort."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_idd_beats_one_shot_detection():
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    phy.config.seed = 11
    B, K, M, nb, k, n = 192, 4, 4, 4, 600, 1200
    nsym = n // nb
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    mapper = phy.mapping.Mapper("qam", nb)
    src = phy.mapping.BinarySource()
    b = src([B, K, k])
    c = enc(b)                                                    # [B,K,n]
    x = mapper(c).as_subclass(torch.Tensor)                       # [B,K,nsym]
    x = x.permute(0, 2, 1).contiguous()                           # [B,nsym,K]
    h = phy.utils.complex_normal([B, nsym, M, K], 1.0).as_subclass(torch.Tensor)
    no = 0.22
    w = phy.utils.complex_normal([B, nsym, M], no).as_subclass(torch.Tensor)
    y = (h @ x.unsqueeze(-1)).squeeze(-1) + w
    s = (no * torch.eye(M, dtype=torch.complex64, device=y.device)).expand(B, nsym, M, M)

    def to_cw(llr):                                               # [B,nsym,K,nb] -> [B,K,n]
        return llr.as_subclass(torch.Tensor).permute(0, 2, 1, 3).reshape(B, K, n)

    def to_sym(llr):                                              # [B,K,n] -> [B,nsym,K,nb]
        return llr.as_subclass(torch.Tensor).reshape(B, K, nsym, nb).permute(0, 2, 1, 3).contiguous()

    # codeword-bit error rate (the first 2Z systematic bits are punctured, so compare with c, not b)
    ber = lambda llr_cw: float(((llr_cw.as_subclass(torch.Tensor) > 0).float() != c.as_subclass(torch.Tensor)).float().mean())

    # one-shot: LMMSE detector, 12 decoder iterations
    lin = phy.mimo.LinearDetector("lmmse", "bit", "maxlog", constellation_type="qam", num_bits_per_symbol=nb)
    dec12 = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="boxplus-phi", hard_out=False, return_infobits=False, num_iter=12)
    ber_one_shot = ber(dec12(to_cw(lin(y, h, s))))

    # IDD: 3 x (MMSE-PIC with priors + 4 decoder iterations continuing from the decoder state)
    pic = phy.mimo.MMSEPICDetector("bit", "maxlog", num_iter=1, constellation_type="qam", num_bits_per_symbol=nb)
    dec4 = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="boxplus-phi", hard_out=False, return_infobits=False, num_iter=4,
                                      return_state=True)
    prior = torch.zeros((B, nsym, K, nb), dtype=torch.float32, device=y.device)
    state, bers = None, []
    for _ in range(3):
        llr_det = to_cw(pic(y, h, s, prior))                      # extrinsic detector LLRs
        llr_dec, state = dec4(llr_det, msg_v2c=state)             # a-posteriori decoder LLRs, state kept
        bers.append(ber(llr_dec))
        # extrinsic decoder LLRs as new priors (the decoder clips its input and output to llr_max = 20)
        prior = to_sym(llr_dec.as_subclass(torch.Tensor) - torch.clamp(llr_det, -20., 20.))
    assert 1e-4 < ber_one_shot < 0.2, ber_one_shot               # operating point where errors remain
    assert bers[-1] < bers[0], bers                               # iterations help
    assert bers[-1] < 0.5 * ber_one_shot, (bers, ber_one_shot)    # and beat one-shot detection at equal decoder effort


def test_return_state_on_a_large_5g_code():
    """IDD state passing on a code with more than 65535 edges (BG1, Z=384: k=8448, n=25344 -> 121,344 edges, 26,112
    variable nodes): the row-indexed helper kernels of the generic engine walk the rows with a grid stride (grid.y is
    limited to 65535).  min-sum: outputs and the returned v2c state bit for bit against the oracle, and 2 x 3
    iterations with state == 6 iterations."""
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    from oracle.ldpc5g import LDPC5GCode
    from oracle import ldpc_bp as obp
    _ffi.device()
    k, n, B = 8448, 25344, 3
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    code = LDPC5GCode(k, n)
    rng = np.random.default_rng(7)
    u = rng.integers(0, 2, (B, k)).astype(np.float32)
    c = code.encode(u)
    llr = ((2 * c - 1) * 1.2 + rng.normal(size=c.shape)).astype(np.float32) * 2
    mk = lambda it: phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", hard_out=False, num_iter=it, return_state=True)
    x3, st3 = mk(3)(llr)
    ref = obp.LDPC5GDecoder(code, cn_update="minsum", hard_out=False, num_iter=3, return_state=True)
    assert ref.num_edges > 65535 and tuple(st3.shape) == (ref.num_edges, B)
    xr, sr = ref.decode5g(llr)
    assert np.array_equal(x3.cpu().numpy(), xr) and np.array_equal(st3.cpu().numpy(), sr)
    x6a, st6a = mk(3)(llr, msg_v2c=st3)
    x6, st6 = mk(6)(llr)
    assert torch.equal(x6a.as_subclass(torch.Tensor), x6.as_subclass(torch.Tensor))
    assert torch.equal(st6a.as_subclass(torch.Tensor), st6.as_subclass(torch.Tensor))


def test_idd_chain_matches_the_reference_executed_chain():
    """tests/golden/idd_ref_golden.npz = the reference's OWN IddModel chain (ofdm.LinearDetector -> LDPC5GDecoder with state
    -> ofdm.MMSEPICDetector with priors -> LDPC5GDecoder(msg_v2c=state); KBestDetector, EPDetector) executed from its
    source files under a NumPy stand-in for TensorFlow (tools/gen_idd_ref_golden.py).  The HIP path on the same received
    grid: detector LLRs within 1e-5 of their scale, the min-sum decoder's soft output, state and decisions bit for bit."""
    import os
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "idd_ref_golden.npz"))
    n_ue, m = 4, 4
    N = 48 * 12 * m
    rg = phy.ofdm.ResourceGrid(num_ofdm_symbols=14, pilot_ofdm_symbol_indices=[2, 11], fft_size=48, num_tx=n_ue,
                               pilot_pattern="kronecker", subcarrier_spacing=30e3)
    sm = phy.mimo.StreamManagement(np.ones([1, n_ue]), 1)
    h = g["h"]
    hf = np.ascontiguousarray(np.broadcast_to(h[..., None, None], h.shape + (14, 48)))
    y, no = g["y"], g["no"]
    ev = np.zeros(hf.shape, np.float32)
    close = lambda a, b, tol=1e-5: np.abs(a.cpu().numpy().reshape(b.shape) - b).max() <= 4 * tol * np.abs(b).max()
    kw = dict(constellation_type="qam", num_bits_per_symbol=m, hard_out=False)
    llr0 = phy.ofdm.LinearDetector("lmmse", "bit", "maxlog", rg, sm, **kw)(y, hf, ev, no)
    assert close(llr0, g["llr_lmmse"])
    enc = phy.fec.ldpc.LDPC5GEncoder(N // 2, N, num_bits_per_symbol=m)
    D = phy.fec.ldpc.LDPC5GDecoder
    llr_dec, state = D(enc, return_infobits=False, num_iter=12, return_state=True, hard_out=False, cn_update="minsum")(g["llr_lmmse"])
    assert np.array_equal(llr_dec.cpu().numpy(), g["llr_dec"])
    assert tuple(state.shape) == tuple(g["state_shape"]) and np.array_equal(state.cpu().numpy()[:4096], g["state_head"])
    pic = phy.ofdm.MMSEPICDetector(output="bit", demapping_method="maxlog", resource_grid=rg, stream_management=sm, num_iter=1,
                                   constellation_type="qam", num_bits_per_symbol=m, hard_out=False)
    assert close(pic(y, hf, g["llr_dec"], ev, no), g["llr_pic"])
    bh, _ = D(enc, return_infobits=True, return_state=True, hard_out=True, num_iter=12, cn_update="minsum")(g["llr_pic"], msg_v2c=state)
    assert np.array_equal(bh.cpu().numpy().astype(np.uint8), g["b_hat"])
    kb = phy.ofdm.KBestDetector("bit", n_ue, 64, rg, sm, **kw)(y, hf, ev, no).cpu().numpy()
    assert np.mean(np.isclose(kb, g["llr_kbest"], rtol=1e-4, atol=1e-3)) > 0.995
    ep = phy.ofdm.EPDetector("bit", rg, sm, m, l=10, hard_out=False)(y, hf, ev, no).cpu().numpy()
    assert np.mean(np.isclose(ep, g["llr_ep"], rtol=1e-3, atol=1e-2)) > 0.995
