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


# round 6: return_state / msg_v2c on the GENERATED kernels (message image in / out, the [num_edges, batch] tensor deferred)
@pytest.mark.parametrize("k,n,m,bg", [(2816, 8448, 6, "bg1"), (768, 1536, 2, None), (1024, 2048, None, "bg1"), (1234, 2468, 4, None),
                                     (100, 200, None, None)])
def test_return_state_on_generated_kernels(k, n, m, bg):
    """Every code class of the generator (Z = 128 constants; several codewords per workgroup with the pair across two codewords;
    pairs inside a codeword; partly filled last chunk; tiny Z), the three rules, both output forms: outputs and states of
    chained calls equal the HBM-resident engine's bit for bit (min-sum family) and the oracle's msg_v2c, whether the state comes
    back untouched (never formed), as a copy (converted in) or was read in between (formed, image kept)."""
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    from sionna_amd.phy.block import pending_of
    from oracle.ldpc5g import LDPC5GCode
    from oracle import ldpc_bp as obp
    _ffi.device()
    T = lambda x: x.as_subclass(torch.Tensor)
    B = 37
    phy.config.seed = 5
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    mm = m or 2
    no = phy.utils.ebnodb2no(2.0, mm, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", mm)(phy.channel.AWGN()(phy.mapping.Mapper("qam", mm)(enc(u)), no), no)
    for cn in ("minsum", "offset-minsum", "boxplus-phi"):
        for rib in (True, False):
            dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=3, hard_out=False, return_state=True, return_infobits=rib)
            ref = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=3, hard_out=False, return_state=True, return_infobits=rib)
            ref._onchip_ok = False
            x1, s1 = dec(llr)
            assert dec._state_lay[1] is not None, "the generated kernel with state did not run"
            assert pending_of(s1) is not None
            x2, s2 = dec(llr, msg_v2c=s1)                          # untouched: the image goes straight back
            assert pending_of(s1) is not None
            s1c = T(s1).clone()                                     # reading forms it ...
            assert pending_of(s1) is None
            x2b, _ = dec(llr, msg_v2c=s1)                           # ... formed, not modified: the kept image
            x3, s3 = dec(llr, msg_v2c=s1c)                          # a copy: converted in
            y1, t1 = ref(llr)
            y2, t2 = ref(llr, msg_v2c=t1)
            pairs = ((x1, y1), (s1, t1), (x2, y2), (s2, t2), (x2b, y2), (x3, y2), (s3, t2))
            if cn == "boxplus-phi":
                assert all(torch.allclose(T(a), T(b), rtol=1e-4, atol=1e-3) for a, b in pairs), (cn, rib)
            else:
                assert all(torch.equal(T(a), T(b)) for a, b in pairs), (cn, rib)
            s1.mul_(1.0)                                            # an in-place operation: the kept image no longer counts
            x4, _ = dec(llr, msg_v2c=s1)
            assert torch.equal(T(x4), T(x2))
    code = LDPC5GCode(k, n, m, enc._bg)
    od = obp.LDPC5GDecoder(code, cn_update="minsum", hard_out=False, return_infobits=True, num_iter=3, return_state=True)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=3, hard_out=False, return_state=True)
    x, s = dec(llr[:5])
    xo, so = od.decode(od.rate_recover(T(llr[:5]).cpu().numpy()))
    assert np.array_equal(T(x).cpu().numpy(), xo[:, :k]) and np.array_equal(T(s).cpu().numpy(), so)


def test_return_state_shape_errors_on_generated_kernels():
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    enc = phy.fec.ldpc.LDPC5GEncoder(768, 1536)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=2, hard_out=False, return_state=True)
    llr = torch.randn(8, 1536, device="cuda")
    _, s = dec(llr)
    with pytest.raises(ValueError):
        dec(llr[:4], msg_v2c=s)


def test_state_conversion_float4_path():
    """one codeword per pass, a batch of whole float4s and full 64-pass tiles: samd_ldpc5g_state_convert_f32 moves 16 bytes per
    lane on the reference's side (both directions), plus a ragged last tile"""
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    T = lambda x: x.as_subclass(torch.Tensor)
    k, n, m = 2816, 8448, 6
    phy.config.seed = 9
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    no = phy.utils.ebnodb2no(3.0, m, k / n)
    for B in (128, 200):
        u = phy.mapping.BinarySource()([B, k])
        llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=2, hard_out=False, return_state=True)
        ref = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=2, hard_out=False, return_state=True)
        ref._onchip_ok = False
        x1, s1 = dec(llr)
        y1, t1 = ref(llr)
        assert dec._state_lay[1] is not None and dec._state_lay[1][1] == 1
        assert torch.equal(T(s1), T(t1)) and torch.equal(T(x1), T(y1))
        x2, s2 = dec(llr, msg_v2c=T(t1).clone())                  # a foreign tensor: converted in
        y2, t2 = ref(llr, msg_v2c=t1)
        assert torch.equal(T(x2), T(y2)) and torch.equal(T(s2), T(t2))
