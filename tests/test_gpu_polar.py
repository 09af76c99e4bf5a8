"""GPU parity tests of the CRC / Polar part of the hot path (north-star config C5) against the
reference's golden vectors (tests/golden/{crc,polar}_golden.npz) and oracle/polar.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import polar as op, polar_c as pc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CRC = np.load(os.path.join(GOLD, "crc_golden.npz"))
POL = np.load(os.path.join(GOLD, "polar_golden.npz"))


@pytest.fixture(scope="module")
def phy():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("pol", ["CRC24A", "CRC24B", "CRC24C", "CRC16", "CRC11", "CRC6"])
def test_crc_golden_and_oracle(phy, pol):
    enc = phy.fec.crc.CRCEncoder(pol)
    dec = phy.fec.crc.CRCDecoder(enc)
    u, ref = CRC[f"crc_u_{pol}"], CRC[f"crc_x_ref_np_{pol}"]
    x = _np(enc(u))
    assert np.array_equal(x.reshape(-1)[-len(ref):], ref) and (enc.k, enc.n) == (u.shape[-1], x.shape[-1])
    rng = np.random.default_rng(1)
    bits = rng.integers(0, 2, (7, 3, 333)).astype(np.float32)
    xc = enc(bits)
    assert np.array_equal(_np(xc), op.crc_encode(bits, pol))
    info, valid = dec(xc)
    assert np.array_equal(_np(info), bits) and bool(valid.all()) and valid.shape == (7, 3, 1)
    bad = _np(xc).copy()
    bad[..., 5] = 1 - bad[..., 5]
    assert not bool(dec(bad)[1].any())


@pytest.mark.parametrize("name", ["E45_k30_K41", "E70_k32_K43", "E127_k29_K40", "E1023_k400_K411", "E70_k28_K39"])
def test_polar5g_encoder_golden(phy, name):
    u, c_ref = POL[f"{name}_u"], POL[f"{name}_c"]
    enc = phy.fec.polar.Polar5GEncoder(u.shape[1], c_ref.shape[1])
    assert np.array_equal(_np(enc(u)), c_ref)


@pytest.mark.parametrize("k,n,ch", [(512, 1024, "uplink"), (40, 200, "downlink"), (100, 150, "uplink"), (500, 1088, "uplink"),
                                    (20, 576, "downlink"), (300, 400, "uplink")])
def test_polar5g_encoder_vs_oracle(phy, k, n, ch):
    enc = phy.fec.polar.Polar5GEncoder(k, n, channel_type=ch)
    u = np.random.default_rng(k).integers(0, 2, (2, 9, k)).astype(np.float32)
    assert np.array_equal(_np(enc(u)), op.Polar5GCode(k, n, ch).encode(u))
    # plain PolarEncoder
    fr, info = phy.fec.polar.generate_5g_ranking(64, 256)
    pe = phy.fec.polar.PolarEncoder(fr, 256)
    v = np.random.default_rng(0).integers(0, 2, (5, 64)).astype(np.float32)
    assert np.array_equal(_np(pe(v)), op.polar_encode(v, info, 256))


@pytest.mark.parametrize("name", ["P_128_37", "P_128_110", "P_256_128"])
def test_sc_and_scl1_golden(phy, name):
    a, lch, uhat = POL[f"{name}_Avec"], POL[f"{name}_Lch"], POL[f"{name}_uhat"]
    frozen = np.where(a == 0)[0]
    n = len(a)
    logits = (-1. * lch).astype(np.float32)
    assert np.array_equal(_np(phy.fec.polar.PolarSCDecoder(frozen, n)(logits)), uhat)
    for fast in (False, True):
        dec = phy.fec.polar.PolarSCLDecoder(frozen, n, list_size=1, use_fast_scl=fast)
        assert np.array_equal(_np(dec(logits)), uhat)


@pytest.mark.parametrize("n,k,L,crc,fast", [(128, 64, 8, None, True), (128, 64, 4, "CRC11", True), (256, 100, 8, "CRC11", False),
                                            (64, 40, 2, "CRC6", True), (1024, 523, 8, "CRC11", True), (512, 300, 16, "CRC24C", True),
                                            (256, 128, 32, "CRC11", True), (64, 30, 16, None, True), (32, 16, 8, None, True),
                                            (1024, 200, 4, "CRC24C", True), (512, 400, 8, "CRC16", True),
                                            (256, 128, 2, "CRC11", True), (512, 200, 1, None, True), (128, 80, 2, None, False),
                                            (1024, 700, 2, "CRC24C", True), (1024, 512, 1, "CRC11", False)])
def test_scl_vs_oracle(phy, n, k, L, crc, fast):
    """List decoding against the C oracle in the float32 specification arithmetic (oracle/polar_scl.c, the decoder
    whose float64 instantiation reproduces the reference's own NumPy twin): hard decisions and CRC status bit for
    bit.  Second witness: the NumPy float32 restatement of the TF path (libm softplus, NumPy summation order), which
    may pick another survivor at a near-tie."""
    frozen, info = phy.fec.polar.generate_5g_ranking(k, n)
    rng = np.random.default_rng(n + k + L)
    B = 48 if n >= 512 else 200
    kc = op.CRC_POLYS[crc][0] if crc else 0
    u = rng.integers(0, 2, (B, k - kc)).astype(np.float32)
    uc = op.crc_encode(u, crc) if crc else u
    c = op.polar_encode(uc, info, n)
    for sigma in (0.75, 0.95):
        y = (2 * c - 1) + sigma * rng.normal(size=c.shape)
        logits = (2 * y / sigma ** 2).astype(np.float32)
        dec = phy.fec.polar.PolarSCLDecoder(frozen, n, list_size=L, crc_degree=crc, use_fast_scl=fast,
                                            return_crc_status=crc is not None)
        out = dec(logits)
        got, status = (out if crc else (out, None))
        ref, ref_status = pc.SCLDecoder(frozen, n, L, crc, fast).decode(logits)
        assert np.array_equal(_np(got), ref), f"{(~np.all(_np(got) == ref, axis=1)).sum()} of {B} codewords differ"
        if crc:
            assert np.array_equal(_np(status).astype(bool), ref_status)
    ref2, _ = op.SCLDecoder(frozen, n, L, crc, fast).decode(logits)
    assert np.mean(np.all(_np(got) == ref2, axis=1)) >= 0.9


@pytest.mark.parametrize("ebno,B", [(2.5, 4096), (1.0, 2048)])
def test_c5_scl8_bit_exact_at_scale(phy, ebno, B):
    """BASELINE config 5 (Polar5G uplink k=512 -> n_polar=1024 with CRC11: 523 information positions, SCL-8, QPSK
    over AWGN): 4096 codewords at the bench's 2.5 dB (practically error free) and 2048 at 1.0 dB (in the waterfall,
    where list decisions are contested) through the GPU chain; decoded bits and CRC status equal the C oracle's on
    the same LLRs, bit for bit."""
    k, n, m = 512, 1024, 2
    enc = phy.fec.polar.Polar5GEncoder(k, n)
    dec = phy.fec.polar.Polar5GDecoder(enc, "SCL", list_size=8, return_crc_status=True)
    phy.config.seed = 55
    u = phy.mapping.BinarySource()([B, k])
    no = phy.utils.ebnodb2no(ebno, m, k / n)
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
    u_hat, status = dec(llr)
    ref, ref_status = pc.polar5g_decode(op.Polar5GCode(k, n, "uplink"), _np(llr), 8, return_crc_status=True)
    assert np.array_equal(_np(u_hat), ref), f"{(~np.all(_np(u_hat) == ref, axis=1)).sum()} of {B} codewords differ"
    assert np.array_equal(_np(status).astype(bool), ref_status)
    bler = np.mean(np.any(_np(u_hat) != _np(u), axis=1))
    if ebno < 2.0:
        assert 0.0 < bler < 0.9, bler              # both outcomes occur in the batch
    else:
        assert bler < 0.01, bler


# random (k, n, channel) draws over the 5G ranges: puncturing, shortening and repetition, with and without the
# uplink interleaver / the downlink input interleaver, odd CRC and parity-check-bit configurations
RANDOM_POLAR = [(595, 818, 'uplink'), (734, 931, 'uplink'), (84, 465, 'uplink'), (885, 981, 'uplink'),
                (351, 443, 'uplink'), (417, 1018, 'uplink'), (13, 339, 'downlink'), (405, 705, 'uplink'),
                (306, 450, 'uplink'), (409, 695, 'uplink'), (312, 918, 'uplink'), (63, 373, 'uplink'),
                (301, 1075, 'uplink'), (336, 1072, 'uplink'), (102, 201, 'downlink'), (57, 389, 'downlink'),
                (204, 865, 'uplink'), (560, 885, 'uplink'), (92, 731, 'uplink'), (423, 657, 'uplink'),
                (81, 291, 'downlink'), (307, 405, 'uplink'), (978, 1007, 'uplink'), (929, 1040, 'uplink'),
                (683, 710, 'uplink'), (861, 881, 'uplink'), (74, 757, 'uplink'), (100, 532, 'downlink')]


@pytest.mark.parametrize("k,n,ch", RANDOM_POLAR)
def test_polar5g_random_configs(phy, k, n, ch):
    """Encoder bit-exact against the oracle; noise-free logits decode back to the information bits with a passing
    CRC for SC, SCL-8 and hybrid SCL (rate matching / recovery and the interleavers are exercised end to end)."""
    enc = phy.fec.polar.Polar5GEncoder(k, n, channel_type=ch)
    u = np.random.default_rng(k * 2000 + n).integers(0, 2, (6, k)).astype(np.float32)
    c = _np(enc(u))
    assert np.array_equal(c, op.Polar5GCode(k, n, ch).encode(u))
    logits = (8.0 * (2 * c - 1)).astype(np.float32)
    for dec_type in ("SC", "SCL", "hybSCL"):
        dec = phy.fec.polar.Polar5GDecoder(enc, dec_type=dec_type, list_size=8, return_crc_status=True)
        u_hat, status = dec(logits)
        assert np.array_equal(_np(u_hat), u), dec_type
        assert np.all(_np(status)), dec_type


@pytest.mark.parametrize("k,n,ch,dec_type", [(64, 128, "uplink", "SCL"), (30, 70, "uplink", "SC"), (40, 200, "downlink", "SCL"),
                                             (100, 150, "uplink", "SCL"), (300, 1088, "uplink", "SCL"), (512, 1024, "uplink", "SCL")])
def test_polar5g_decoder_chain(phy, k, n, ch, dec_type):
    enc = phy.fec.polar.Polar5GEncoder(k, n, channel_type=ch)
    dec = phy.fec.polar.Polar5GDecoder(enc, dec_type=dec_type, list_size=8, return_crc_status=True)
    code = op.Polar5GCode(k, n, ch)
    rng = np.random.default_rng(k + n)
    B = 32
    u = rng.integers(0, 2, (B, k)).astype(np.float32)
    c = _np(enc(u))
    sigma = 0.6
    logits = (2 * ((2 * c - 1) + sigma * rng.normal(size=c.shape)) / sigma ** 2).astype(np.float32)
    u_hat, status = dec(logits)
    ref, ref_status = pc.polar5g_decode(code, logits, 8, return_crc_status=True, dec_type=dec_type)   # C oracle: bit for bit
    assert np.array_equal(_np(u_hat), ref) and np.array_equal(_np(status).astype(bool), ref_status)
    assert np.mean(np.all(_np(u_hat) == op.polar5g_decode(code, logits, dec_type, 8), axis=1)) >= 0.9   # NumPy witness
    ok = np.all(_np(u_hat) == u, axis=1)
    assert np.array_equal(_np(status)[ok], np.ones(ok.sum(), bool))        # decoded words pass their CRC
    assert ok.mean() > 0.5


@pytest.mark.parametrize("k,n,ch,sigma", [(64, 128, "uplink", 0.75), (300, 1088, "uplink", 1.15), (512, 1024, "uplink", 0.85),
                                           (40, 200, "downlink", 1.0)])
def test_hybrid_scl(phy, k, n, ch, sigma):
    """dec_type="hybSCL" (decoding.py:1292-1334): words whose CRC holds after SC keep the SC result, the
    others get exactly the SCL result."""
    enc = phy.fec.polar.Polar5GEncoder(k, n, channel_type=ch)
    mk = lambda t: phy.fec.polar.Polar5GDecoder(enc, dec_type=t, list_size=8, return_crc_status=True)
    rng = np.random.default_rng(k)
    B = 256
    u = rng.integers(0, 2, (B, k)).astype(np.float32)
    c = _np(enc(u))
    logits = (2 * ((2 * c - 1) + sigma * rng.normal(size=c.shape)) / sigma ** 2).astype(np.float32)
    u_h, st_h = mk("hybSCL")(logits)
    u_l, st_l = mk("SCL")(logits)
    u_s, st_s = mk("SC")(logits)
    u_h, u_l, u_s, st_h, st_l, st_s = (_np(t) for t in (u_h, u_l, u_s, st_h, st_l, st_s))
    if ch == "uplink":
        keep = st_s                                   # SC CRC status == the hybrid's switch
        assert 0 < keep.sum() < B, keep.sum()
        assert np.array_equal(u_h[keep], u_s[keep]) and np.array_equal(u_h[~keep], u_l[~keep])
        assert np.array_equal(st_h[keep], np.ones(keep.sum(), bool)) and np.array_equal(st_h[~keep], st_l[~keep])
    else:
        # downlink: the reference checks the SC CRC without undoing the input interleaver, so (almost)
        # every word takes the list decoder
        assert np.mean(np.all(u_h == u_l, axis=1)) > 0.99
    bler = lambda x: np.mean(np.any(x != u, axis=1))
    assert bler(u_l) <= bler(u_h) + 0.02 and bler(u_h) <= bler(u_s)
    with pytest.raises(ValueError):
        phy.fec.polar.PolarSCLDecoder(enc.frozen_pos, enc.n_polar, use_hybrid_sc=True)
    assert phy.fec.polar.Polar5GDecoder(enc, dec_type="BP").dec_type == "BP"   # the BP decoder: tests/test_gpu_polar_bp.py
    with pytest.raises(ValueError):
        phy.fec.polar.Polar5GDecoder(enc, dec_type="ML")


def test_c5_full_batch_properties(phy):
    """Config C5 (Polar5G uplink k=512 n=1024, SCL-8) on a large batch: noiseless round trip and
    monotone error rate."""
    enc = phy.fec.polar.Polar5GEncoder(512, 1024)
    dec = phy.fec.polar.Polar5GDecoder(enc, "SCL", list_size=8)
    phy.config.seed = 3
    u = phy.mapping.BinarySource()([4096, 512])
    c = enc(u)
    assert torch.equal(dec(8.0 * (2 * c - 1)).as_subclass(torch.Tensor), u.as_subclass(torch.Tensor))
    blers = []
    for sigma in (1.0, 0.7):
        y = (2 * c - 1) + sigma * torch.randn_like(c)
        u_hat = dec(2 * y / sigma ** 2)
        blers.append(float((u_hat != u).any(dim=1).float().mean()))
    assert blers[0] > blers[1] and blers[1] < 0.05


REFX = np.load(os.path.join(GOLD, "polar5g_ref_golden.npz"))


@pytest.mark.parametrize("i", range(len(REFX["cases"])))
def test_polar5g_chain_matches_reference_execution(phy, i):
    """The HIP encoder / decoders against the reference's OWN Polar5GEncoder / Polar5GDecoder (SC, the TensorFlow SCL-8 and
    its NumPy twin, SCL-4, hybrid SCL-8, CRC status) executed under the NumPy stand-in for TensorFlow
    (tests/golden/polar5g_ref_golden.npz, tools/gen_polar5g_ref_golden.py): codewords, decisions and CRC status bit for
    bit, at noise levels where SC loses blocks that the list recovers.  Oracle twin: tests/test_oracle_ref_exec_polar.py."""
    k, n, down, B = (int(v) for v in REFX["cases"][i])
    ch = "downlink" if down else "uplink"
    g = {key.split("/", 1)[1]: REFX[key] for key in REFX.files if key.startswith(f"{i}/")}
    unpack = lambda a, w: np.unpackbits(a, axis=1)[:, :w].astype(np.float32)
    enc = phy.fec.polar.Polar5GEncoder(k, n, channel_type=ch)
    assert enc.n_polar == int(g["n_polar"]) and np.array_equal(np.asarray(enc.frozen_pos), g["frozen_pos"])
    u = unpack(g["u"], k)
    assert np.array_equal(_np(enc(u)), unpack(g["c"], n))
    logits = g["logits"]
    assert np.array_equal(_np(phy.fec.polar.Polar5GDecoder(enc, dec_type="SC")(logits)), unpack(g["u_hat_sc"], k))
    for name, kw in (("scl8_tf", dict(dec_type="SCL", list_size=8)), ("scl4_tf", dict(dec_type="SCL", list_size=4)),
                     ("hyb8", dict(dec_type="hybSCL", list_size=8))):
        if f"u_hat_{name}" not in g:
            continue
        uh, st = phy.fec.polar.Polar5GDecoder(enc, return_crc_status=True, **kw)(logits)
        assert np.array_equal(_np(uh), unpack(g[f"u_hat_{name}"], k)), name
        assert np.array_equal(_np(st).astype(np.uint8).reshape(-1), g[f"crc_{name}"].reshape(-1)), name
