"""The specialised 5G LDPC decoder (csrc/ldpc5g_jit.cpp: per-wave straight-line programs generated for one code, compiled
with hipRTC for gfx950) on the GPU: soft outputs array_equal to oracle/ldpc_bp.c and to the generic kernel, and the
specialised kernel is the one that ran (samd_ldpc5g_jit_launches).  CPU twin of the generated source:
tests/test_jit_emu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.ldpc5g import LDPC5GCode
from oracle import ldpc_bp as obp, cbind


@pytest.fixture(scope="module")
def phy():
    import sionna_amd.phy as p
    from sionna_amd import _ffi
    _ffi.device()
    return p


def _np(t):
    return t.detach().cpu().numpy()


def _opt(key, value="1"):
    from sionna_amd import _ffi
    return _ffi.option(key, value)


def _launches(enc, dec):
    from sionna_amd import _ffi
    return int(_ffi.lib().samd_ldpc5g_jit_launches(enc._handle(dec._nb_pruned_nodes)))


def _noisy_llr(code, batch, seed, sigma=0.8):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 2, (batch, code.k)).astype(np.float32)
    c = code.encode(u)
    y = (2 * c - 1) + sigma * rng.normal(size=c.shape)
    return (2 * y / sigma ** 2).astype(np.float32)


def _reference(code, llr, cn, it, infobits, m):
    odec = obp.LDPC5GDecoder(code, cn_update=cn, hard_out=False, return_infobits=infobits, num_iter=it)
    xr = cbind.bp_decode(odec, odec.rate_recover(llr), num_iter=it, hard_out=0)
    if infobits:
        return xr[:, :code.k]
    x_nf = np.concatenate([xr[:, :code.k], xr[:, code.k_ldpc:]], axis=1)          # decoding.py:1506-1531
    ref = x_nf[:, 2 * code.z:2 * code.z + code.n]
    return ref[:, code.out_int] if m is not None else ref


CODES = [(2816, 8448, "bg1", 6), (2816, 8448, "bg1", None), (2816, 5632, "bg1", 2), (5632, 8448, "bg1", None)]
# codes of the any-lifting-size programs (round 6): BASELINE C4's code (BG2, Z = 80, six codewords per workgroup) and C1's (BG1,
# Z = 48), fillers / k, n not multiples of 64 / pruned tail of the last base row / a partly filled last chunk / tiny Z
GENERAL_CODES = [(768, 1536, None, 2), (1024, 2048, "bg1", None), (100, 200, None, None), (4000, 6000, None, 4),
                 (3000, 4500, "bg1", 6), (20, 60, None, None), (2816, 8436, "bg1", 6), (1234, 2468, None, 4),
                 # messages beyond LDS: the last base rows' blocks in the L2 workspace row of the workgroup, odd lifting size
                 (6144, 9216, "bg1", None), (5632, 11264, "bg1", 2), (64, 128, None, None)]


@pytest.mark.parametrize("k,n,bg,m", CODES + GENERAL_CODES)
@pytest.mark.parametrize("grid", [None, "2"])
def test_specialised_kernel_bit_exact_vs_oracle(phy, k, n, bg, m, grid):
    """small batches through the specialised kernel (SAMD_LDPC_JIT=2: any batch size); grid = 2 workgroups: each decodes
    several codewords in sequence (the codeword loop with its prefetch of the next codeword's channel values)"""
    code = LDPC5GCode(k, n, m, bg)
    llr = _noisy_llr(code, 7 if (k, n, bg, m) in CODES else 150, k + n)    # (several groups of codewords per workgroup)
    llr[0, :7] = 0
    llr[1] = np.round(llr[1])
    llr[2, ::5] *= 40
    import contextlib
    with _opt("SAMD_LDPC_JIT", "2"), (_opt("SAMD_ONCHIP_GRID", grid) if grid else contextlib.nullcontext()):
        enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
        for cn, it, infobits in (("minsum", 1, True), ("minsum", 6, False), ("offset-minsum", 5, True), ("minsum", 20, True)):
            dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, hard_out=False, return_infobits=infobits, num_iter=it)
            before = _launches(enc, dec)
            got = _np(dec(llr))
            assert _launches(enc, dec) == before + 1, "the specialised kernel did not run"
            ref = _reference(code, llr, cn, it, infobits, m)
            assert np.array_equal(got, ref), f"{cn} it={it} infobits={infobits}: {np.mean(got != ref):.3e} differ"
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=8)            # hard decisions
        ref = _reference(code, llr, "minsum", 8, True, m)
        assert np.array_equal(_np(dec(llr)), (0 >= -ref).astype(np.float32))


@pytest.mark.parametrize("opts", [{"SAMD_JIT_LAYOUT": "1"}, {"SAMD_JIT_LAYOUT": "1", "SAMD_JIT_PIPE": "2", "SAMD_JIT_PREFETCH": "0"},
                                  {"SAMD_JIT_SCHED": "1", "SAMD_JIT_PIPE": "2", "SAMD_JIT_XOR128": "1"}, {"SAMD_JIT_CMP_AHEAD": "2"}])
def test_generator_variants_bit_exact(phy, opts):
    """the generator's other forms of the C2 kernel (interleaved message layout with 8-byte DS instructions, own schedule,
    pipelined loads, xor positions): the same soft outputs as the oracle"""
    import contextlib
    k, n, m, bg = 2816, 8448, 6, "bg1"
    code = LDPC5GCode(k, n, m, bg)
    llr = _noisy_llr(code, 9, 99)
    llr[0, :9] = 0
    llr[1] = np.round(llr[1])
    with contextlib.ExitStack() as st:
        st.enter_context(_opt("SAMD_LDPC_JIT", "2"))
        st.enter_context(_opt("SAMD_ONCHIP_GRID", "2"))
        for kk, vv in opts.items():
            st.enter_context(_opt(kk, vv))
        enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
        for cn, it, infobits in (("minsum", 7, True), ("offset-minsum", 4, False)):
            dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, hard_out=False, return_infobits=infobits, num_iter=it)
            before = _launches(enc, dec)
            got = _np(dec(llr))
            assert _launches(enc, dec) == before + 1
            assert np.array_equal(got, _reference(code, llr, cn, it, infobits, m)), (cn, opts)


@pytest.mark.parametrize("k,n,bg,m,rolled", [(2816, 8448, "bg1", 6, "1"), (2816, 8448, "bg1", 6, "0"), (768, 1536, None, 2, "1"),
                                             (1024, 2048, "bg1", None, "1"), (1234, 2468, None, 4, "0")])
def test_boxplus_phi_on_the_generated_kernel_bit_exact(phy, k, n, bg, m, rolled):
    """the defined phi of round 5 inside a generated kernel - round 6: check-node loops rolled (the default; the unrolled form
    stays selectable), every even lifting size: soft outputs array_equal to the oracle, both output forms"""
    import contextlib
    code = LDPC5GCode(k, n, m, bg)
    llr = _noisy_llr(code, 6 if k == 2816 else 50, 7, sigma=0.7)
    llr[0, :9] = 0
    with contextlib.ExitStack() as st:
        for kk, vv in (("SAMD_LDPC_JIT", "2"), ("SAMD_JIT_PHI_ROLLED", rolled), ("SAMD_ONCHIP_GRID", "2")):
            st.enter_context(_opt(kk, vv))
        enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
        for it, infobits in ((1, True), (6, False)):
            dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="boxplus-phi", hard_out=False, return_infobits=infobits, num_iter=it)
            before = _launches(enc, dec)
            got = _np(dec(llr))
            assert _launches(enc, dec) == before + 1, "the generated kernel did not run"
            assert np.array_equal(got, _reference(code, llr, "boxplus-phi", it, infobits, m)), (it, infobits)


def test_c2_at_scale_specialised_equals_generic_and_oracle(phy):
    """BASELINE config C2 in the waterfall, 4096 codewords, 20 iterations: the default policy picks the specialised kernel
    (batch >= 1024); its soft outputs equal the generic kernel's (SAMD_LDPC_JIT=0) and the oracle's on a sample"""
    k, n, m, B = 2816, 8448, 6, 4096
    phy.config.seed = 4243
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    no = phy.utils.ebnodb2no(4.0, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
    for cn in ("minsum", "offset-minsum"):
        dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=20, hard_out=False)
        before = _launches(enc, dec)
        got = _np(dec(llr))
        assert _launches(enc, dec) == before + 1, "the specialised kernel did not run"
        with _opt("SAMD_LDPC_JIT", "0"):
            enc0 = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
            dec0 = phy.fec.ldpc.LDPC5GDecoder(enc0, cn_update=cn, num_iter=20, hard_out=False)
            gen = _np(dec0(llr))
            assert _launches(enc0, dec0) == 0
        assert np.array_equal(got, gen), f"{cn}: {np.mean(got != gen):.3e} of the soft outputs differ from the generic kernel"
        code = LDPC5GCode(k, n, m, "bg1")
        odec = obp.LDPC5GDecoder(code, cn_update=cn, num_iter=20, hard_out=False)
        ref = cbind.bp_decode(odec, odec.rate_recover(_np(llr[:256])))[:, :k]
        assert np.array_equal(got[:256], ref)
    frac_err = np.mean(np.any((got > 0) != (_np(u) > 0), axis=1))
    assert 0.0 < frac_err < 1.0


def test_small_batches_keep_the_generic_kernel(phy):
    """default policy: below SAMD_LDPC_JIT_MIN_BATCH codewords nothing is compiled"""
    k, n, m = 2816, 8448, 6
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg="bg1")
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=3)
    dec(torch.zeros((8, n), device="cuda"))
    assert _launches(enc, dec) == 0
