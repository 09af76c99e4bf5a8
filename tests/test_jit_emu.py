"""The specialised 5G LDPC decoder (csrc/ldpc5g_jit.cpp) held to the oracle WITHOUT a GPU.

libsionna_amd.so generates, for one code, straight-line per-wave programs (block offsets, shifts, rate-matching offsets as
constants) and compiles them with hipRTC for gfx950.  The generator is host code: here it runs on a handle built under
SAMD_HOST_ONLY, the generated text (node updates csrc/jit/ldpc5g_jit_templates.h + per-wave programs) is compiled with g++
against 64-wide CPU stand-ins of the per-lane operations (tests/jit_emu) and executed with 16 threads per workgroup - the
schedule, every address and offset and the node arithmetic are what the GPU will run; only the operation definitions
differ.  Soft outputs must be array_equal to oracle/ldpc_bp.c.  (`-m gpu` twin: tests/test_gpu_jit.py.)
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ldpc5g import LDPC5GCode          # noqa: E402
from oracle import ldpc_bp as obp, cbind      # noqa: E402
from tools import jit_dump                    # noqa: E402

EMU = os.path.join(ROOT, "tests", "jit_emu")


def _build_emu(tmp_path, h, infobits, tag, cn="minsum"):
    src = jit_dump.jit_source(h, infobits, 0, cn)
    inc = tmp_path / f"emu_src_{tag}.h"
    inc.write_text(src)
    so = tmp_path / f"emu_{tag}.so"
    subprocess.check_call(["g++", "-O0", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", "-Wno-unknown-pragmas",
                           "-pthread", f"-DJIT_EMU_SRC=\"{inc}\"", "-I", EMU, "-o", str(so),
                           os.path.join(EMU, "jit_emu_main.cpp")])
    lib = C.CDLL(str(so))
    lib.jit_emu_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int]
    return lib, src


def _noisy_llr(code, batch, seed, sigma=0.8):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 2, (batch, code.k)).astype(np.float32)
    c = code.encode(u)
    y = (2 * c - 1) + sigma * rng.normal(size=c.shape)
    return (2 * y / sigma ** 2).astype(np.float32)


def _reference(code, llr, cn, it, infobits, m, hard, offset=0.5):
    odec = obp.LDPC5GDecoder(code, cn_update=cn, hard_out=False, return_infobits=infobits, num_iter=it)
    xr = cbind.bp_decode(odec, odec.rate_recover(llr), num_iter=it, hard_out=0, offset=offset)
    if infobits:
        ref = xr[:, :code.k]
    else:                                                        # decoding.py:1506-1531
        x_nf = np.concatenate([xr[:, :code.k], xr[:, code.k_ldpc:]], axis=1)
        ref = x_nf[:, 2 * code.z:2 * code.z + code.n]
        if m is not None:
            ref = ref[:, code.out_int]
    return (ref >= 0).astype(np.float32) if hard else ref        # hard: 0 >= x_hat (internal sign) = returned logit >= 0


# BASELINE config C2; the same code without interleaver; rate 1/2 at Z = 128 (pruned graph); Z = 256 at rate 2/3
CODES = [(2816, 8448, "bg1", 6), (2816, 8448, "bg1", None), (2816, 5632, "bg1", 2), (5632, 8448, "bg1", None)]


# generator options exercised besides the defaults: the interleaved message layout (8-byte DS instructions, Z = 128), an own
# LPT schedule with pipelined loads and the second chunk's positions by xor
VARIANTS = [{}, {"SAMD_JIT_LAYOUT": "1"}, {"SAMD_JIT_SCHED": "1", "SAMD_JIT_PIPE": "2", "SAMD_JIT_XOR128": "1", "SAMD_JIT_PREFETCH": "0"},
            {"SAMD_JIT_CMP_AHEAD": "2"}, {"SAMD_JIT_WAVES": "12"}, {"SAMD_JIT_PHI_ROLLED": "0"},
            {"SAMD_JIT_VST32": "1"}, {"SAMD_JIT_A1": "0"}, {"SAMD_JIT_PHI_TAB32": "1"},
            {"SAMD_JIT_PHI_TAB0": "0", "SAMD_JIT_PHI_LEAN": "0"}, {"SAMD_JIT_PREFETCH": "1"}]


@pytest.mark.parametrize("k,n,bg,m", CODES)
@pytest.mark.parametrize("variant", range(len(VARIANTS)))
def test_generated_programs_match_oracle(tmp_path, k, n, bg, m, variant):
    from sionna_amd import _ffi
    if variant and (k, n, m) != (2816, 8448, 6):
        pytest.skip("generator variants are exercised on the C2 code")
    code = LDPC5GCode(k, n, m, bg)
    h = jit_dump.host_only_handle(k, n, m, bg)
    assert _ffi.lib().samd_ldpc5g_jit_supported(h) == 1
    for kk, vv in VARIANTS[variant].items():
        _ffi.set_option(kk, vv)
    try:
        _run_generated(tmp_path, code, h, k, n, m, full=variant == 0)
    finally:
        for kk in VARIANTS[variant]:
            _ffi.set_option(kk, None)
    _ffi.lib().samd_ldpc5g_destroy(h)


def _run_generated(tmp_path, code, h, k, n, m, full=True):
    batch, grid = 5, 2                                           # workgroup 0 decodes 3 codewords in sequence, workgroup 1 two
    llr = _noisy_llr(code, batch, k + n)
    llr[0, :7] = 0
    llr[1] = np.round(llr[1])                                    # exact ties
    llr[2, ::5] *= 40                                            # clipping
    # (a generator variant: both output forms on the min-sum kernel, one on the others - each build is ~6 s of g++)
    for infobits in (1, 0):
      for rule, cases in (("minsum", (("minsum", 1, 0), ("minsum", 6, 0), ("minsum", 3, 1))),
                          ("offset-minsum", (("offset-minsum", 4, 0), ("minsum", 2, 0))),      # one kernel per rule (offset 0 = min-sum)
                          ("boxplus-phi", (("boxplus-phi", 1, 0), ("boxplus-phi", 5, 0), ("boxplus-phi", 3, 1)))):
        if not full and (rule == "offset-minsum" or (rule == "boxplus-phi" and infobits == 0)):
            continue
        lib, src = _build_emu(tmp_path, h, infobits, f"{k}_{n}_{m}_{infobits}_{rule}", rule)
        assert "jit_wave_11" in src
        for cn, it, hard in cases:
            out = np.full((batch, k if infobits else n), np.nan, np.float32)
            x = np.ascontiguousarray(llr)
            lib.jit_emu_decode(x.ctypes.data, out.ctypes.data, batch, it, 20.0, 0.5 if cn == "offset-minsum" else 0.0,
                               hard, grid)
            ref = _reference(code, llr, cn, it, bool(infobits), m, hard)
            assert np.array_equal(out, ref), f"{cn} it={it} infobits={infobits} hard={hard}: " \
                                             f"{np.mean(out != ref):.3e} differ, nan {np.isnan(out).sum()}"


# round 6: the any-lifting-size programs.  BASELINE C4's code (BG2, Z = 80: six codewords per workgroup, pruned tail, fillers,
# interleaver), C1's (BG1, Z = 48), a partly filled last chunk with k / n not multiples of anything, tiny Z (35 codewords per
# workgroup), a Z = 128 code outside the constant-offset class
GENERAL = [(768, 1536, None, 2, ("minsum", "offset-minsum", "boxplus-phi")), (1024, 2048, "bg1", None, ("minsum", "boxplus-phi")),
           (1234, 2468, None, 4, ("offset-minsum",)), (100, 200, None, None, ("minsum",)), (2816, 8436, "bg1", 6, ("minsum",)),
           (6144, 9216, "bg1", None, ("minsum",))]     # (messages beyond LDS: the last base rows' blocks in the workspace row)


@pytest.mark.parametrize("k,n,bg,m,rules", GENERAL)
def test_generated_programs_any_lifting_size_match_oracle(tmp_path, k, n, bg, m, rules):
    from sionna_amd import _ffi
    h, enc, _ = jit_dump.host_only_handle(k, n, m, bg, return_obj=True)
    code = LDPC5GCode(k, n, m, enc._bg)
    assert _ffi.lib().samd_ldpc5g_jit_supported(h) == 1
    for infobits in (1, 0):
        for rule in rules:
            if infobits == 0 and rule != rules[0]:
                continue
            lib, src = _build_emu(tmp_path, h, infobits, f"g{k}_{n}_{infobits}_{rule}", rule)
            assert "#define JIT_GENERAL 1" in src
            group = int(src.split("// JIT_GROUP ")[1].split()[0])
            batch, grid = 2 * group + 1, 2                            # workgroup 0: two groups (the second one not full), workgroup 1: one
            llr = _noisy_llr(code, batch, k + n)
            llr[0, :7] = 0
            llr[1] = np.round(llr[1])
            llr[2, ::5] *= 40
            for it, hard in ((1, 0), (4, 0), (3, 1)):
                out = np.full((batch, k if infobits else n), np.nan, np.float32)
                x = np.ascontiguousarray(llr)
                lib.jit_emu_decode(x.ctypes.data, out.ctypes.data, batch, it, 20.0, 0.5 if rule == "offset-minsum" else 0.0, hard, grid)
                ref = _reference(code, llr, rule, it, bool(infobits), m, hard)
                assert np.array_equal(out, ref), f"{rule} it={it} infobits={infobits} hard={hard}: {np.mean(out != ref):.3e} differ"
    _ffi.lib().samd_ldpc5g_destroy(h)


def test_jit_class_boundaries():
    """odd lifting sizes and codes whose messages exceed LDS keep the generic kernels; every other code has a generated one"""
    from sionna_amd import _ffi
    for k, n, bg, m, want in ((1024, 2048, "bg1", None, 1), (2816, 8436, "bg1", 6, 1), (768, 1536, None, 2, 1), (30, 90, None, None, 1),
                              (8448, 25344, "bg1", None, 0)):
        h, enc, _ = jit_dump.host_only_handle(k, n, m, bg, return_obj=True)
        assert _ffi.lib().samd_ldpc5g_jit_supported(h) == want, (k, n, bg, m, enc._z)
        _ffi.lib().samd_ldpc5g_destroy(h)


# round 6: the state variant (return_state / msg_v2c on the generated kernels).  The image a workgroup pass writes, mapped with
# samd_ldpc5g_state_map, must be the oracle's msg_v2c; decoding on from an image must equal decoding in one go.
STATE_CODES = [(2816, 8448, "bg1", 6, "minsum"), (2816, 8448, "bg1", 6, "boxplus-phi"), (768, 1536, None, 2, "minsum"),
               (1024, 2048, "bg1", None, "minsum"), (1234, 2468, None, 4, "offset-minsum")]


@pytest.mark.parametrize("k,n,bg,m,rule", STATE_CODES)
def test_generated_state_variant_matches_oracle_state(tmp_path, k, n, bg, m, rule):
    from sionna_amd import _ffi
    lib0 = _ffi.lib()
    h, enc, _ = jit_dump.host_only_handle(k, n, m, bg, return_obj=True)
    code = LDPC5GCode(k, n, m, enc._bg)
    mode = _ffi.CN_MODES[rule]
    img, cwpp = C.c_int(), C.c_int()
    rc = lib0.samd_ldpc5g_state_layout(h, mode, C.byref(img), C.byref(cwpp))
    assert rc == 0, lib0.samd_last_error().decode()
    img, cwpp = img.value, cwpp.value
    cw, cn, vn = (np.empty(img, np.int32) for _ in range(3))
    assert lib0.samd_ldpc5g_state_map(h, mode, cw.ctypes.data_as(C.c_void_p), cn.ctypes.data_as(C.c_void_p),
                                      vn.ctypes.data_as(C.c_void_p)) == 0
    _ffi.set_option("SAMD_JIT_STATE", "1")
    try:
        lib, src = _build_emu(tmp_path, h, 1, f"st{k}_{n}_{rule}", rule)
    finally:
        _ffi.set_option("SAMD_JIT_STATE", None)
    assert "jit_copy_l2g" in src and f"// JIT_IMG_BYTES {4 * img}" in src
    lib.jit_emu_decode_state.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p]
    batch, grid = 2 * cwpp + 1, 2
    passes = -(-batch // cwpp)
    llr = _noisy_llr(code, batch, k + n + 1)
    llr[1] = np.round(llr[1])
    off = 0.5 if rule == "offset-minsum" else 0.0

    def run(it, st_in, want):
        out = np.full((batch, k), np.nan, np.float32)
        st_out = np.full((passes, img), np.nan, np.float32) if want else None
        lib.jit_emu_decode_state(llr.ctypes.data, out.ctypes.data, batch, it, 20.0, off, 0, grid,
                                 None if st_in is None else st_in.ctypes.data, None if st_out is None else st_out.ctypes.data)
        return out, st_out

    out5, im5 = run(5, None, True)
    _, im2 = run(2, None, True)
    out23, im23 = run(3, im2, True)
    assert np.array_equal(out5, out23) and np.array_equal(im5, im23)
    assert np.array_equal(run(5, None, False)[0], out5)
    # the oracle's state (VN-major edge order, logit sign) against the mapped image
    odec = obp.LDPC5GDecoder(code, cn_update=rule, hard_out=False, return_infobits=True, num_iter=2, return_state=True)
    key = odec.vn_idx.astype(np.int64) * odec.num_cns + odec.cn_idx
    assert np.all(np.diff(key) > 0)
    live = cw >= 0
    e = np.searchsorted(key, vn[live].astype(np.int64) * odec.num_cns + cn[live])
    assert np.array_equal(key[e], vn[live].astype(np.int64) * odec.num_cns + cn[live])
    assert len(np.unique(np.stack([cw[live], e]), axis=1).T) == live.sum() == cwpp * odec.num_edges
    if rule != "offset-minsum":                                  # (the NumPy oracle's offset rule takes its default offset)
        _, st = odec.decode(odec.rate_recover(llr), num_iter=2)      # [E, B]
        got = np.full_like(st, np.nan)
        q = np.nonzero(live)[0]
        for p_ in range(passes):
            b = p_ * cwpp + cw[q]
            ok = b < batch
            got[e[ok], b[ok]] = -im2[p_, q[ok]]
        if rule == "minsum":
            assert np.array_equal(got, st)
        else:
            assert np.mean(np.isclose(got, st, rtol=1e-4, atol=1e-3)) > 0.999
    lib0.samd_ldpc5g_destroy(h)
