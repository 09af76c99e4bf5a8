"""CPU-side tests: the C-ABI library loads and exports every symbol of include/sionna_amd.h,
host-side construction matches the oracle, API surface and error behaviour mirror the
reference (constructor checks of decoding.py:191-271 / encoding.py:71-102)."""
import numpy as np
import pytest
import scipy.sparse as sp

from sionna_amd import _ffi
import sionna_amd.phy as phy
from sionna_amd.phy.fec.ldpc import LDPC5GEncoder, LDPC5GDecoder, LDPCBPDecoder
from oracle.ldpc5g import LDPC5GCode
from oracle import ldpc_bp as obp, mapping as omap, utils as outil


def test_library_exports_every_declared_symbol():
    lib = _ffi.lib()
    decl = _ffi.declared_symbols()
    assert len(decl) >= 19
    assert [s for s in decl if not hasattr(lib, s)] == []
    assert set(decl) == set(_ffi._SIGNATURES), "ctypes table out of sync with the header"
    assert lib.samd_version() >= 100


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        phy.mapping.Mapper("qam", 2)(np.zeros((1, 4), np.float32))
    with pytest.raises(RuntimeError):
        LDPC5GEncoder(100, 200)(np.zeros((1, 100), np.float32))


@pytest.mark.parametrize("k,n,bg,m", [(1024, 2048, "bg1", None), (2816, 8448, "bg1", 6), (64, 128, None, 2),
                                      (500, 1000, None, None), (3840, 4800, "bg2", 4), (8448, 25344, None, None),
                                      (12, 20, None, None), (292, 900, None, None), (3825, 5000, None, None)])
def test_code_construction_matches_oracle(k, n, bg, m):
    e = LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    c = LDPC5GCode(k, n, num_bits_per_symbol=m, bg=bg)
    assert (e.z, e._i_ls, e.k_ldpc, e.n_ldpc, e._bg) == (c.z, c.i_ls, c.k_ldpc, c.n_ldpc, c.bg)
    assert (e.pcm != c.pcm).nnz == 0
    if m:
        assert np.array_equal(e.out_int, c.out_int) and np.array_equal(e.out_int_inv, c.out_int_inv)
    for prune in (True, False):
        d = LDPC5GDecoder(e, cn_update="minsum", prune_pcm=prune)
        od = obp.LDPC5GDecoder(c, cn_update="minsum", prune_pcm=prune)
        assert (d.num_vns, d.num_cns, d.num_edges, d._nb_pruned_nodes) == (od.num_vns, od.num_cns, od.num_edges, od.nb_pruned)
        assert np.array_equal(d._cn_idx, od.cn_idx) and np.array_equal(d._vn_idx, od.vn_idx)


def test_constellations_match_oracle():
    for m in (2, 4, 6, 8, 10):
        assert np.array_equal(phy.mapping.qam(m), omap.qam(m))
        assert abs(np.mean(np.abs(phy.mapping.qam(m)) ** 2) - 1) < 1e-6
    assert phy.mapping.pam_gray([0, 1, 1]) == omap.pam_gray(np.array([0, 1, 1]))
    c = phy.mapping.Constellation("qam", 4)
    assert c.num_points == 16 and c.points.dtype == np.complex64
    with pytest.raises(ValueError):
        phy.mapping.Constellation("qam", 3)
    with pytest.raises(ValueError):
        c.points = np.zeros(16)


def test_ebnodb2no_and_hard_decisions():
    for db in (-3.0, 0.0, 4.5):
        assert phy.utils.ebnodb2no(db, 6, 1 / 3) == outil.ebnodb2no(db, 6, 1 / 3)
    x = np.array([-1.0, 0.0, 2.0], np.float32)
    assert np.array_equal(phy.utils.hard_decisions(x), [0, 0, 1])


def test_constructor_errors_mirror_reference():
    with pytest.raises(TypeError):
        LDPC5GEncoder("a", 10)
    with pytest.raises(ValueError):
        LDPC5GEncoder(9000, 20000)
    with pytest.raises(ValueError):
        LDPC5GEncoder(100, 600)                    # r < 1/5
    with pytest.raises(ValueError):
        LDPC5GEncoder(1000, 4000, bg="bg1")        # r < 1/3 on BG1
    with pytest.raises(ValueError):
        LDPC5GEncoder(100, 200, bg="bg3")
    enc = LDPC5GEncoder(100, 200, unknown_kwarg=True)          # swallowed like block.py:25
    pcm = np.array([[1, 1, 0], [0, 1, 1]], np.float32)
    with pytest.raises(TypeError):
        LDPCBPDecoder(pcm, hard_out="yes")
    with pytest.raises(TypeError):
        LDPCBPDecoder(pcm, num_iter=1.5)
    with pytest.raises(ValueError):
        LDPCBPDecoder(pcm, num_iter=-1)
    with pytest.raises(ValueError):
        LDPCBPDecoder(pcm * 2)
    with pytest.raises(TypeError):
        LDPCBPDecoder([[1, 0]])
    with pytest.raises(TypeError):
        LDPCBPDecoder(pcm, cn_update="nope")
    with pytest.raises(TypeError):
        LDPCBPDecoder(pcm, cn_type="minsum")
    with pytest.raises(TypeError):
        LDPC5GDecoder(pcm)
    with pytest.raises(TypeError):
        LDPC5GDecoder(enc, return_infobits=1)
    with pytest.raises(TypeError):
        LDPCBPDecoder(pcm, c2v_callbacks=3)
    dcb = LDPCBPDecoder(pcm, c2v_callbacks=[lambda m, it: m])   # accepted: runs on the torch engine of custom.py ...
    assert dcb._custom
    with pytest.raises(RuntimeError):                           # ... on the device only - no CPU fallback
        dcb(np.zeros((1, 3), np.float32))
    d = LDPCBPDecoder(sp.csr_matrix(pcm), cn_update="minsum", num_iter=3)
    assert (d.num_cns, d.num_vns, d.num_edges, d.num_iter) == (2, 3, 4, 3)
    d.num_iter = 5
    assert d.num_iter == 5
    with pytest.raises(ValueError):
        d.llr_max = -1
    assert phy.config.precision == "single"
    with pytest.raises(ValueError):
        phy.config.precision = "half"


def test_rng_streams_differ_per_rank():
    from sionna_amd.phy.config import PhiloxGenerator
    g0, g1 = PhiloxGenerator(42, rank=0), PhiloxGenerator(42, rank=1)
    assert g0.seed == 42 and g1.seed != 42
    a = outil.random_bits(g0.seed, 0, 4096)
    b = outil.random_bits(g1.seed, 0, 4096)
    assert 0.4 < np.mean(a != b) < 0.6
    assert g0.next_call() == 0 and g0.next_call() == 1


def test_tensor_shape_helpers():
    """utils/tensors.py of the reference: pure reshapes (reference tests test/unit/utils/test_tensors.py)."""
    import torch
    from sionna_amd.phy import utils as u
    x = torch.arange(24.).reshape(2, 3, 4)
    assert tuple(u.expand_to_rank(x, 5, axis=-1).shape) == (2, 3, 4, 1, 1)
    assert tuple(u.expand_to_rank(x, 5, axis=0).shape) == (1, 1, 2, 3, 4)
    assert tuple(u.expand_to_rank(x, 2).shape) == (2, 3, 4)
    assert tuple(u.insert_dims(x, 2, axis=1).shape) == (2, 1, 1, 3, 4) and tuple(u.insert_dims(x, 1, axis=-2).shape) == (2, 3, 1, 4)
    assert tuple(u.flatten_dims(x, 2, 0).shape) == (6, 4) and tuple(u.flatten_dims(x, 2, 1).shape) == (2, 12)
    assert tuple(u.flatten_last_dims(x).shape) == (2, 12) and tuple(u.flatten_last_dims(x, 3).shape) == (24,)
    assert tuple(u.split_dim(x, [2, 2], 2).shape) == (2, 3, 2, 2)
    assert torch.equal(u.split_dim(u.flatten_dims(x, 2, 1), [3, 4], 1), x)
    assert float(u.db(100.)) == 20.0 and float(u.log2(8.)) == 3.0 and float(u.log10(1000.)) == 3.0
    with pytest.raises(AssertionError):
        u.flatten_dims(x, 1, 0)


def test_library_never_calls_getenv_and_option_registry_works():
    """VERDICT r3 / SURVEY 8(b): no hidden global state on the C-ABI's compute paths.  The library copies SAMD_* from the
    environment once at load; the source holds no getenv call at all, and the registry entry works without a GPU."""
    import glob
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    srcs = glob.glob(os.path.join(ROOT, "sionna_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "sionna_amd", "csrc", "*.inc")) + \
        glob.glob(os.path.join(ROOT, "sionna_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "sionna_amd", "csrc", "*.cpp"))
    assert len(srcs) > 20
    for path in srcs:
        code = "\n".join(ln.split("//")[0] for ln in open(path).read().splitlines())
        assert "getenv" not in code, path
        if not path.endswith("common.h"):
            assert "hipFuncSetAttribute" not in code, f"{path}: use SAMD_SET_MAX_LDS (once per kernel and device)"
    g = _ffi.options_generation()
    _ffi.set_option("SAMD_TEST_KEY", 7)
    _ffi.set_option("SAMD_TEST_KEY", None)
    assert _ffi.options_generation() == g + 2
    import pytest
    with pytest.raises(ValueError):
        _ffi.set_option("HOME", "x")


def test_integration_stubs_match_the_abi():
    """every ``lib.samd_*(...)`` call written out in INTEGRATION.md names an exported entry point and passes as many arguments as
    include/sionna_amd.h declares (stubs that splat a prepared argument list with ``*`` are checked by name only)"""
    import os
    import re
    with open(os.path.join(os.path.dirname(__file__), "..", "INTEGRATION.md")) as f:
        src = f.read()
    seen = 0
    for m in re.finditer(r"lib\.(samd_\w+)\(", src):
        name, i, depth = m.group(1), m.end(), 1
        j = i
        while depth and j < len(src):
            depth += (src[j] == "(") - (src[j] == ")")
            j += 1
        parts, d, cur = [], 0, ""
        for c in src[i:j - 1]:
            d += (c in "([") - (c in ")]")
            if c == "," and d == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += c
        if cur.strip():
            parts.append(cur)
        assert name in _ffi._SIGNATURES, f"INTEGRATION.md calls {name}, which include/sionna_amd.h does not declare"
        if not any(p.strip().startswith("*") for p in parts):
            assert len(parts) == len(_ffi._SIGNATURES[name][1]), f"INTEGRATION.md: {name} is written with {len(parts)} arguments, the ABI has {len(_ffi._SIGNATURES[name][1])}"
        seen += 1
    assert seen >= 15


def test_readme_states_the_number_of_entry_points():
    import os
    import re
    with open(os.path.join(os.path.dirname(__file__), "..", "README.md")) as f:
        m = re.search(r"the C-ABI \((\d+) entry points\)", f.read())
    assert m and int(m.group(1)) == len(_ffi.declared_symbols())


def test_option_context_restores_the_previous_value():
    """`with _ffi.option(...)` used to DELETE the key on exit - a nested or pre-set switch was lost."""
    _ffi.set_option("SAMD_TEST_NEST", "outer")
    try:
        with _ffi.option("SAMD_TEST_NEST", "inner"):
            assert _ffi.get_option("SAMD_TEST_NEST") == "inner"
            with _ffi.option("SAMD_TEST_NEST", 3):
                assert _ffi.get_option("SAMD_TEST_NEST") == "3"
            assert _ffi.get_option("SAMD_TEST_NEST") == "inner"
        assert _ffi.get_option("SAMD_TEST_NEST") == "outer"
    finally:
        _ffi.set_option("SAMD_TEST_NEST", None)
    assert _ffi.get_option("SAMD_TEST_NEST") is None
    with _ffi.option("SAMD_TEST_NEST"):
        assert _ffi.get_option("SAMD_TEST_NEST") == "1"
    assert _ffi.get_option("SAMD_TEST_NEST") is None


def test_generated_kernel_code_objects_are_cached_on_disk(tmp_path):
    """the hipRTC compile (~3 s per code, rule and output form, in every process) happens once per
    machine - <dir>/<sha256(source | hipRTC version | target)>.co; a second PROCESS reads the code object instead."""
    import hashlib
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prog = ("import sys, time, ctypes as C; sys.path.insert(0, %r)\n"
            "from tools import jit_dump\nfrom sionna_amd import _ffi\n"
            "h = jit_dump.host_only_handle(768, 1536, 2, None)\n"
            "t = time.time(); code = jit_dump.jit_code(h, 1, 'minsum'); dt = time.time() - t\n"
            "src = jit_dump.jit_source(h, 1, 1, 'minsum')\n"
            "st = (C.c_long * 2)(); _ffi.lib().samd_ldpc5g_jit_cache_stats(st)\n"
            "import hashlib; print(st[0], st[1], len(code), hashlib.sha256(code).hexdigest(), '%%.3f' %% dt)\n"
            "open(sys.argv[1], 'w').write(src)\n" % ROOT)
    env = dict(os.environ, XDG_CACHE_HOME=str(tmp_path / "xdg"))
    env.pop("SAMD_JIT_CACHE_DIR", None)
    runs = [subprocess.check_output([sys.executable, "-c", prog, str(tmp_path / f"src{i}.txt")], env=env).decode().split() for i in range(2)]
    assert runs[0][:2] == ["1", "0"] and runs[1][:2] == ["0", "1"], runs          # compiled once, then read from disk
    assert runs[0][2:4] == runs[1][2:4]                                           # the same code object
    assert float(runs[1][4]) < 0.5, runs                                          # start-to-kernel without the compiler
    files = os.listdir(tmp_path / "xdg" / "sionna_amd")
    assert len(files) == 1 and files[0].endswith(".co")
    # the file name is the SHA-256 of source | hipRTC version | target (the library's own implementation against hashlib)
    import ctypes as C
    rtc = C.CDLL("libhiprtc.so")
    ma, mi = C.c_int(), C.c_int()
    assert rtc.hiprtcVersion(C.byref(ma), C.byref(mi)) == 0
    src = open(tmp_path / "src0.txt").read()
    want = hashlib.sha256((src + f"|hiprtc {ma.value}.{mi.value}|gfx950").encode()).hexdigest()
    assert files[0] == want + ".co"
    # SAMD_JIT_CACHE=0 switches the cache off
    env2 = dict(env, SAMD_JIT_CACHE="0", XDG_CACHE_HOME=str(tmp_path / "xdg2"))
    r = subprocess.check_output([sys.executable, "-c", prog, str(tmp_path / "src2.txt")], env=env2).decode().split()
    assert r[:2] == ["1", "0"] and not os.path.exists(tmp_path / "xdg2" / "sionna_amd")


def test_comm_builds_without_the_rccl_development_header():
    """csrc/comm.cpp binds librccl with dlopen; without <rccl/rccl.h> it states the few declarations it uses."""
    import os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(ROOT, "sionna_amd", "csrc", "comm.cpp")).read()
    assert "__has_include(<rccl/rccl.h>)" in src and "typedef struct ncclComm* ncclComm_t;" in src
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if os.path.exists(hdr):                                   # the stand-in declarations agree with the real header
        real = open(hdr).read()
        import re
        assert re.search(r"ncclSum\s*=\s*0", real) and re.search(r"ncclInt64\s*=\s*4", real) and "#define NCCL_UNIQUE_ID_BYTES 128" in real
