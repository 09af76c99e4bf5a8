"""Re-entrancy of the C-ABI ("re-entrant per handle + stream; no hidden global state"): two host threads,
each with its own HIP stream and its own handles, decode concurrently - one of them with handles that were CREATED under
a development switch - and both get the single-threaded results.  ctypes releases the GIL around every library call, so
the launches, the option registry and the per-(kernel, device) attribute cache are exercised from two threads at once."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _link(phy, k, n, seed, cn):
    rng = np.random.default_rng(seed)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    u = rng.integers(0, 2, (96, k)).astype(np.float32)
    c = enc(u).numpy()
    y = (1 - 2 * c) + 0.75 * rng.normal(size=c.shape)
    llr = (-2 * y / 0.75 ** 2).astype(np.float32)
    return enc, llr, cn


def test_two_threads_two_streams_two_option_sets():
    import torch
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    jobs = [_link(phy, 1024, 2048, 1, "minsum"), _link(phy, 2816, 8448, 2, "offset-minsum"), _link(phy, 500, 1000, 3, "boxplus-phi")]
    # single-threaded references: default engines and the compressed-state engine
    refs = []
    for enc, llr, cn in jobs:
        refs.append(phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=8, hard_out=False)(llr).numpy())
    # thread B's handles are created under SAMD_ONCHIP_COMPRESSED (they keep it for life), thread A's under the defaults
    with _ffi.option("SAMD_ONCHIP_COMPRESSED"):
        encs_b = [phy.fec.ldpc.LDPC5GEncoder(enc.k, enc.n) for enc, _, _ in jobs]
        decs_b = [phy.fec.ldpc.LDPC5GDecoder(e, cn_update=cn, num_iter=8, hard_out=False) for e, (_, _, cn) in zip(encs_b, jobs)]
        for d, (_, llr, _) in zip(decs_b, jobs):
            d(llr[:2])                                                    # handle creation happens on first use
        gen_b = _ffi.options_generation()
    decs_a = [phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=8, hard_out=False) for enc, _, cn in jobs]
    for d, (_, llr, _) in zip(decs_a, jobs):
        d(llr[:2])
    # freeze the handle caches: the generation moved when the `with` block ended, so pin both sets to "current"
    for e in encs_b + [j[0] for j in jobs]:
        e._handles_gen = _ffi.options_generation()
    assert gen_b != _ffi.options_generation()
    out = {"a": [None] * len(jobs), "b": [None] * len(jobs)}
    errs = []

    def work(tag, decs):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for rep in range(6):
                    for i, (d, (_, llr, _)) in enumerate(zip(decs, jobs)):
                        res = d(llr).numpy()
                        if out[tag][i] is None:
                            out[tag][i] = res
                        elif not np.array_equal(out[tag][i], res):
                            errs.append((tag, i, rep, "not reproducible"))
        except Exception as e:  # pylint: disable=broad-except
            errs.append((tag, repr(e)))

    ta = threading.Thread(target=work, args=("a", decs_a))
    tb = threading.Thread(target=work, args=("b", decs_b))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errs, errs
    for i, (_, _, cn) in enumerate(jobs):
        assert np.array_equal(out["a"][i], refs[i]), (cn, "thread A differs from the single-threaded run")
        if cn != "boxplus-phi":          # min-sum family: every engine is bit-identical to the oracle, hence to each other
            assert np.array_equal(out["b"][i], refs[i]), (cn, "compressed-state engine in thread B differs")
        else:
            assert np.array_equal(out["b"][i], refs[i])                # the switch does not touch the boxplus engines


def test_set_option_rejects_foreign_keys_and_is_visible_to_new_handles():
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    with pytest.raises(ValueError):
        _ffi.set_option("PATH", "x")
    g0 = _ffi.options_generation()
    enc = phy.fec.ldpc.LDPC5GEncoder(1024, 2048)
    h0 = enc._handle(0).value
    assert enc._handle(0).value == h0                                   # cached while nothing changed
    with _ffi.option("SAMD_ENC_BYTES"):
        assert _ffi.options_generation() == g0 + 1
        assert enc._handle(0).value != h0                               # a new handle under the new options
    assert _ffi.options_generation() == g0 + 2
