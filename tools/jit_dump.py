"""Source / code object / ISA statistics of the specialised 5G LDPC decoder (csrc/ldpc5g_jit.cpp) for one code.

Runs WITHOUT a GPU: the handle is built under SAMD_HOST_ONLY (tables and schedules only), the source generator and
hipRTC are host code.  `python tools/jit_dump.py --k 2816 --n 8448 --m 6 --bg bg1 --out /tmp/jit` writes
<out>/src.hip, <out>/emu_src.h (without the gfx950 operations: what tests/jit_emu compiles), <out>/code.co and
<out>/code.s, and prints register / size / instruction statistics of the compiled kernel.
"""
import argparse
import collections
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def host_only_handle(k, n, m, bg, return_obj=False):
    """samd_ldpc5g_t built without a device for the code LDPC5GDecoder(LDPC5GEncoder(k, n, m, bg)) would use."""
    from sionna_amd import _ffi
    import sionna_amd.phy as phy
    lib = _ffi.lib()
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=bg)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum")
    _ffi.set_option("SAMD_HOST_ONLY", "1")
    try:
        h = C.c_void_p()
        _ffi.check(lib.samd_ldpc5g_create(
            1 if enc._bg == "bg1" else 2, enc._z, enc._bg_rows.ctypes.data_as(C.c_void_p),
            enc._bg_cols.ctypes.data_as(C.c_void_p), enc._bg_shifts.ctypes.data_as(C.c_void_p), len(enc._bg_rows),
            enc._k, enc._n, 0 if m is None else int(m), int(dec._nb_pruned_nodes), C.byref(h)), "samd_ldpc5g_create")
    finally:
        _ffi.set_option("SAMD_HOST_ONLY", None)
    return (h, enc, dec) if return_obj else h


def jit_source(h, return_infobits, with_ops):
    from sionna_amd import _ffi
    lib = _ffi.lib()
    n = lib.samd_ldpc5g_jit_source(h, int(return_infobits), int(with_ops), None, 0)
    if n < 0:
        raise NotImplementedError(lib.samd_last_error().decode())
    buf = C.create_string_buffer(n + 1)
    lib.samd_ldpc5g_jit_source(h, int(return_infobits), int(with_ops), buf, n + 1)
    return buf.value.decode()


def jit_code(h, return_infobits):
    from sionna_amd import _ffi
    lib = _ffi.lib()
    n = lib.samd_ldpc5g_jit_code(h, int(return_infobits), None, 0)
    if n < 0:
        raise RuntimeError(lib.samd_last_error().decode())
    buf = C.create_string_buffer(n)
    lib.samd_ldpc5g_jit_code(h, int(return_infobits), buf, n)
    return buf.raw


def isa_stats(asm):
    """instruction classes of the kernel body + what the iteration loop (between the outermost s_barrier pair) costs"""
    cls = collections.Counter()
    for line in asm.splitlines():
        mm = re.match(r"\s+([a-z_0-9]+)\s", line)
        if not mm:
            continue
        op = mm.group(1)
        if op.startswith("v_"):
            cls["valu"] += 1
        elif op.startswith("ds_"):
            cls["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cls["vmem"] += 1
            if op.startswith("scratch_"):
                cls["scratch"] += 1
        elif op in ("s_waitcnt", "s_nop", "s_barrier", "s_setprio"):
            cls[op] += 1
        elif op.startswith("s_"):
            cls["salu"] += 1
    return cls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=2816)
    ap.add_argument("--n", type=int, default=8448)
    ap.add_argument("--m", type=int, default=6)
    ap.add_argument("--bg", default="bg1")
    ap.add_argument("--infobits", type=int, default=1)
    ap.add_argument("--out", default="/tmp/jit")
    ap.add_argument("--opt", action="append", default=[], help="SAMD_JIT_*=value (repeatable)")
    a = ap.parse_args()
    from sionna_amd import _ffi
    for kv in a.opt:
        key, val = kv.split("=", 1)
        _ffi.set_option(key, val)
    os.makedirs(a.out, exist_ok=True)
    h = host_only_handle(a.k, a.n, a.m if a.m > 0 else None, a.bg if a.bg != "auto" else None)
    lib = _ffi.lib()
    print("supported:", lib.samd_ldpc5g_jit_supported(h))
    src = jit_source(h, a.infobits, 1)
    open(os.path.join(a.out, "src.hip"), "w").write(src)
    open(os.path.join(a.out, "emu_src.h"), "w").write(jit_source(h, a.infobits, 0))
    print(f"source: {len(src)} bytes, {src.count(chr(10))} lines")
    code = jit_code(h, a.infobits)
    co = os.path.join(a.out, "code.co")
    open(co, "wb").write(code)
    asm = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", co]).decode()
    open(os.path.join(a.out, "code.s"), "w").write(asm)
    notes = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co]).decode()
    for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
                ".group_segment_fixed_size", ".private_segment_fixed_size"):
        mm = re.search(re.escape(key) + r":\s+(\d+)", notes)
        print(f"  {key[1:]:28s} {mm.group(1) if mm else '?'}")
    body = asm[asm.index("<samd_ldpc5g_jit>:"):]
    print(f"  code bytes                   {4 * sum(len(l.split('//')[1].split(':')[1].split()) for l in body.splitlines() if '//' in l and ':' in l.split('//')[1])}")
    st = isa_stats(body)
    print("  instructions:", dict(st))


if __name__ == "__main__":
    main()
