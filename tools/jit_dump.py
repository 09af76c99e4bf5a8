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


def jit_source(h, return_infobits, with_ops, cn="minsum"):
    from sionna_amd import _ffi
    lib = _ffi.lib()
    mode = _ffi.CN_MODES[cn]
    n = lib.samd_ldpc5g_jit_source(h, int(return_infobits), mode, int(with_ops), None, 0)
    if n < 0:
        raise NotImplementedError(lib.samd_last_error().decode())
    buf = C.create_string_buffer(n + 1)
    lib.samd_ldpc5g_jit_source(h, int(return_infobits), mode, int(with_ops), buf, n + 1)
    return buf.value.decode()


def jit_code(h, return_infobits, cn="minsum"):
    from sionna_amd import _ffi
    lib = _ffi.lib()
    mode = _ffi.CN_MODES[cn]
    n = lib.samd_ldpc5g_jit_code(h, int(return_infobits), mode, None, 0)
    if n < 0:
        raise RuntimeError(lib.samd_last_error().decode())
    buf = C.create_string_buffer(n)
    lib.samd_ldpc5g_jit_code(h, int(return_infobits), mode, buf, n)
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


# cycles one wave64 DS instruction occupies the CU's LDS pipeline (MI355X_MICROARCH.md, section LDS: reads = LDS-array
# cycles, stores = the VGPR -> LDS transfer at 2 cycles per source dword incl. the address register)
LDS_CYCLES = {"ds_read_b32": 2, "ds_read_b64": 2, "ds_read2_b32": 4, "ds_read2st64_b32": 4, "ds_read_b128": 4, "ds_read2_b64": 8,
              "ds_write_b32": 4, "ds_write_b64": 6, "ds_write2_b32": 6, "ds_write2st64_b32": 6, "ds_write_b128": 13,
              "ds_write2_b64": 13, "ds_write_addtid_b32": 2, "ds_read_addtid_b32": 2, "ds_read2st64_b64": 8, "ds_write2st64_b64": 13}


def per_iteration_stats(asm):
    """The kernel is 16 wave programs, each [prologue + init | barrier | CN phase | barrier | VN phase | barrier]: instruction
    classes and LDS-pipeline cycles of ONE iteration summed over the waves (what a CU executes per iteration and codeword)."""
    body = asm[re.search(r"<samd_ldpc5g_jit\w*>:", asm).start():]
    segs = [[]]
    for line in body.splitlines():
        mm = re.match(r"\s+([a-z_0-9]+)\s", line)
        if not mm:
            continue
        segs[-1].append(mm.group(1))
        if mm.group(1) == "s_barrier":
            segs.append([])
    out = {}
    for name, sel in (("cn_phase", segs[1::3]), ("vn_phase", segs[2::3]), ("per_codeword", segs[0:-1:3])):
        ops = [o for sg in sel for o in sg]
        unknown = sorted({o for o in ops if o.startswith("ds_") and o not in LDS_CYCLES})
        out[name] = {"valu": sum(o.startswith("v_") for o in ops), "salu": sum(o.startswith("s_") and o not in ("s_waitcnt", "s_nop", "s_barrier", "s_setprio") for o in ops),
                     "lds_insts": sum(o.startswith("ds_") for o in ops), "s_nop": ops.count("s_nop"),
                     "lds_pipe_cycles": sum(LDS_CYCLES.get(o, 0) for o in ops),
                     "lds_mix": dict(collections.Counter(o for o in ops if o.startswith("ds_"))),
                     "valu_per_wave": [sum(o.startswith("v_") for o in sg) for sg in sel]}
        if unknown:
            out[name]["unpriced_ds"] = unknown
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=2816)
    ap.add_argument("--n", type=int, default=8448)
    ap.add_argument("--m", type=int, default=6)
    ap.add_argument("--bg", default="bg1")
    ap.add_argument("--infobits", type=int, default=1)
    ap.add_argument("--cn", default="minsum", choices=["minsum", "offset-minsum", "boxplus-phi"])
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
    src = jit_source(h, a.infobits, 1, a.cn)
    open(os.path.join(a.out, "src.hip"), "w").write(src)
    open(os.path.join(a.out, "emu_src.h"), "w").write(jit_source(h, a.infobits, 0, a.cn))
    print(f"source: {len(src)} bytes, {src.count(chr(10))} lines")
    code = jit_code(h, a.infobits, a.cn)
    co = os.path.join(a.out, "code.co")
    open(co, "wb").write(code)
    asm = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", co]).decode()
    open(os.path.join(a.out, "code.s"), "w").write(asm)
    notes = subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co]).decode()
    for key in (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count",
                ".group_segment_fixed_size", ".private_segment_fixed_size"):
        mm = re.search(re.escape(key) + r":\s+(\d+)", notes)
        print(f"  {key[1:]:28s} {mm.group(1) if mm else '?'}")
    body = asm[re.search(r"<samd_ldpc5g_jit\w*>:", asm).start():]
    print(f"  code bytes                   {4 * sum(len(l.split('//')[1].split(':')[1].split()) for l in body.splitlines() if '//' in l and ':' in l.split('//')[1])}")
    st = isa_stats(body)
    print("  instructions:", dict(st))
    it = per_iteration_stats(asm)
    import json
    json.dump(it, open(os.path.join(a.out, "per_iteration.json"), "w"), indent=1)
    for ph in ("cn_phase", "vn_phase", "per_codeword"):
        d = it[ph]
        print(f"  {ph:13s} valu {d['valu']:5d}  salu {d['salu']:4d}  lds insts {d['lds_insts']:4d}  lds pipe cycles {d['lds_pipe_cycles']:5d}  {d['lds_mix']}")
    tot = it["cn_phase"]["lds_pipe_cycles"] + it["vn_phase"]["lds_pipe_cycles"]
    print(f"  per iteration and CU: {it['cn_phase']['valu'] + it['vn_phase']['valu']} VALU, {tot} LDS-pipeline cycles")


if __name__ == "__main__":
    main()
