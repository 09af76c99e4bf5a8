"""The C-ABI is usable from plain C: tests/cabi/cabi_demo.c (C99, includes only include/sionna_amd.h
and the HIP runtime API) is compiled with gcc, linked against sionna_amd/lib/libsionna_amd.so and
run as a separate process - no Python, no torch in that process."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path, name="cabi_demo"):
    exe = os.path.join(tmp_path, name)
    cmd = ["gcc", "-std=gnu99" if name == "comm_demo" else "-std=c99", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "cabi", name + ".c"), "-o", exe, "-L", os.path.join(ROOT, "sionna_amd", "lib"),
           "-lsionna_amd", "-L", "/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "sionna_amd", "lib"),
           "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True)
    return exe


def test_header_is_plain_c_and_client_links(tmp_path):
    """CPU part: the header compiles as C99 and a C client links against the library."""
    subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "sionna_amd.h")], check=True)
    assert os.path.exists(_build(str(tmp_path)))
    assert os.path.exists(_build(str(tmp_path), "comm_demo"))


@pytest.mark.gpu
def test_c_client_round_trip(tmp_path):
    import sionna_amd.phy as phy
    enc = phy.fec.ldpc.LDPC5GEncoder(1024, 2048, bg="bg1")
    code = os.path.join(str(tmp_path), "code.txt")
    with open(code, "w") as f:
        f.write(f"1 {enc.z} {enc.k} {enc.n} {len(enc._bg_rows)}\n")
        for r, c, s in zip(enc._bg_rows, enc._bg_cols, enc._bg_shifts):
            f.write(f"{int(r)} {int(c)} {int(s)}\n")
    exe = _build(str(tmp_path))
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    res = subprocess.run([exe, code], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0 and "CABI_DEMO_OK" in res.stdout, res.stdout + res.stderr


@pytest.mark.gpu
def test_c_client_rccl_allreduce(tmp_path):
    """tests/cabi/comm_demo.c: communicator id, samd_comm_create, three all-reduces of the int64 counters and teardown from
    a torch-free C process - a one-member group on the leased GPU (RCCL loads, binds the device, reduces int64 on gfx950)."""
    exe = _build(str(tmp_path), "comm_demo")
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    res = subprocess.run([exe, "0", "1", os.path.join(str(tmp_path), "comm.id")], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0 and "COMM_DEMO_OK" in res.stdout, res.stdout + res.stderr
