"""RCCL on the hardware that IS available: a one-member process group on the single leased MI355X (Next 4).

Proves before any 8-GPU run that librccl loads, ``init_distributed`` binds ``cuda:LOCAL_RANK``, and the int64 SUM
all-reduce of the error counters (the path's only collective, reference utils/misc.py:616-655) executes on gfx950 -
through ``spawn_sim_ber`` / ``sim_ber(distribute="all")`` and through ``bench.py`` launched exactly the way the driver
launches it for N > 1 (``python -m torch.distributed.run --nproc-per-node N ...``), with N = 1."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_ldpc_link(k, n, ebno_offset):
    """module-level factory (spawn_sim_ber pickles it by name): QPSK + AWGN + 5G LDPC min-sum BP-10"""
    sys.path.insert(0, ROOT)
    import sionna_amd.phy as phy
    phy.config.seed = 99
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=10)
    src, mapper, demap, chan = phy.mapping.BinarySource(), phy.mapping.Mapper("qam", 2), phy.mapping.Demapper("app", "qam", 2), phy.channel.AWGN()

    def mc_fun(batch_size, ebno_db):
        no = phy.utils.ebnodb2no(ebno_db + ebno_offset, 2, k / n)
        u = src([batch_size, k])
        return u, dec(demap(chan(mapper(enc(u)), no), no))
    return mc_fun


@pytest.mark.timeout(600)
def test_one_member_rccl_group_reduces_the_error_counters():
    import torch
    import torch.distributed as dist
    assert dist.is_nccl_available()
    from sionna_amd.phy.utils import sim_ber, spawn_sim_ber
    ebno = np.array([1.0, 2.0, 3.0])
    kw = dict(batch_size=2000, max_mc_iter=6, verbose=False, early_stop=False, num_target_block_errors=500)
    ber_r, bler_r = spawn_sim_ber(make_ldpc_link, ebno, make_args=(200, 400, 0.0), nprocs=1, backend="nccl", **kw)
    ber_p, bler_p = sim_ber(make_ldpc_link(200, 400, 0.0), ebno, **kw)
    assert np.array_equal(np.asarray(ber_r), np.asarray(ber_p)) and np.array_equal(np.asarray(bler_r), np.asarray(bler_p))
    assert float(bler_p[0]) > float(bler_p[2]) > 0


@pytest.mark.timeout(900)
def test_bench_under_the_drivers_launcher_with_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--batch", "8192", "--also", "none", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["collectives"] == {"backend": "nccl", "group_size": 1}
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["bler"] < 1.0


@pytest.mark.timeout(600)
def test_c_abi_communicator_one_member():
    """samd_comm_* (include/sionna_amd.h "Multi-GPU"): the C host's all-reduce of the int64 counters, here through ctypes
    inside the torch process (the already-mapped librccl is reused): id, create, in-place sum on the current stream,
    teardown.  (A torch-free C process does the same in tests/test_gpu_cabi_c.py.)"""
    import ctypes as C
    import torch
    from sionna_amd import _ffi
    dev = _ffi.device()
    lib = _ffi.lib()
    ident = (C.c_ubyte * 128)()
    _ffi.check(lib.samd_comm_unique_id(ident), "samd_comm_unique_id")
    assert any(ident)
    comm = C.c_void_p()
    _ffi.check(lib.samd_comm_create(ident, 0, 1, C.byref(comm)), "samd_comm_create")
    assert comm.value and lib.samd_comm_rank(comm) == 0 and lib.samd_comm_world_size(comm) == 1
    t = torch.tensor([3, 5, 7, 2 ** 40 + 11], dtype=torch.int64, device=dev)
    for _ in range(2):
        _ffi.check(lib.samd_comm_allreduce_sum_i64(comm, _ffi.ptr(t), 4, _ffi.stream()), "samd_comm_allreduce_sum_i64")
    torch.cuda.synchronize()
    assert t.tolist() == [3, 5, 7, 2 ** 40 + 11]
    lib.samd_comm_destroy(comm)
    with pytest.raises(ValueError):
        _ffi.check(lib.samd_comm_create(ident, 1, 1, C.byref(comm)), "samd_comm_create")
