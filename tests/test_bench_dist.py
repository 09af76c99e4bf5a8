"""bench.py's N > 1 logic (step = decode + count + all-reduce of four int64 counters; barrier-bracketed timing; MAX over
ranks of the wall time, SUM of the counters) executed with world size 2 on CPU tensors over gloo.  On the GPU node the
same functions run with backend nccl (= RCCL over xGMI); no 1 -> 8 GPU curve has been measured yet."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, r, lr = bench.world_info()
    B, k = 64, 100
    g = torch.Generator().manual_seed(100 + rank)                      # per-rank data, like the per-rank Philox streams
    u = torch.randint(0, 2, (B, k), generator=g).float()
    flips = (torch.rand((B, k), generator=g) < (0.01 * (rank + 1))).float()

    def count_into(b, b_hat, acc):
        acc[0] += int((b != b_hat).sum())
        acc[1] += int((b != b_hat).any(dim=1).sum())

    counters = torch.zeros(4, dtype=torch.int64)
    step = bench.counted_step(lambda: (u + flips) % 2, u, counters, count_into, w)
    t_wall, c = bench.timed_steps(step, steps=5, warmup=2, world=w, device=torch.device("cpu"), device_sync=lambda: None,
                                  counters=counters)
    local = [int(flips.sum()) * 5, int(flips.any(dim=1).sum()) * 5, B * k * 5, B * 5]
    q.put((rank, w, t_wall, c.tolist(), local))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bench_step_and_reduce_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, w0, t0, c0, l0), (r1, w1, t1, c1, l1) = res
    assert w0 == w1 == 2
    assert t0 == t1 > 0                                  # MAX over ranks: the same number everywhere
    assert c0 == c1 == [a + b for a, b in zip(l0, l1)]   # SUM over ranks of the EXACTLY-K-steps counters (warm-up excluded)
    assert c0[2] == 2 * 64 * 100 * 5 and c0[3] == 2 * 64 * 5


def test_bench_roofline_helpers():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.b_msg(20, 2816) == 13684736
    r = bench.onchip_roofline("no-such-kernel", "x", 10, 1.0)
    assert r["bound"] == "valu" and r["achieved"] is None and "note" in r
    if os.path.exists(bench.COUNTERS):
        rec = bench.load_counters("ldpc5g_ms")
        assert rec is None or {"valu_insts_per_unit", "lds_array_cycles_per_unit", "hbm_bytes_per_unit", "stale"} <= set(rec)
        if rec:
            rf = bench.c2_roofline("minsum", True, 65536, 2816, 20, 30.0)
            assert 0 < rf["frac"] < 1 and rf["unit"] == "G wave64-inst/s" and rf["hbm_resident_equiv"]["algorithmic_bytes_per_decode"] == 13684736


@pytest.mark.timeout(300)
def test_bench_gpus_n_self_launches_its_ranks():
    """`python bench.py --gpus 2` from a plain shell (no WORLD_SIZE) re-executes itself as a 2-rank torch.distributed.run
    launch and rank 0 prints ONE line with n_gpus 2 and the SUM of both ranks' counters (--dry-dist: control path on host
    tensors over gloo; the GPU run of the same path is `SAMD_BENCH_BACKEND=gloo python bench.py --gpus 2` on a 1-GPU box)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-dist", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["parallelism"] == "dp2" and rec["steps"] == 4
    assert rec["counters"][2] == 2 * rec["local_bits_per_rank"]      # counters = 2 x the single-rank ones


@pytest.mark.timeout(600)
def test_bench_gpus_8_self_launch_dry():
    """the driver's 8-GPU shape end to end on the control path: `python bench.py --gpus 8` becomes eight ranks, rank 0
    prints ONE line, the counters are eight times one rank's, all eight ranks are seen by a collective and their random
    streams differ pairwise"""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-dist", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=560, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["config"]["parallelism"] == "dp8"
    assert rec["counters"][2] == 8 * rec["local_bits_per_rank"]
    assert rec["ranks_seen"] == 8 and rec["distinct_random_streams"] is True


def test_bench_self_launch_command_line():
    sys.path.insert(0, ROOT)
    import bench
    argv = bench.self_launch_argv(8, ["--gpus", "8", "--steps", "3"], port=29511)
    assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[-4:] == ["--gpus", "8", "--steps", "3"]
