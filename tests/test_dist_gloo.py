"""World-size-2 tests of the data-parallel sim_ber path on CPU (gloo backend).

On the GPU node the same code runs with backend nccl (= RCCL over xGMI); the only collective of
the path is the SUM all-reduce of four int64 error counters."""
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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sionna_amd.phy.utils import sim_ber
    from sionna_amd.phy.config import PhiloxGenerator
    from oracle import utils as outil

    gen = PhiloxGenerator(1234)            # rank comes from the environment
    calls = []

    def mc_fun(batch_size, ebno_db):
        # error injector on the rank's own random stream (host tensors): bit error probability
        # 1/16 below 3 dB, 0 above
        calls.append(float(ebno_db))
        n = batch_size * 100
        u = outil.random_bits(gen.seed, gen.next_call(), n).reshape(batch_size, 100)
        flip = outil.random_bits(gen.seed, gen.next_call(), 4 * n).reshape(4, batch_size, 100)
        err = (flip.sum(0) == 4).astype(np.float32) if ebno_db < 3 else np.zeros_like(u)
        return torch.from_numpy(u), torch.from_numpy(np.abs(u - err).astype(np.float32))

    ber, bler = sim_ber(mc_fun, np.array([0.0, 6.0, 20.0]), batch_size=50, max_mc_iter=8, distribute="all",
                        verbose=False, early_stop=False)
    single_calls = len(calls)
    # a rule that needs per-iteration counters: stop after >= 300 bit errors (global)
    calls.clear()
    ber2, _ = sim_ber(mc_fun, np.array([0.0]), batch_size=50, max_mc_iter=100, num_target_bit_errors=300,
                      distribute="all", verbose=False)
    first = outil.random_bits(gen.seed, 0, 64).astype(np.uint8).tolist()        # the rank's first draw
    q.put((rank, ber.numpy().tolist(), bler.numpy().tolist(), single_calls, len(calls), gen.seed,
           ber2.numpy().tolist(), first))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sim_ber_two_ranks_gloo():
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
    (r0, ber0, bler0, n0, m0, seed0, b20, _), (r1, ber1, bler1, n1, m1, seed1, b21, _) = res
    # identical, globally reduced results on both ranks
    assert ber0 == ber1 and bler0 == bler1 and b20 == b21
    # max_mc_iter is divided by the number of replicas (misc.py:651-655): 3 SNR points x 4 iterations
    assert n0 == n1 == 12
    # distinct random streams per rank (test_utils.py:112-127 requirement)
    assert seed0 != seed1 and seed0 == 1234
    # BER ~ 1/16 at 0 dB, exact zero at 6 and 20 dB
    assert abs(ber0[0] - 1 / 16) < 0.01 and ber0[1] == 0.0 and ber0[2] == 0.0
    # early stop on the GLOBAL counter: 300 errors need ~300/(1/16*5000*2) -> 1 iteration per rank
    assert m0 == m1 and m0 <= 2


@pytest.mark.timeout(600)
def test_sim_ber_eight_ranks_gloo():
    """The node's shape - EIGHT ranks (round-4 verdict, next #5): one bootstrap of eight processes, eight distinct Philox
    streams (pairwise different first draws: the reference's own tested requirement, test/unit/utils/test_utils.py:112-127),
    max_mc_iter divided by the number of replicas (misc.py:651-655), every rank returns the same all-reduced figures."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=500) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    assert all(r[1] == res[0][1] and r[2] == res[0][2] and r[6] == res[0][6] for r in res)      # identical global results
    assert all(r[3] == 3 for r in res)                      # 3 SNR points x (8 iterations / 8 replicas)
    seeds = [r[5] for r in res]
    assert len(set(seeds)) == world and seeds[0] == 1234    # rank 0 keeps the user's seed
    firsts = [tuple(r[7]) for r in res]
    assert len(set(firsts)) == world                        # pairwise different first draws
    assert abs(res[0][1][0] - 1 / 16) < 0.01 and res[0][1][1] == 0.0 and res[0][1][2] == 0.0
    assert all(r[4] == res[0][4] and r[4] <= 2 for r in res)   # the global stop rule fires on every rank in the same iteration


def make_injector(p_err_num):
    """module-level factory for spawn_sim_ber: every rank builds its own mc_fun (here an error injector on the rank's
    own Philox stream)"""
    sys.path.insert(0, ROOT)
    from sionna_amd.phy.config import PhiloxGenerator
    from oracle import utils as outil
    gen = PhiloxGenerator(77)

    def mc_fun(batch_size, ebno_db):
        n = batch_size * 64
        u = outil.random_bits(gen.seed, gen.next_call(), n).reshape(batch_size, 64)
        flip = outil.random_bits(gen.seed, gen.next_call(), p_err_num * n).reshape(p_err_num, batch_size, 64)
        err = (flip.sum(0) == p_err_num).astype(np.float32) if ebno_db < 3 else np.zeros_like(u)
        return torch.from_numpy(u), torch.from_numpy(np.abs(u - err).astype(np.float32))
    return mc_fun


@pytest.mark.timeout(300)
def test_spawn_sim_ber_fans_out_from_a_plain_process():
    """sim_ber(distribute="all") from a process that is not a rank: spawn_sim_ber starts the ranks itself (gloo here,
    RCCL on a GPU node) and returns the all-reduced result - the reference's single-process fan-out (misc.py:616-655)."""
    from sionna_amd.phy.utils import spawn_sim_ber
    ber, bler = spawn_sim_ber(make_injector, np.array([0.0, 10.0]), batch_size=100, max_mc_iter=8, make_args=(3,),
                              nprocs=2, backend="gloo", verbose=False, early_stop=False)
    assert abs(float(ber[0]) - 1 / 8) < 0.01 and float(ber[1]) == 0.0 and float(bler[0]) > 0.99


@pytest.mark.timeout(300)
def test_spawn_sim_ber_single_rank_runs_a_real_one_member_group():
    """nprocs=1 (the default on a 1-GPU or CPU-only box): the child is a rank of a ONE-member process group, so the
    all-reduce path executes; the child tears down only what it initialised and exits 0 (round-3 advisor finding: it used
    to raise in destroy_process_group after delivering the result)."""
    from sionna_amd.phy.utils import spawn_sim_ber
    ber1, bler1 = spawn_sim_ber(make_injector, np.array([0.0, 10.0]), batch_size=100, max_mc_iter=8, make_args=(3,),
                                nprocs=1, backend="gloo", verbose=False, early_stop=False)
    # same stream, same answer from a plain in-process run
    from sionna_amd.phy.utils import sim_ber
    ber0, bler0 = sim_ber(make_injector(3), np.array([0.0, 10.0]), batch_size=100, max_mc_iter=8, verbose=False, early_stop=False)
    assert np.array_equal(np.asarray(ber1), np.asarray(ber0)) and np.array_equal(np.asarray(bler1), np.asarray(bler0))


def _failing_factory():
    raise ValueError("model construction failed on purpose")


@pytest.mark.timeout(120)
def test_spawn_sim_ber_reports_a_failing_rank(capfd):
    from sionna_amd.phy.utils import spawn_sim_ber
    with pytest.raises(RuntimeError, match="a rank exited with an error"):
        spawn_sim_ber(_failing_factory, np.array([0.0]), batch_size=10, max_mc_iter=1, nprocs=1, backend="gloo", verbose=False)
    assert "model construction failed on purpose" in capfd.readouterr().err     # the real error is visible, not a teardown one


def test_distribute_all_never_silently_single_gpu(monkeypatch):
    """not a rank + several GPUs visible -> sim_ber(distribute="all") raises; one GPU / none -> off like the reference"""
    from sionna_amd.phy.utils import misc
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    with pytest.raises(RuntimeError, match="8 GPUs are visible"):
        misc._dist_world("all")
    assert misc._dist_world([0]) == (False, 1)            # a single selected device
    with pytest.raises(RuntimeError):
        misc._dist_world([0, 3])
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert misc._dist_world("all") == (False, 1)
    assert misc._dist_world(None) == (False, 1)
    with pytest.raises(ValueError):
        misc._dist_world("some")
