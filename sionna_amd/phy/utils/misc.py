"""SNR helpers and the Monte-Carlo driver - mirror of reference
src/sionna/phy/utils/misc.py (``ebnodb2no`` :171-251, ``hard_decisions`` :254-271,
``complex_normal`` :19-54, ``sim_ber`` :329-865).

Multi-GPU: the reference runs ``mc_fun`` under ``tf.distribute.MirroredStrategy`` and
gathers the full ``b`` / ``b_hat`` tensors to one device (misc.py:541-548).  Here every
rank (one process per GPU, ``torch.distributed``, backend nccl = RCCL over xGMI) runs
``mc_fun`` on its own Philox stream, counts its own errors on the device
(``samd_count_errors_f32``) and the ranks exchange ONE all-reduce of four int64 counters
per Monte-Carlo iteration; all ranks therefore take identical stop decisions.
"""
import time

import numpy as np
import torch

from ... import _ffi
from ..config import config, dtypes
from .metrics import count_errors_into


def ebnodb2no(ebno_db, num_bits_per_symbol, coderate, resource_grid=None, precision=None):
    """No for a given Eb/No in dB, Es = 1 (misc.py:171-251); float32 arithmetic for
    precision="single" like the reference."""
    p = config.precision if precision is None else precision
    f = dtypes[p]["np"]["rdtype"]
    if isinstance(ebno_db, torch.Tensor):
        ebno_db = ebno_db.detach().cpu().numpy()
    ebno = np.power(f(10), np.asarray(ebno_db, dtype=f) / f(10))
    energy_per_symbol = 1.
    if resource_grid is not None:
        energy_per_symbol /= resource_grid.num_streams_per_tx
        cp_overhead = resource_grid.cyclic_prefix_length / resource_grid.fft_size
        num_syms = (resource_grid.num_ofdm_symbols * (1 + cp_overhead)
                    * resource_grid.num_effective_subcarriers)
        energy_per_symbol *= num_syms / resource_grid.num_data_symbols
    no = f(1) / (ebno * f(coderate) * f(num_bits_per_symbol) / f(energy_per_symbol))
    return f(no) if np.ndim(no) == 0 else no.astype(f)


def hard_decisions(llr):
    """Positive -> 1, non-positive -> 0 (misc.py:254-271)."""
    if isinstance(llr, torch.Tensor):
        return (llr > 0).to(llr.dtype)
    llr = np.asarray(llr)
    return (llr > 0).astype(llr.dtype)


def complex_normal(shape, var=1.0, precision=None):
    """CN(0, var) samples on the device Philox stream (misc.py:19-54)."""
    shape = tuple(int(s) for s in shape)
    if (config.precision if precision is None else precision) == "double":
        # the float32 stream's uniforms, Box-Muller evaluated in double (oracle/f64_ofdm.py::complex_normal)
        x = torch.empty(shape, dtype=torch.complex128, device=_ffi.device())
        no = torch.tensor([float(var)], dtype=torch.float64, device=x.device)
        rng = config.rng
        _ffi.check(_ffi.lib().samd_awgn_c128(None, _ffi.ptr(no), 1, rng.seed, rng.next_call(), x.numel(), _ffi.ptr(x),
                                             _ffi.stream()), "complex_normal")
        return x
    x = torch.zeros(shape, dtype=torch.complex64, device=_ffi.device())
    no = torch.tensor([float(var)], dtype=torch.float32, device=x.device)
    rng = config.rng
    _ffi.check(_ffi.lib().samd_awgn_c64(_ffi.ptr(x), _ffi.ptr(no), 1, rng.seed, rng.next_call(), x.numel(),
                                        _ffi.ptr(x), _ffi.stream()), "complex_normal")
    return x


def get_throughput(batch_size, ebno_db, model, repetitions=1):
    """Information-bit throughput of ``model(batch_size, ebno_db) -> (u, u_hat)`` in bit/s -
    twin of the notebooks' helper (tutorials/phy/5G_Channel_Coding_Polar_vs_LDPC_Codes.ipynb)."""
    u, _ = model(batch_size, ebno_db)            # warm-up / build
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repetitions):
        u, _ = model(batch_size, ebno_db)
    torch.cuda.synchronize()
    return u.numel() * repetitions / (time.perf_counter() - t0)


# ---------------------------------------------------------------------------------- sim_ber
def _dist_world(distribute):
    """(enabled, world_size) for the requested ``distribute`` mode.

    The reference fans out from ONE process over every visible GPU (misc.py:616-655).  Here a rank is a process, so
    ``distribute`` means "all ranks of the initialised process group".  Asked for in a process that was NOT launched as
    a rank while several GPUs are visible, it raises instead of silently simulating on one GPU; with at most one GPU
    visible it is off, like the reference with a single logical device (misc.py:620-622)."""
    import torch.distributed as dist
    if distribute is None:
        return False, 1
    if not (distribute == "all" or distribute is True or isinstance(distribute, (tuple, list))):
        raise ValueError("Unknown value for distribute.")
    if dist.is_available() and dist.is_initialized():
        return True, dist.get_world_size()
    n_vis = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if isinstance(distribute, (tuple, list)):
        n_vis = len([i for i in distribute if i < n_vis])
    if n_vis > 1:
        raise RuntimeError(
            f"sim_ber(distribute={distribute!r}): {n_vis} GPUs are visible but this process is not a rank of a "
            "torch.distributed process group, and a rank is one process per GPU here. Start the script with "
            f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node={n_vis} --master-addr 127.0.0.1 script.py` "
            "(+ sionna_amd.phy.utils.init_distributed() at its top), or call "
            "sionna_amd.phy.utils.spawn_sim_ber(make_mc_fun, ...) which starts the ranks itself.")
    return False, 1


def init_distributed(backend=None):
    """Join the process group described by the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*):
    binds the rank to ``cuda:LOCAL_RANK`` and initialises RCCL (backend "nccl"), or gloo when no GPU is visible.
    Returns ``(rank, world_size)``; a process that was not launched as a rank gets ``(0, 1)`` and nothing happens.  A
    launcher that started exactly ONE rank (``torch.distributed.run --nproc-per-node 1``, ``spawn_sim_ber(nprocs=1)``)
    still gets a real one-member group: the same RCCL initialisation, device binding and int64 all-reduce as on 8 GPUs."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ
    if world <= 1 and not launched:
        return 0, 1
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    have_gpu = torch.cuda.is_available() and torch.cuda.device_count() > 0
    if backend is None:
        backend = "nccl" if have_gpu else "gloo"
    if have_gpu:
        local_rank %= torch.cuda.device_count()
        torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def _spawn_rank(rank, world, port, backend, make_mc_fun, make_args, sim_kwargs, queue):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    try:
        init_distributed(backend)
        mc_fun = make_mc_fun(*make_args)
        ber, bler = sim_ber(mc_fun, distribute="all", **sim_kwargs)
        if rank == 0:
            queue.put((np.asarray(ber.cpu()), np.asarray(bler.cpu())))
    finally:
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()


def spawn_sim_ber(make_mc_fun, ebno_dbs, batch_size, max_mc_iter, *, make_args=(), nprocs=None, backend=None, **kwargs):
    """``sim_ber(..., distribute="all")`` from a process that is not a rank (a notebook): starts one process per
    visible GPU (``nprocs``; RCCL, or gloo without GPUs), each builds its own model with the importable, module-level
    ``make_mc_fun(*make_args)`` (device handles cannot be pickled across processes, a factory can), runs ``sim_ber`` on
    its own Philox stream, and rank 0's ``(ber, bler)`` - the all-reduced, global figures - are returned.  The
    single-process fan-out of the reference (misc.py:616-655, ``strategy.run`` :541-548) becomes this."""
    import socket
    import torch.multiprocessing as mp
    if nprocs is None:
        nprocs = torch.cuda.device_count() if torch.cuda.is_available() else 1
    nprocs = max(int(nprocs), 1)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    sim_kwargs = dict(ebno_dbs=np.asarray(ebno_dbs), batch_size=batch_size, max_mc_iter=max_mc_iter, **kwargs)
    procs = [ctx.Process(target=_spawn_rank, args=(r, nprocs, port, backend, make_mc_fun, tuple(make_args), sim_kwargs, queue))
             for r in range(nprocs)]
    for p in procs:
        p.start()
    res = None
    try:
        while res is None:
            try:
                res = queue.get(timeout=1.0)
            except Exception:  # pylint: disable=broad-except  (queue.Empty: keep waiting while the ranks live)
                if any(p.exitcode not in (None, 0) for p in procs):
                    raise RuntimeError("spawn_sim_ber: a rank exited with an error (see its traceback above)") from None
                if all(p.exitcode is not None for p in procs):
                    raise RuntimeError("spawn_sim_ber: the ranks exited without a result") from None
    finally:
        for p in procs:
            p.join(timeout=120 if res is not None else 5)
            if p.is_alive():
                p.terminate()
    from ..block import wrap
    rdtype = dtypes[kwargs.get("precision") or config.precision]["torch"]["rdtype"]
    return wrap(torch.from_numpy(res[0]).to(rdtype)), wrap(torch.from_numpy(res[1]).to(rdtype))


def _all_reduce_counters(vec):
    """SUM all-reduce of the int64 counter vector across ranks (RCCL on GPU, gloo on CPU)."""
    import torch.distributed as dist
    dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return vec


def sim_ber(mc_fun, ebno_dbs, batch_size, max_mc_iter, soft_estimates=False, num_target_bit_errors=None,
            num_target_block_errors=None, target_ber=None, target_bler=None, early_stop=True,
            graph_mode=None, distribute=None, verbose=True, forward_keyboard_interrupt=True,
            callback=None, precision=None):
    # pylint: disable=line-too-long
    """Monte-Carlo BER/BLER simulation with the reference's signature, stopping rules,
    status codes and progress table (misc.py:329-865).

    ``graph_mode`` is accepted for compatibility ("graph"/"xla" have no meaning here: the
    blocks launch pre-compiled HIP kernels).  ``distribute="all"`` uses every rank of the
    initialised ``torch.distributed`` process group: ``max_mc_iter`` is divided by the number
    of ranks (misc.py:651-655) and the error counters are all-reduced.
    Returns ``(ber, bler)`` as host tensors.
    """
    if precision is None:
        precision = config.precision
    rdtype = dtypes[precision]["torch"]["rdtype"]

    STATUS_NA, STATUS_MAX_IT, STATUS_NO_ERR, STATUS_TARGET_BIT = 0, 1, 2, 3
    STATUS_TARGET_BLOCK, STATUS_TARGET_BER, STATUS_TARGET_BLER, STATUS_CB_STOP = 4, 5, 6, 7
    status_levels = {
        STATUS_NA: "not simulated", STATUS_MAX_IT: "reached max iterations",
        STATUS_NO_ERR: "no errors - early stop", STATUS_TARGET_BIT: "reached target bit errors",
        STATUS_TARGET_BLOCK: "reached target block errors",
        STATUS_TARGET_BER: "reached target BER - early stop",
        STATUS_TARGET_BLER: "reached target BLER - early stop",
        STATUS_CB_STOP: "callback triggered stopping"}
    header_text = ["EbNo [dB]", "BER", "BLER", "bit errors", "num bits", "block errors", "num blocks",
                   "runtime [s]", "status"]
    row_fmt = "{: >9} |{: >11} |{: >11} |{: >12} |{: >12} |{: >13} |{: >12} |{: >12} |{: >10}"

    if not isinstance(early_stop, bool):
        raise TypeError("early_stop must be bool.")
    if not isinstance(soft_estimates, bool):
        raise TypeError("soft_estimates must be bool.")
    if not isinstance(verbose, bool):
        raise TypeError("verbose must be bool.")
    if target_ber is not None:
        if not early_stop:
            print("Warning: early stop is deactivated. target_ber is ignored.")
    else:
        target_ber = -1.
    if target_bler is not None:
        if not early_stop:
            print("Warning: early stop is deactivated. target_bler is ignored.")
    else:
        target_bler = -1.
    if graph_mode is None:
        graph_mode = "default"
    if not isinstance(graph_mode, str):
        raise TypeError("graph_mode must be str.")
    if graph_mode not in ("default", "graph", "xla"):
        raise TypeError("Unknown graph_mode selected.")

    run_multi, num_replicas = _dist_world(distribute)
    rank0 = True
    if run_multi:
        import torch.distributed as dist
        rank0 = dist.get_rank() == 0
        max_mc_iter = int(np.ceil(max_mc_iter / num_replicas))
        if rank0:
            print(f"Distributing simulation across {num_replicas} devices.")
            print(f"Reducing max_mc_iter to {max_mc_iter}")
    # every rank must take the same sync/all-reduce decisions -> derive them from the
    # user's flag, print on rank 0 only
    user_verbose = verbose
    verbose = verbose and rank0

    if isinstance(ebno_dbs, torch.Tensor):
        ebno_dbs = ebno_dbs.detach().cpu().numpy()
    ebno_dbs = np.atleast_1d(np.asarray(ebno_dbs, dtype=dtypes[precision]["np"]["rdtype"]))
    batch_size = int(batch_size)
    num_points = len(ebno_dbs)
    # int64 statistics (misc.py:661-674)
    bit_errors = np.zeros(num_points, np.int64)
    block_errors = np.zeros(num_points, np.int64)
    nb_bits = np.zeros(num_points, np.int64)
    nb_blocks = np.zeros(num_points, np.int64)
    status = np.zeros(num_points)
    runtime = np.zeros(num_points)

    def _print_progress(is_final, rt, idx_snr, idx_it, header=None):
        end_str = "\n" if is_final else "\r"
        if header is not None:
            row_text, end_str = header, "\n"
        else:
            with np.errstate(divide="ignore", invalid="ignore"):
                ber_np = np.nan_to_num(np.float64(bit_errors[idx_snr]) / np.float64(nb_bits[idx_snr]))
                bler_np = np.nan_to_num(np.float64(block_errors[idx_snr]) / np.float64(nb_blocks[idx_snr]))
            if status[idx_snr] == STATUS_NA:
                status_txt = f"iter: {idx_it:.0f}/{max_mc_iter:.0f}"
            else:
                status_txt = status_levels[int(status[idx_snr])]
            row_text = [str(np.round(ebno_dbs[idx_snr], 3)), f"{ber_np:.4e}", f"{bler_np:.4e}",
                        np.round(bit_errors[idx_snr], 0), np.round(nb_bits[idx_snr], 0),
                        np.round(block_errors[idx_snr], 0), np.round(nb_blocks[idx_snr], 0),
                        np.round(rt, 1), status_txt]
        print(row_fmt.format(*row_text), end=end_str)

    # The device counters must be read on the host (+ all-reduced) EVERY iteration only when a stopping rule or a callback
    # needs them.  The progress line alone (verbose, the default) gets them after the first and then after every 8th
    # iteration - a deterministic schedule, so that all ranks take the same decision - instead of a device-to-host
    # synchronisation and an all-reduce per Monte-Carlo iteration.
    need_sync = (num_target_bit_errors is not None or num_target_block_errors is not None or callback is not None)
    progress_sync = user_verbose
    cb_state = sim_ber.CALLBACK_CONTINUE
    i = 0
    try:
        for i in range(num_points):
            runtime[i] = time.perf_counter()
            iter_count = -1
            acc = None                 # device (or host) accumulator [bit_err, block_err, bits, blocks]
            synced = np.zeros(4, np.int64)

            def _flush():
                """bring the accumulated counters of this SNR point to the host (+ all-reduce)"""
                nonlocal acc
                if acc is None:
                    return
                vec = acc
                if run_multi:
                    vec = _all_reduce_counters(vec.clone())
                vals = vec.detach().cpu().numpy().astype(np.int64)
                acc = None
                synced[:] += vals
                bit_errors[i], block_errors[i], nb_bits[i], nb_blocks[i] = synced

            for ii in range(max_mc_iter):
                iter_count += 1
                outputs = mc_fun(batch_size=batch_size, ebno_db=ebno_dbs[i])
                b, b_hat = outputs[0], outputs[1]
                if not isinstance(b, torch.Tensor):
                    b = torch.from_numpy(np.ascontiguousarray(np.asarray(b)))
                if not isinstance(b_hat, torch.Tensor):
                    b_hat = torch.from_numpy(np.ascontiguousarray(np.asarray(b_hat)))
                if acc is None:
                    acc = torch.zeros(4, dtype=torch.int64, device=b.device)
                count_errors_into(b, b_hat, acc[:2], soft=soft_estimates)       # misc.py:713-718
                bit_n = b.numel()
                block_n = bit_n // b.shape[-1] if b.dim() > 0 and b.shape[-1] > 0 else bit_n
                acc[2:] += torch.tensor([bit_n, block_n], dtype=torch.int64).to(acc.device, non_blocking=True)
                if need_sync or (progress_sync and (ii == 0 or (ii + 1) % 8 == 0)):
                    _flush()

                cb_state = sim_ber.CALLBACK_CONTINUE
                if callback is not None:
                    cb_state = callback(ii, i, ebno_dbs, bit_errors, block_errors, nb_bits, nb_blocks)
                    if cb_state in (sim_ber.CALLBACK_STOP, sim_ber.CALLBACK_NEXT_SNR):
                        runtime[i] = time.perf_counter() - runtime[i]
                        status[i] = STATUS_CB_STOP
                        break

                if verbose:
                    if i == 0 and iter_count == 0:
                        _print_progress(True, 0, 0, 0, header=header_text)
                        print("-" * 135)
                    _print_progress(False, time.perf_counter() - runtime[i], i, ii)

                if num_target_bit_errors is not None and bit_errors[i] >= num_target_bit_errors:
                    status[i] = STATUS_TARGET_BIT
                    runtime[i] = time.perf_counter() - runtime[i]
                    break
                if num_target_block_errors is not None and block_errors[i] >= num_target_block_errors:
                    runtime[i] = time.perf_counter() - runtime[i]
                    status[i] = STATUS_TARGET_BLOCK
                    break
                if iter_count == max_mc_iter - 1:
                    _flush()
                    runtime[i] = time.perf_counter() - runtime[i]
                    status[i] = STATUS_MAX_IT
            _flush()

            if verbose:
                _print_progress(True, runtime[i], i, iter_count)

            if early_stop:
                if block_errors[i] == 0:
                    status[i] = STATUS_NO_ERR
                    if verbose:
                        print(f"\nSimulation stopped as no error occurred @ EbNo = {ebno_dbs[i]:.1f} dB.\n")
                    break
                ber_true = bit_errors[i] / nb_bits[i]
                bler_true = block_errors[i] / nb_blocks[i]
                if ber_true < target_ber:
                    status[i] = STATUS_TARGET_BER
                    if verbose:
                        print(f"\nSimulation stopped as target BER is reached@ EbNo = {ebno_dbs[i]:.1f} dB.\n")
                    break
                if bler_true < target_bler:
                    status[i] = STATUS_TARGET_BLER
                    if verbose:
                        print(f"\nSimulation stopped as target BLER is reached @ EbNo = {ebno_dbs[i]:.1f} dB.\n")
                    break

            if cb_state is sim_ber.CALLBACK_STOP:
                status[i] = STATUS_CB_STOP
                if verbose:
                    print(f"\nSimulation stopped by callback function @ EbNo = {ebno_dbs[i]:.1f} dB.\n")
                break

    except KeyboardInterrupt as e:
        if forward_keyboard_interrupt:
            raise e
        print(f"\nSimulation stopped by the user @ EbNo = {ebno_dbs[i]} dB.")
        for idx in range(i + 1, num_points):          # misc.py:841-846
            bit_errors[idx] += -1
            block_errors[idx] += -1
            nb_bits[idx] += 1
            nb_blocks[idx] += 1

    with np.errstate(divide="ignore", invalid="ignore"):
        ber = bit_errors.astype(np.float64) / nb_bits.astype(np.float64)
        bler = block_errors.astype(np.float64) / nb_blocks.astype(np.float64)
    ber = np.where(np.isnan(ber), 0., ber)
    bler = np.where(np.isnan(bler), 0., bler)
    from ..block import wrap
    return wrap(torch.from_numpy(ber).to(rdtype)), wrap(torch.from_numpy(bler).to(rdtype))


sim_ber.CALLBACK_CONTINUE = None
sim_ber.CALLBACK_STOP = 2
sim_ber.CALLBACK_NEXT_SNR = 1
