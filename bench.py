#!/usr/bin/env python3
"""Headline benchmark: codeword-decodes/sec, 5G LDPC BG1 n=8448 rate 1/3, BP 20 iterations
(BASELINE.json metric, config C2; C3 = the same sharded over ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--cn-update minsum|boxplus-phi|...]

One step = one pass of the decoder hot path over one resident batch: LDPC5GDecoder.call on a
[B, 8448] float32 LLR tensor that is already in HBM (generated untimed by the library's own
chain: BinarySource -> LDPC5GEncoder -> 64-QAM Mapper -> AWGN -> app Demapper), followed by
the error count against the transmitted bits and, for N > 1 ranks, the RCCL all-reduce of the
four int64 counters (the only collective of the path; ranks are otherwise independent ->
weak scaling, B codewords per GPU).  Rank 0 prints ONE JSON line.

`roofline`     the dominant kernel keeps its messages on chip, so it is priced against the unit it loads, not
               against HBM: bound "valu", achieved = wave64 VALU instructions per second = (SQ_INSTS_VALU per
               decode, from the committed rocprofv3 PMC summary profiles/counters.json, a property of the kernel
               binary) x decodes per launch / the launch duration measured HERE with HIP events on the launch
               stream; peak = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (MI355X_MICROARCH.md).
               `lds_frac` is the same for the LDS array (SQ_LDS_IDX_ACTIVE cycles per decode).  `traffic` = HBM
               bytes per launch from the PMC passes (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction),
               `hbm_resident_equiv` keeps SURVEY 8(d)'s B_msg figure of the HBM-resident formulation.
               `counters.stale` is true when the kernel source changed after the counters were collected.
`also`         the reference's DEFAULT rule (boxplus-phi) on its on-chip engine, same line format.
`cpu_baseline` oracle/ldpc_bp.c (plain-C restatement of the reference algorithm, OpenMP over codewords, all host
               cores) on a bounded sample of the same LLRs, rank 0, N=1 - for BOTH rules.
`extra`        short runs of the secondary workloads (BASELINE configs C4 and C5) with their own roofline and
               cpu_baseline, so that the driver's default invocation times them too (`--no-extra` skips).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K_INFO, N_CW, M_BITS, BG = 2816, 8448, 6, "bg1"
N_VN, N_CN, N_EDGES = 8704, 5888, 40448
HBM_PEAK_GBPS = 8000.0
NUM_SIMD, CLOCK_GHZ, NUM_CU = 1024, 2.4, 256
VALU_PEAK_GINST = NUM_SIMD * CLOCK_GHZ / 2.0          # wave64 VALU instructions per ns -> G inst/s (2 cycles each)
LDS_PIPE_PEAK_GCYC = NUM_CU * CLOCK_GHZ               # LDS-pipeline cycles per ns: one LDS per CU
COUNTERS = os.path.join(ROOT, "profiles", "counters.json")


def b_msg(num_iter, k_out):
    """algorithmic bytes per decode of the HBM-resident formulation."""
    return num_iter * (16 * N_EDGES + 4 * N_VN) + 4 * N_CW + 4 * k_out


# ------------------------------------------------------------------ counters (tools/pmc_counters.py)
def _sha16(paths):
    """hash over a kernel's own source list (counters.json `sources`; older entries name one `source`)"""
    hsh = hashlib.sha256()
    try:
        for path in ([paths] if isinstance(paths, str) else paths):
            with open(os.path.join(ROOT, path), "rb") as f:
                hsh.update(f.read())
        return hsh.hexdigest()[:16]
    except OSError:
        return None


def _load_json(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def load_counters(kernel_key):
    """Per-unit PMC counters of one kernel from profiles/counters.json (None if absent)."""
    try:
        with open(COUNTERS) as f:
            allc = json.load(f)
        rec = allc["kernels"].get(kernel_key)
    except (OSError, ValueError, KeyError):
        return None
    if rec:
        rec = dict(rec)
        srcs = rec.get("sources") or rec.get("source")
        rec["stale"] = bool(srcs and _sha16(srcs) != rec.get("source_sha16"))
        rec["collected_on"] = rec.get("collected_on") or allc.get("collected_on")     # (entries merged later carry their own)
        # static VALU mix x measured issue times (tools/valu_mix.py) and the ablation shares (tools/gpu_ablation.sh)
        rec["mix"] = _load_json("r03_valu_mix.json").get("kernels", {}).get(kernel_key)
        rec["ablation"] = _load_json("r03_ablation.json").get(kernel_key)
    return rec


def onchip_roofline(kernel_key, kernel_name, units, ms, extra=None):
    """VALU-issue / LDS roofline of an on-chip kernel: counters per unit x units / live duration."""
    rec = load_counters(kernel_key)
    out = {"bound": "valu", "achieved": None, "peak": round(VALU_PEAK_GINST, 1), "unit": "G wave64-inst/s",
           "frac": None, "traffic": None, "kernel": kernel_name, "ms_per_launch": round(ms, 3)}
    if rec:
        ginst = rec["valu_insts_per_unit"] * units / (ms * 1e-3) / 1e9
        out.update({
            "achieved": round(ginst, 1), "frac": round(ginst / VALU_PEAK_GINST, 4),
            "lds_frac": round(rec["lds_array_cycles_per_unit"] * units / (NUM_CU * CLOCK_GHZ * 1e9 * ms * 1e-3), 4),
            "salu_per_valu": round(rec["salu_insts_per_unit"] / max(rec["valu_insts_per_unit"], 1), 3),
            "traffic": int(rec["hbm_bytes_per_unit"] * units),
            "counters": {"file": "profiles/counters.json", "from": rec.get("from"), "per_unit": rec.get("unit"),
                         "valu_insts_per_unit": rec["valu_insts_per_unit"], "measured_units_per_launch": rec.get("units_per_launch"),
                         "stale": rec["stale"], "collected_on": rec.get("collected_on")}})
        # WHY the fraction is what it is: shares of a wave's resident cycles (PMC), the barrier share (ablation build
        # without the two workgroup barriers), and the VALU pipe's busy share under the instruction mix's own issue
        # times (no instruction of these kernels issues at the nominal 2 cycles: tools/valu_mix.py)
        for kf in ("wait_frac", "issue_stall_frac", "active_frac"):
            if rec.get(kf) is not None:
                out[kf] = rec[kf]
        if rec.get("ablation"):
            out["barrier_frac"] = rec["ablation"].get("barrier_frac")
            out["ablation"] = {k_: rec["ablation"][k_] for k_ in ("node_arithmetic_frac", "lds_frac_of_time", "lds_plus_barriers_frac",
                                                                  "fixed_per_codeword_frac", "from") if k_ in rec["ablation"]}
        # (the static mix of profiles/r03_valu_mix.json was taken on the round-3 kernels: the boxplus-phi kernels changed their
        # arithmetic in round 5 - a busy share above 1 came out of the old mix - and have no current one)
        if rec.get("mix") and kernel_key not in ("ldpc5g_bp", "ldpc5g_bp_fast", "ldpc5g_jit_phi"):
            ns = rec["mix"]["ns_per_valu_inst_est"]
            out["valu_busy_est"] = round(rec["valu_insts_per_unit"] * units * ns * 1e-9 / (NUM_SIMD * ms * 1e-3), 4)
            out["valu_mix"] = {"ns_per_inst_est": ns, "class_counts": rec["mix"]["class_counts"],
                               "note": "static mix of the kernel x issue times measured by tools/ubench/valu_rate.hip "
                                       "(profiles/r03b/valu_rate_r03b.txt); peak under this mix = 1024 SIMDs / ns_per_inst"}
        if rec.get("lds_pipe_cycles_per_unit"):
            # The kernel generated for the code (csrc/ldpc5g_jit.cpp) is bound by the CU's LDS PIPELINE - every message
            # crosses LDS four times per iteration (read + write in each phase) and a DS instruction occupies the pipeline
            # for the cycles MI355X_MICROARCH.md (section LDS) gives: 2 per 32-lane read pass, 2 per source dword of a store.
            # achieved = those cycles per decode (static: the code object's DS instruction mix, profiles/counters.json) x
            # decodes per launch / the launch time measured here; peak = one LDS cycle per CU and clock.
            gcyc = rec["lds_pipe_cycles_per_unit"] * units / (ms * 1e-3) / 1e9
            out.update({"bound": "lds", "unit": "G LDS-pipeline cycles/s", "achieved": round(gcyc, 1), "peak": round(LDS_PIPE_PEAK_GCYC, 1),
                        "frac": round(gcyc / LDS_PIPE_PEAK_GCYC, 4), "valu_frac": round(ginst / VALU_PEAK_GINST, 4),
                        "valu_achieved_ginst": round(ginst, 1),
                        "lds_pipe": {"cycles_per_decode": rec["lds_pipe_cycles_per_unit"], "static": rec.get("static")},
                        "icache": rec.get("icache")})
    else:
        out["note"] = "profiles/counters.json has no entry for this kernel: run `bash tools/gpu_trip.sh <tag> pmc` (tools/pmc_counters.py)"
    if extra:
        out.update(extra)
    return out


# ------------------------------------------------------------------ timing helpers (shared by all workloads; CPU-testable)
def world_info():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def _grouped(world):
    """Collectives run when there is more than one rank - or when a launcher started this process as the single rank of a
    one-member group (`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the RCCL path end to end on one GPU)."""
    if world > 1:
        return True
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def barrier(world, device_sync):
    if _grouped(world):
        import torch.distributed as dist
        dist.barrier()
    device_sync()


def reduce_sum(t, world):
    if _grouped(world):
        import torch.distributed as dist
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def reduce_max(value, world, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if _grouped(world):
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def ranks_seen(world, rank, device):
    """How many ranks took part in a collective of this run, and whether their Philox streams differ pairwise (the first
    draw of every rank's generator, gathered): what the bench line reports as `rccl_ranks_seen` for N > 1."""
    if not _grouped(world):
        return 1, True
    import torch.distributed as dist
    from sionna_amd.phy.config import PhiloxGenerator
    gen = PhiloxGenerator(20250923, rank=rank)
    mine = torch.tensor([rank, gen.seed & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
    gathered = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, mine)
    ranks = {int(g[0]) for g in gathered}
    seeds = {int(g[1]) for g in gathered}
    return len(ranks), len(seeds) == len(gathered)


def self_launch_argv(n_gpus, argv=None, port=None):
    """Command line that runs this file as N ranks of ONE node (one process per GPU, RCCL over xGMI) - exactly the
    launch the driver uses for N > 1; `python bench.py --gpus N` from a plain shell re-executes itself as this."""
    import socket
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(sys.argv[1:] if argv is None else argv)


def self_launch(n_gpus):
    """`bench.py --gpus N` without WORLD_SIZE: replace this process by the N-rank launch (rank 0 prints the ONE line)."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("SAMD_BENCH_BACKEND", "nccl") == "nccl" and have < n_gpus and "--dry-dist" not in sys.argv:
        print(f"bench.py: --gpus {n_gpus} but only {have} GPU(s) visible (SAMD_BENCH_BACKEND=gloo shares devices "
              "between ranks for a code-path check)", file=sys.stderr)
        sys.exit(2)
    argv = self_launch_argv(n_gpus)
    sys.stdout.flush()
    os.environ["SAMD_BENCH_CHILD"] = "1"    # the ranks must not launch again (e.g. a launcher that gave them WORLD_SIZE=1)
    for k_ in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k_, None)
    os.execv(sys.executable, argv)


def timed_steps(step, steps, warmup, world, device, device_sync, counters=None):
    """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + device synchronisation on both sides;
    returns (max-over-ranks wall seconds, sum-over-ranks counters)."""
    for _ in range(warmup):
        step(None)
    if counters is not None:
        counters.zero_()
    barrier(world, device_sync)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    barrier(world, device_sync)
    t_wall = reduce_max(time.perf_counter() - t0, world, device)
    return t_wall, (reduce_sum(counters, world).cpu().numpy() if counters is not None else None)


def counted_step(decode, u, counters, count_into, world, events=None):
    """The hot-path step: decode, count errors against the transmitted bits, all-reduce the four int64 counters."""
    def step(i):
        ev = events[i] if (events is not None and i is not None) else None
        if ev is not None:
            ev[0].record()              # HIP events on the launch stream bracket the decoder kernels only
        u_hat = decode()
        if ev is not None:
            ev[1].record()
        count_into(u, u_hat, counters[:2])
        counters[2] += u.numel()
        counters[3] += u.shape[0]
        if world > 1:                   # the path's only collective: 4 x int64 error counters (RCCL)
            reduce_sum(counters, world)
    return step


# ------------------------------------------------------------------ C4: OFDM 4x2 LMMSE
def bench_c4(args, short=False):
    """BASELINE config C4: OFDM 14x76, TDL-A 300 ns, 4x2, LS-NN + fused per-RE LMMSE, QPSK + LDPC k=768 n=1536 per
    stream, batch 8192.  One step = LMMSEEqualizer.call on a resident batch; metric resource-elements/s; HBM roofline
    with 120 algorithmic bytes per RE (y 32 + H 64 in, x_hat 16 + no_eff 8 out)."""
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    phy.config.seed = 4
    steps, warmup = (max(3, args.steps // 2), 1) if short else (args.steps, args.warmup)
    B, k, n, m = (args.batch if args.batch != 65536 else 8192), 768, 1536, 2
    ebno = 10.0 if args.ebno_db == 4.5 else args.ebno_db
    rg = phy.ofdm.ResourceGrid(14, 76, 15e3, num_tx=1, num_streams_per_tx=2, cyclic_prefix_length=6,
                               num_guard_carriers=[5, 6], dc_null=True, pilot_pattern="kronecker",
                               pilot_ofdm_symbol_indices=[2, 11])
    sm = phy.mimo.StreamManagement([[1]], 2)
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n)
    dec = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update="minsum", num_iter=20)
    src, mapper, rgm = phy.mapping.BinarySource(), phy.mapping.Mapper("qam", m), phy.ofdm.ResourceGridMapper(rg)
    tdl = phy.channel.tr38901.TDL("A", 300e-9, 2.6e9, min_speed=10., num_rx_ant=4, num_tx_ant=2)
    ch = phy.channel.OFDMChannel(tdl, rg, normalize_channel=True, return_channel=True)
    est, eq = phy.ofdm.LSChannelEstimator(rg), phy.ofdm.LMMSEEqualizer(rg, sm)
    demap = phy.mapping.Demapper("app", "qam", m)
    no = phy.utils.ebnodb2no(ebno, m, k / n, rg)

    def chain():
        b = src([B, 1, 2, k])
        y, h = ch(rgm(mapper(enc(b))), no)
        h_hat, ev = est(y, no)
        x_hat, no_eff = eq(y, h_hat, ev, no)
        return b, dec(demap(x_hat, no_eff)), (y, h_hat, ev)

    b, b_hat, (y, h_hat, ev) = chain()
    torch.cuda.synchronize()
    for _ in range(warmup):
        eq(y, h_hat, ev, no)
    ev_t = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e0, e1 in ev_t:
        e0.record(); eq(y, h_hat, ev, no); e1.record()
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    ms_call = float(np.mean([a.elapsed_time(c) for a, c in ev_t]))     # one Block-API call: host work + launch + kernel
    # kernel time: 20 launches through the C-ABI with prepared arguments between ONE pair of HIP events on the launch
    # stream - the queue stays full (a C-ABI launch costs ~10 us of host time, the kernel ~190 us), so elapsed / 20 is the
    # kernel's own duration (rocprofv3: profiles/*kernel_stats*), which the host-bound per-call figure above hides
    keep, head, tabs, dims = eq._prepare(y, h_hat, ev, no)
    xk = torch.empty((B, rg.num_tx, rg.num_streams_per_tx, rg.num_data_symbols), dtype=torch.complex64, device=y.device)
    nk = torch.empty((B, rg.num_tx, rg.num_streams_per_tx, rg.num_data_symbols), dtype=torch.float32, device=y.device)
    lib, st, reps_k = _ffi.lib(), _ffi.stream(), 20
    def launch():
        return lib.samd_ofdm_lmmse_c64(*head, *tabs, *dims, int(eq._mode), _ffi.ptr(xk), _ffi.ptr(nk), st)
    launch(); launch()
    torch.cuda.synchronize()
    k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k0.record()
    for _ in range(reps_k):
        launch()
    k1.record()
    torch.cuda.synchronize()
    ms = k0.elapsed_time(k1) / reps_k
    del keep
    # the receiver front end LS-NN -> LMMSE -> demapper: three launches through the separate blocks (estimator with
    # defer=False) against ONE launch of the fused kernel that LinearDetector runs on the estimator's deferred h_hat
    est_e = phy.ofdm.LSChannelEstimator(rg, defer=False)
    det = phy.ofdm.LinearDetector("lmmse", "bit", "app", rg, sm, constellation_type="qam", num_bits_per_symbol=m)

    def rx_separate():
        hh, evv = est_e(y, no)
        xx, nn = eq(y, hh, evv, no)
        return demap(xx, nn)

    def rx_fused():
        hh, evv = est(y, no)
        return det(y, hh, evv, no)

    def timed(fn, reps_=10):
        fn(); fn()
        torch.cuda.synchronize()
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        for _ in range(reps_):
            fn()
        a1.record()
        torch.cuda.synchronize()
        return a0.elapsed_time(a1) / reps_
    same_bits = bool(torch.equal(rx_separate(), rx_fused()))
    ms_sep, ms_fused = timed(rx_separate), timed(rx_fused)
    # channel generation: the two kernels behind OFDMChannel, timed on their own
    fs = 1.0 / rg.ofdm_symbol_duration
    freqs = phy.channel.subcarrier_frequencies(rg.fft_size, rg.subcarrier_spacing)
    a_t, tau_t = tdl(B, rg.num_ofdm_symbols, fs)
    ms_tdl = timed(lambda: tdl(B, rg.num_ofdm_symbols, fs), 5)
    ms_c2o = timed(lambda: phy.channel.cir_to_ofdm_channel(freqs, a_t, tau_t, normalize=True), 5)
    c2o_out = B * 4 * 2 * rg.num_ofdm_symbols * rg.fft_size * 8              # h_freq [B, 1, 4, 1, 2, T, fft] complex64
    c2o_in = int(a_t.numel()) * 8 + int(tau_t.numel()) * 4
    rec_c2o, rec_tdl = load_counters("cir_to_ofdm"), load_counters("tdl_cir")
    channel_kernels = {
        "cir_to_ofdm": {"kernel": "cir_to_ofdm_pass_kernel<24, 24, 4> (phase table + taps in LDS, paths in passes of 4 with the next row's taps requested ahead, results staged in registers, one store per value)",
                        "ms_per_launch": round(ms_c2o, 4),
                        "roofline": {"bound": "hbm", "achieved": round((c2o_in + c2o_out) / (ms_c2o * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS,
                                     "unit": "GB/s", "frac": round((c2o_in + c2o_out) / (ms_c2o * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                     "algorithmic_bytes": c2o_in + c2o_out,
                                     "traffic": int(rec_c2o["hbm_bytes_per_unit"] * rec_c2o["units_per_launch"]) if rec_c2o else None,
                                     "write_bytes_over_output": round(rec_c2o["write_size_kb_per_launch"] * 1024 / c2o_out, 3) if rec_c2o else None}},
        "tdl_cir": {"kernel": "tdl_cir_kernel (sum of 20 sinusoids per tap, Philox draws in the kernel)", "ms_per_launch": round(ms_tdl, 4),
                    "roofline": {"bound": "valu", "unit": "G wave64-inst/s", "peak": round(VALU_PEAK_GINST, 1),
                                 "achieved": round(rec_tdl["valu_insts_per_unit"] * rec_tdl["units_per_launch"] / (ms_tdl * 1e-3) / 1e9, 1) if rec_tdl else None,
                                 "frac": round(rec_tdl["valu_insts_per_unit"] * rec_tdl["units_per_launch"] / (ms_tdl * 1e-3) / 1e9 / VALU_PEAK_GINST, 4) if rec_tdl else None,
                                 "traffic": int(rec_tdl["hbm_bytes_per_unit"] * rec_tdl["units_per_launch"]) if rec_tdl else None,
                                 "output_bytes": int(a_t.numel()) * 8}}}
    del a_t, tau_t
    # the channel stage as the chain runs it (round 6): OFDMChannel = tdl_cir + ONE launch for cir_to_ofdm_channel + ApplyOFDMChannel
    # (cir_to_ofdm_fused_kernel: h_freq, 558 MB here, is deferred and never written) + awgn_kernel in place on y
    x_rg = rgm(mapper(enc(src([B, 1, 2, k]))))
    ms_stage = timed(lambda: ch(x_rg, no), 5)
    with _ffi.option("SAMD_NO_FUSED_CHANNEL"):
        ms_stage_sep = timed(lambda: ch(x_rg, no), 5)
    channel_kernels["ofdm_channel_stage"] = {
        "what": "OFDMChannel(TDL-A, normalize_channel, return_channel=True)(x, no): tdl_cir + cir_to_ofdm_fused (h_freq deferred) + awgn",
        "ms_per_call": round(ms_stage, 4), "ms_per_call_separate_kernels": round(ms_stage_sep, 4),
        "note": "separate = cir_to_ofdm_pass + apply_ofdm_channel + awgn with h_freq materialised (SAMD_NO_FUSED_CHANNEL)"}
    del x_rg
    n_data_re = B * rg.num_data_symbols
    ach = n_data_re * 120 / (ms * 1e-3) / 1e9
    chain()                                   # (handles rebuilt after the development switch above: not part of the timing)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 2 if short else 3
    for _ in range(reps):
        chain()
    torch.cuda.synchronize()
    t_e2e = (time.perf_counter() - t0) / reps
    rec = load_counters("ofdm_lmmse")
    out = {"metric": "LMMSE-equalised resource elements/sec (4x2, config C4)", "value": round(n_data_re * steps / t_wall, 1),
           "unit": "resource-elements/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": round(t_wall / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "c64", "data": "synthetic",
           "config": {"workload": f"C4: OFDM 14x76 (64 eff. subcarriers, pilots at symbols 2,11), TDL-A 300 ns, 4 rx x 2 streams, "
                                  f"LS-NN + LMMSE, QPSK, LDPC k=768 n=1536 per stream, batch {B}", "batch": B, "ebno_db": ebno},
           "ber": float((b != b_hat).float().mean()),
           "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBPS, 4),
                        "traffic": int(rec["hbm_bytes_per_unit"] * n_data_re) if rec else None,
                        "kernel": "ofdm_lmmse_diag_kernel<4,2,1> (whole OFDMEqualizer.call in one launch)",
                        "algorithmic_bytes_per_re": 120, "ms_per_launch": round(ms, 4),
                        "ms_per_block_api_call": round(ms_call, 4), "host_overhead_ms": round(ms_call - ms, 4),
                        "note": "ms_per_launch = 20 back-to-back C-ABI launches between one pair of HIP events (queue "
                                "full); the Block-API call adds the host overhead reported beside it"},
           "receiver_front_end": {
               "stages": "LSChannelEstimator(nn) -> LMMSEEqualizer -> Demapper(app)", "ms_three_launches": round(ms_sep, 4),
               "ms_fused_one_launch": round(ms_fused, 4), "bit_identical": same_bits,
               "kernel": "ofdm_lsnn_lmmse_kernel<4,2,1,app,1> (h_hat deferred, never written)",
               "algorithmic_GBps_at_120B_per_RE": round(n_data_re * 120 / (ms_fused * 1e-3) / 1e9, 1),
               "frac_of_hbm_peak_at_120B_per_RE": round(n_data_re * 120 / (ms_fused * 1e-3) / 1e9 / HBM_PEAK_GBPS, 3),
               "note": "SURVEY 8(d)'s 120 B/RE assume h_hat (64 B/RE) is read from HBM; the fused kernel reads y (32 B/RE + "
                       "pilot rows from L2) and writes 16 B/RE of LLRs, so this fraction may exceed what an h_hat-reading "
                       "kernel can reach - it is reported for comparison with the separate-kernel figure above"},
           "channel_kernels": channel_kernels,
           "end_to_end": {"codewords_per_s": round(2 * B / t_e2e, 1), "ms_per_batch": round(t_e2e * 1e3, 2)}}
    if not short:
        # time-domain variant of the same chain: OFDMModulator (rocFFT) -> TimeChannel -> OFDMDemodulator (rocFFT)
        bw = rg.bandwidth
        l_min, l_max = phy.channel.time_lag_discrete_time_channel(bw)
        tch = phy.channel.TimeChannel(tdl, bw, rg.num_time_samples, l_min=l_min, l_max=l_max, normalize_channel=True)
        omod, odem = phy.ofdm.OFDMModulator(rg.cyclic_prefix_length), phy.ofdm.OFDMDemodulator(rg.fft_size, l_min, rg.cyclic_prefix_length)
        est_lin = phy.ofdm.LSChannelEstimator(rg, interpolation_type="lin")
        Bt = min(B, 2048)

        def chain_td():
            bb = src([Bt, 1, 2, k])
            yt = odem(tch(omod(rgm(mapper(enc(bb)))), no))
            hh, evv = est_lin(yt, no)
            xh, ne = eq(yt, hh, evv, no)
            return bb, dec(demap(xh, ne))
        bt, bt_hat = chain_td()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            chain_td()
        torch.cuda.synchronize()
        t_td = (time.perf_counter() - t0) / 3
        out["end_to_end_time_domain"] = {"codewords_per_s": round(2 * Bt / t_td, 1), "ms_per_batch": round(t_td * 1e3, 2),
                                         "batch": Bt, "ber": float((bt != bt_hat).float().mean()),
                                         "stages": "rocFFT OFDM mod/demod, cir_to_time_channel + ApplyTimeChannel "
                                                   f"(l_min={l_min}, l_max={l_max}), LS with linear interpolation"}
    if not args.no_cpu_baseline:
        from oracle import ofdm as o, mimo_f32 as of32
        org = o.ResourceGrid(14, 76, 15e3, num_tx=1, num_streams_per_tx=2, cyclic_prefix_length=6,
                             num_guard_carriers=[5, 6], dc_null=True, pilot_pattern="kronecker",
                             pilot_ofdm_symbol_indices=[2, 11])
        osm = o.StreamManagement([[1]], 2)
        ns = min(B, 2048 if short else 8192)
        yc, hc, evc = y[:ns].cpu().numpy(), h_hat[:ns].cpu().numpy(), ev.cpu().numpy()
        t0 = time.perf_counter()
        xo, _ = of32.ofdm_equalize(org, osm, yc, hc, evc, np.float32(float(no)))
        t_cpu = time.perf_counter() - t0
        x_gpu, _ = eq(y[:ns], h_hat[:ns], ev, no)
        out["cpu_baseline"] = {"value": round(ns * rg.num_data_symbols / t_cpu, 1), "unit": "resource-elements/s",
                               "cores": 1, "kind": "port",
                               "sample": f"{ns} batch items, oracle/mimo_f32.py (NumPy float32, the order-defined oracle), {t_cpu:.1f} s",
                               "bit_exact_with_gpu": bool(np.array_equal(x_gpu.cpu().numpy(), xo))}
    return out


# ------------------------------------------------------------------ C5: Polar SCL-8
def bench_c5(args, short=False):
    """BASELINE config C5: Polar5G uplink k=512 n=1024 (CRC11, k_polar=523), CRC-aided SCL list 8, batch 32768.
    One step = Polar5GDecoder.call on resident LLRs; metric codeword-decodes/s; VALU-issue roofline (the kernel keeps
    the list state on chip)."""
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    phy.config.seed = 5
    steps, warmup = (max(3, args.steps // 2), 1) if short else (args.steps, args.warmup)
    B, k, n, m = (args.batch if args.batch != 65536 else 32768), 512, 1024, 2
    ebno = 2.5 if args.ebno_db == 4.5 else args.ebno_db
    enc = phy.fec.polar.Polar5GEncoder(k, n)
    dec = phy.fec.polar.Polar5GDecoder(enc, "SCL", list_size=8)
    no = phy.utils.ebnodb2no(ebno, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    y = phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no)
    llr = phy.mapping.Demapper("app", "qam", m)(y, no)
    torch.cuda.synchronize()
    for _ in range(warmup):
        dec(llr)
    ev_t = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e0, e1 in ev_t:
        e0.record(); u_hat = dec(llr); e1.record()
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    ms = float(np.mean([a.elapsed_time(c) for a, c in ev_t]))
    out = {"metric": "codeword-decodes/sec (Polar5G n=1024 k=512, SCL-8)", "value": round(B * steps / t_wall, 1),
           "unit": "codewords/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
           "ms_per_step": round(t_wall / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"C5: Polar5G uplink k=512 n=1024 (CRC11), SCL list 8, QPSK AWGN, batch {B}",
                      "batch": B, "ebno_db": ebno},
           "bler": float((u_hat != u).any(dim=1).float().mean()),
           "roofline": onchip_roofline("polar_scl", "polar_scl_reg_kernel<8> (one wave per codeword, low stages in registers)", B, ms,
                                       {"compulsory_io_gbps": round((4 * n + 4 * k) * B / (ms * 1e-3) / 1e9, 2),
                                        "traffic_note": "traffic = the L2 scratch of the top tree stages (40 KB per resident "
                                                        "codeword x 8192 codewords exceed L2 + MALL and stream through HBM), "
                                                        "not re-reads of the 4n + 4k compulsory bytes"})}
    if not args.no_cpu_baseline:
        from oracle import polar as op, polar_c as pc
        code = op.Polar5GCode(k, n)
        cores = pc.num_threads()
        ns = min(B, (32 if short else 96) * cores)
        sample = llr[:ns].cpu().numpy()
        t0 = time.perf_counter()
        ref = pc.polar5g_decode(code, sample, 8)
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(ns / t_cpu, 1), "unit": "codewords/s", "cores": cores, "kind": "port",
                               "sample": f"{ns} codewords of the same LLR batch, oracle/polar_scl.c (float32 specification "
                                         f"arithmetic, OpenMP over codewords), {t_cpu:.1f} s",
                               "bit_exact_with_gpu": bool(np.array_equal(ref, u_hat[:ns].cpu().numpy()))}
    return out


def bench_c5_bp(args, short=False):
    """The C5 code under the OTHER iterative Polar decoder: PolarBPDecoder, 20 flooding iterations (reference
    fec/polar/decoding.py:1440-1771), n = 1024, k_polar = 523, batch 32768 - one launch of samd_polar_bp_decode_f32 per step,
    the factor graph of a codeword in the LDS of its workgroup.  Rate and kernel time only: no PMC counters exist for this
    kernel yet (written after the round's GPU minutes were spent), so `roofline.achieved` is the static instruction estimate
    of DESIGN section 4 over the measured time, marked as an estimate."""
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    phy.config.seed = 6
    steps, warmup = (max(3, args.steps // 2), 1) if short else (args.steps, args.warmup)
    B, k, n, m, it = (args.batch if args.batch != 65536 else 32768), 512, 1024, 2, 20
    ebno = 3.0 if args.ebno_db == 4.5 else args.ebno_db
    enc = phy.fec.polar.Polar5GEncoder(k, n)
    dec = phy.fec.polar.Polar5GDecoder(enc, "BP", num_iter=it)
    no = phy.utils.ebnodb2no(ebno, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    llr = phy.mapping.Demapper("app", "qam", m)(phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no), no)
    torch.cuda.synchronize()
    for _ in range(warmup):
        dec(llr)
    ev_t = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e0, e1 in ev_t:
        e0.record(); u_hat = dec(llr); e1.record()
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    ms = float(np.mean([a.elapsed_time(c) for a, c in ev_t]))
    stages = 10
    est_inst = it * (2 * stages - 2) * (n // 2) * 100 / 64                 # ~100 vector instructions per butterfly and lane
    kname = "polar_bp_kernel<false> (messages of a codeword in LDS, 512 lanes per codeword)"
    if load_counters("polar_bp"):                                           # PMC counters exist: the measured roofline
        roof = onchip_roofline("polar_bp", kname, B, ms, {"compulsory_io_gbps": round((4 * n + 4 * k) * B / (ms * 1e-3) / 1e9, 2)})
    else:
        roof = {"bound": "valu", "achieved": round(est_inst * B / (ms * 1e-3) / 1e9, 1), "peak": 1228.8,
                "unit": "G wave64-inst/s", "frac": round(est_inst * B / (ms * 1e-3) / 1e9 / 1228.8, 4), "traffic": None,
                "kernel": kname, "ms_per_launch": round(ms, 3), "estimate": "static instruction count, not PMC",
                "compulsory_io_gbps": round((4 * n + 4 * k) * B / (ms * 1e-3) / 1e9, 2)}
    return {"metric": "codeword-decodes/sec (Polar5G n=1024 k=512, BP-20)", "value": round(B * steps / t_wall, 1),
            "unit": "codewords/s", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": round(t_wall / steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C5 code, PolarBPDecoder 20 iterations through Polar5GDecoder(dec_type='BP'), QPSK AWGN, batch {B}",
                       "batch": B, "ebno_db": ebno},
            "bler": float((u_hat != u).any(dim=1).float().mean()),
            "roofline": roof}


# ------------------------------------------------------------------ C2 / C3: LDPC decode (headline)
def cpu_baseline_c2(llr, k, n, m, cn_update, num_iter, dec, seconds, max_cw):
    from oracle import cbind, ldpc_bp as obp
    from oracle.ldpc5g import LDPC5GCode
    odec = obp.LDPC5GDecoder(LDPC5GCode(k, n, m, BG), cn_update=cn_update, num_iter=num_iter)
    cores = cbind.num_threads()
    probe = odec.rate_recover(llr[:min(llr.shape[0], 8 * cores)].cpu().numpy())
    t0 = time.perf_counter()
    cbind.bp_decode(odec, probe, simd=True)
    t_probe = max(time.perf_counter() - t0, 1e-3)
    ns = int(min(llr.shape[0], max_cw, max(8 * cores, seconds / (t_probe / len(probe)))))
    ns = max(8 * cores, ns // (8 * cores) * (8 * cores))
    ns = min(ns, llr.shape[0])
    l5 = odec.rate_recover(llr[:ns].cpu().numpy())
    t0 = time.perf_counter()
    ref = cbind.bp_decode(odec, l5, simd=True)       # 8 codewords per vector (AVX2), OpenMP over the groups
    t_cpu = time.perf_counter() - t0
    agree = float(np.mean(ref[:, :k] == dec(llr[:ns]).cpu().numpy()))
    return {"value": round(ns / t_cpu, 2), "unit": "codewords/s", "cores": cores, "kind": "port",
            "sample": f"{ns} codewords of the same C2 LLR batch, oracle/ldpc_bp.c ({cn_update}, {num_iter} iterations, 8 codewords "
                      f"per AVX2 vector, OpenMP over the groups; bit-identical to the scalar oracle), {t_cpu:.1f} s",
            "hard_decision_agreement_with_gpu": agree}


def c2_roofline(cn_update, onchip, B, k, num_iter, dec_ms, specialised=False):
    """specialised: the kernel generated for the code ran (csrc/ldpc5g_jit.cpp) - min-sum family or, since round 6, boxplus-phi"""
    bytes_alg = b_msg(num_iter, k) * B
    equiv = {"algorithmic_bytes_per_decode": b_msg(num_iter, k), "gbps": round(bytes_alg / (dec_ms * 1e-3) / 1e9, 1),
             "frac_of_hbm_peak": round(bytes_alg / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
             "note": "SURVEY 8(d) B_msg of the HBM-resident formulation; may exceed 1 because the messages never leave LDS"}
    io = {"compulsory_io_gbps": round((4 * N_CW + 4 * k) * B / (dec_ms * 1e-3) / 1e9, 1)}
    if onchip:
        minsum = cn_update in ("minsum", "offset-minsum")
        key = "ldpc5g_ms" if minsum else ("ldpc5g_bp_fast" if cn_update == "boxplus-phi-fast" else "ldpc5g_bp")
        name = ("ldpc5g_decode_msg_kernel (on-chip min-sum, one float per edge in LDS, channel LLRs in an L2 workspace row, "
                "grouped dispatch)" if minsum else
                "ldpc5g_decode_msg_kernel<..., boxplus> (the same engine with bp_math's boxplus node update)")
        if specialised and minsum:
            key = "ldpc5g_jit"
            name = ("samd_ldpc5g_jit (generated for this code and compiled with hipRTC at the first decode: one straight-line "
                    "program per wave, messages in LDS, channel LLRs and block positions in registers; csrc/ldpc5g_jit.cpp)")
        elif specialised and cn_update == "boxplus-phi":
            key = "ldpc5g_jit_phi"
            name = ("samd_ldpc5g_jit_phi (generated for this code: the defined phi with the check-node loops over a row's edges "
                    "rolled, one phi body per pass, values through the row's own LDS slots; csrc/ldpc5g_jit.cpp)")
        return onchip_roofline(key, name, B, dec_ms, {"hbm_resident_equiv": equiv, **io})
    ach = bytes_alg / (dec_ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
            "traffic": None, "kernel": "cn_pass_kernel + vn_pass_kernel (HBM-resident, 2 launches per iteration)",
            "algorithmic_bytes_per_decode": b_msg(num_iter, k), "decoder_ms_per_launch_set": round(dec_ms, 3), **io}


def dry_dist(args, world, rank):
    """--dry-dist: the N-rank control path of this file (self-launch, env rendezvous, counted_step, barrier-bracketed
    timing, MAX over ranks, SUM of the counters, rank 0 prints ONE line) with a stand-in decoder on host tensors."""
    import torch.distributed as dist
    B, k = min(args.batch, 256), 64
    g = torch.Generator().manual_seed(1000 + rank)
    u = torch.randint(0, 2, (B, k), generator=g).float()
    flips = (torch.rand((B, k), generator=g) < 0.01).float()

    def count_into(b, b_hat, acc):
        acc[0] += int((b != b_hat).sum())
        acc[1] += int((b != b_hat).any(dim=1).sum())

    counters = torch.zeros(4, dtype=torch.int64)
    step = counted_step(lambda: (u + flips) % 2, u, counters, count_into, world)
    t_wall, c = timed_steps(step, args.steps, args.warmup, world, torch.device("cpu"), lambda: None, counters)
    seen, distinct = ranks_seen(world, rank, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"metric": "dry-dist (control path only, no kernels)", "value": round(B * world * args.steps / t_wall, 1),
                          "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(t_wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "data": "synthetic", "config": {"workload": "dry-dist", "parallelism": f"dp{world}", "batch_per_gpu": B},
                          "counters": [int(x) for x in c], "local_bits_per_rank": B * k * args.steps,
                          "ranks_seen": seen, "distinct_random_streams": distinct}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536, help="codewords per GPU")
    ap.add_argument("--num-iter", type=int, default=20)
    ap.add_argument("--cn-update", default="minsum", choices=["minsum", "offset-minsum", "boxplus-phi", "boxplus-phi-fast", "boxplus"])
    ap.add_argument("--engine", default="auto", choices=["auto", "generic"],
                    help="generic forces the HBM-resident decoder also for min-sum")
    ap.add_argument("--ebno-db", type=float, default=4.5)
    ap.add_argument("--also", default="boxplus-phi", help="second CN rule timed with fewer steps ('none' disables)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short C4 / C5 sub-lines")
    ap.add_argument("--dry-dist", action="store_true",
                    help="launch / rendezvous / step / reduce logic only, on host tensors over gloo (no GPU, no kernels): "
                         "what tests/test_bench_dist.py runs through the real `--gpus N` self-launch")
    ap.add_argument("--workload", default="c2", choices=["c2", "c4", "c5", "c5_bp"],
                    help="c2 = headline LDPC decode (default); c4 = OFDM 4x2 LMMSE pass; c5 = Polar SCL-8; c5_bp = Polar BP-20 "
                         "on the C5 code (single GPU)")
    args = ap.parse_args()
    if args.workload == "c5_bp":
        return print(json.dumps(bench_c5_bp(args)))
    if args.workload == "c4":
        return print(json.dumps(bench_c4(args)))
    if args.workload == "c5":
        return print(json.dumps(bench_c5(args)))

    world, rank, local_rank = world_info()
    if args.gpus > 1 and world == 1 and not os.environ.get("SAMD_BENCH_CHILD"):
        return self_launch(args.gpus)       # not started as one of N ranks: become the launcher of N ranks (never returns)
    import torch.distributed as dist
    # SAMD_BENCH_BACKEND=gloo: code-path check of the N>1 logic on a box with fewer GPUs than ranks
    # (ranks then share devices); the driver's runs use the default, RCCL.
    backend = "gloo" if args.dry_dist else os.environ.get("SAMD_BENCH_BACKEND", "nccl")
    if not args.dry_dist:
        if backend != "nccl":
            local_rank %= max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
    if world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ):   # launched as a rank (also as the only one)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if args.dry_dist:
        return dry_dist(args, world, rank)

    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    dev = _ffi.device()
    phy.config.seed = 20250923

    B, k, n, m = args.batch, K_INFO, N_CW, M_BITS
    enc = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=BG)
    src, mapper, awgn = phy.mapping.BinarySource(), phy.mapping.Mapper("qam", m), phy.channel.AWGN()
    demap = phy.mapping.Demapper("app", "qam", m)
    no = phy.utils.ebnodb2no(args.ebno_db, m, k / n)

    def make_dec(cn):
        d = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=cn, num_iter=args.num_iter, hard_out=True, return_infobits=True)
        if args.engine == "generic":
            d._onchip_ok = False
        return d

    # ---- synthetic batch, resident in HBM before the timed region
    t0 = time.perf_counter()
    u = src([B, k])
    llr = demap(awgn(mapper(enc(u)), no), no)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    counters = torch.zeros(4, dtype=torch.int64, device=dev)

    def run(dec, steps, warmup):
        # untimed preparation, independent of --warmup: the first decode of a handle generates and compiles its kernel (hipRTC,
        # ~4 s of host time during which the GPU idles and drops its clocks); two more decodes bring the clocks back.  With
        # --warmup 1 and this missing, the timed steps measured the clock ramp (20.4 instead of 16.1 ms per step, profiles/r05smoke)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        step = counted_step(lambda: dec(llr), u, counters, phy.utils.metrics.count_errors_into, world, ev)
        for _ in range(3):
            step(None)                      # (the counters are zeroed again before the timed region)
        torch.cuda.synchronize()
        t_wall, c = timed_steps(step, steps, warmup, world, dev, torch.cuda.synchronize, counters)
        return t_wall, float(np.mean([a.elapsed_time(b) for a, b in ev])), c

    def specialised_ran(d):
        """did this decoder's launches go to the kernel generated for the code (csrc/ldpc5g_jit.cpp)?"""
        try:
            return int(_ffi.lib().samd_ldpc5g_jit_launches(enc._handle(d._nb_pruned_nodes))) > 0
        except Exception:                                # pylint: disable=broad-except
            return False

    dec = make_dec(args.cn_update)
    t_wall, dec_ms, c = run(dec, args.steps, args.warmup)
    onchip = bool(dec._onchip_ok)
    jit_ran = onchip and args.cn_update in ("minsum", "offset-minsum") and specialised_ran(dec)
    value = B * world * args.steps / t_wall
    out = {
        "metric": "codeword-decodes/sec (n=8448, BP iters=20)", "value": round(value, 1),
        "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(t_wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2: 5G LDPC BG1 k=2816 n=8448 (rate 1/3, Z=128), 64-QAM AWGN LLRs, flooding BP "
                               f"{args.num_iter} iterations, batch {B} per GPU",
                   "cn_update": args.cn_update,
                   "engine": ("on-chip, kernel specialised for the code (hipRTC)" if jit_ran else "on-chip") if onchip else "generic-hbm",
                   "batch_per_gpu": B, "ebno_db": args.ebno_db, "parallelism": f"dp{world}"},
        "ber": float(c[0] / max(c[2], 1)), "bler": float(c[1] / max(c[3], 1)),
        "input_generation_s": round(t_gen, 3),
        "roofline": c2_roofline(args.cn_update, onchip, B, k, args.num_iter, dec_ms, jit_ran),
    }

    if jit_ran:
        # the same decode on the GENERIC on-chip kernel (the lists walked by scalar code, ldpc5g_decode_msg_kernel) in the same
        # run: what the specialisation buys, and that both produce the same decisions
        with _ffi.option("SAMD_LDPC_JIT", "0"):
            enc_g = phy.fec.ldpc.LDPC5GEncoder(k, n, num_bits_per_symbol=m, bg=BG)
            dec_g = phy.fec.ldpc.LDPC5GDecoder(enc_g, cn_update=args.cn_update, num_iter=args.num_iter, hard_out=True, return_infobits=True)
            steps_g = max(2, args.steps // 3)
            tg, ms_g, _ = run(dec_g, steps_g, 1)
            same = bool(torch.equal(dec_g(llr).as_subclass(torch.Tensor), dec(llr).as_subclass(torch.Tensor)))
        out["generic_kernel"] = {"value": round(B * world * steps_g / tg, 1), "unit": "codewords/s", "steps": steps_g,
                                 "ms_per_launch": round(ms_g, 3), "same_decisions_as_specialised": same,
                                 "speedup_of_specialised": round((B * world * args.steps / t_wall) / (B * world * steps_g / tg), 3),
                                 "roofline": c2_roofline(args.cn_update, True, B, k, args.num_iter, ms_g, False)}

    # whole chain source -> encoder -> mapper -> AWGN -> demapper -> decoder -> counters (SURVEY 8d "e2e")
    def chain_step():
        ub = src([B, k])
        ll = demap(awgn(mapper(enc(ub)), no), no)
        phy.utils.metrics.count_errors_into(ub, dec(ll), counters[:2])
    chain_step()
    barrier(world, torch.cuda.synchronize)
    t0 = time.perf_counter()
    n_e2e = 3
    for _ in range(n_e2e):
        chain_step()
    barrier(world, torch.cuda.synchronize)
    t_e2e = (time.perf_counter() - t0) / n_e2e
    out["end_to_end"] = {"codewords_per_s_per_gpu": round(B / t_e2e, 1), "ms_per_batch": round(t_e2e * 1e3, 2),
                         "stages": "BinarySource, LDPC5GEncoder, Mapper, AWGN, Demapper(app), LDPC5GDecoder, count_errors"}

    if not args.no_extra:
        # The Monte-Carlo driver itself (north_star's use case; reference utils/misc.py:694-784): the same chain as mc_fun of
        # sim_ber over three Eb/N0 points.  With num_target_block_errors the driver must read the counters after EVERY iteration
        # (one device synchronisation - and, for N > 1, one all-reduce of 4 x int64 - per iteration), exactly where the reference
        # evaluates its stopping rules; without a target the counters stay on the device until the point ends.
        def mc_fun(batch_size, ebno_db):
            ub = src([batch_size, k])
            no_p = phy.utils.ebnodb2no(float(ebno_db), m, k / n)
            return ub, dec(demap(awgn(mapper(enc(ub)), no_p), no_p))
        n_mc, points = 4, [0.0, 0.5, 1.0]              # block error rate 1 at these points: no early stop, every iteration runs
        sims = {}
        flushes = {"n": 0}
        orig_cpu = torch.Tensor.cpu
        def counting_cpu(self_, *a, **kw):              # device -> host reads inside sim_ber = its synchronisations
            flushes["n"] += 1
            return orig_cpu(self_, *a, **kw)
        import contextlib
        import io
        quiet = contextlib.redirect_stdout(io.StringIO())           # (sim_ber announces the distribution on stdout: this file prints ONE line)
        for tag, kw in (("target_block_errors", {"num_target_block_errors": 10 ** 12}), ("no_target", {})):
            with quiet:
                phy.utils.sim_ber(mc_fun, points[:1], B, world, verbose=False, distribute=("all" if world > 1 else None), **kw)   # warm
            barrier(world, torch.cuda.synchronize)
            flushes["n"] = 0
            torch.Tensor.cpu = counting_cpu
            try:
                t0 = time.perf_counter()
                with quiet:
                    ber_s, bler_s = phy.utils.sim_ber(mc_fun, points, B, n_mc * world, early_stop=True, verbose=False,
                                                      distribute=("all" if world > 1 else None), **kw)
                barrier(world, torch.cuda.synchronize)
                t_s = time.perf_counter() - t0
            finally:
                torch.Tensor.cpu = orig_cpu
            sims[tag] = {"codewords_per_s_per_gpu": round(len(points) * n_mc * B / t_s, 1),
                         "ms_per_iteration": round(t_s / (len(points) * n_mc) * 1e3, 2),
                         "host_syncs_per_iteration": round(flushes["n"] / (len(points) * n_mc), 3),
                         "fraction_of_raw_chain": round((len(points) * n_mc * B / t_s) / (B / t_e2e), 4),
                         "bler": [float(v) for v in np.asarray(bler_s)]}
        out["sim_ber"] = {"what": f"sim_ber(mc_fun = the end_to_end chain, ebno_dbs = {points}, batch_size = {B}, max_mc_iter = {n_mc} per rank, "
                                  "early_stop = True) - codewords per second THROUGH the Monte-Carlo driver, beside end_to_end (the raw chain)",
                          **sims}

    dec2 = None
    if args.also and args.also != "none" and args.also != args.cn_update:
        dec2 = make_dec(args.also)
        steps2 = max(2, args.steps // 3)
        n_before = int(_ffi.lib().samd_ldpc5g_jit_launches(enc._handle(dec2._nb_pruned_nodes)))
        t2, ms2, c2 = run(dec2, steps2, 1)
        on2 = bool(dec2._onchip_ok)
        jit2 = on2 and int(_ffi.lib().samd_ldpc5g_jit_launches(enc._handle(dec2._nb_pruned_nodes))) > n_before
        out["also"] = {"cn_update": args.also,
                       "engine": ("on-chip, kernel specialised for the code (hipRTC)" if jit2 else "on-chip") if on2 else "generic-hbm",
                       "note": "the reference's default check-node rule",
                       "value": round(B * world * steps2 / t2, 1), "unit": "codewords/s", "steps": steps2,
                       "ms_per_step": round(t2 / steps2 * 1e3, 3), "ber": float(c2[0] / max(c2[2], 1)),
                       "bler": float(c2[1] / max(c2[3], 1)),
                       "roofline": c2_roofline(args.also, on2, B, k, args.num_iter, ms2, jit2)}

        if args.also == "boxplus-phi":
            # the same rule on the GPU's transcendental unit (SAMD_CN_BOXPLUS_PHI_FAST): not bit-defined, reported beside
            dec3 = make_dec("boxplus-phi-fast")
            t3, ms3, c3 = run(dec3, steps2, 1)
            out["also"]["fast_math"] = {"cn_update": "boxplus-phi-fast", "value": round(B * world * steps2 / t3, 1),
                                        "unit": "codewords/s", "ms_per_step": round(t3 / steps2 * 1e3, 3),
                                        "ber": float(c3[0] / max(c3[2], 1)), "bler": float(c3[1] / max(c3[3], 1)),
                                        "note": "v_exp_f32 / v_log_f32 instead of the defined exp / log; soft outputs "
                                                "are not bit-identical to the oracle (tests keep the round-2 tolerance bars)"}
            del dec3

    if args.also and args.also != "none":
        # cn_schedule="layered" (one sub-iteration per base row), half the iterations, on its own
        # on-chip engine (csrc/ldpc5g_onchip_ly.hip); same batch, same rule as the headline
        decl = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=args.cn_update, cn_schedule="layered", num_iter=max(1, args.num_iter // 2),
                                          hard_out=True, return_infobits=True)
        stepsl = max(2, args.steps // 3)
        tl, msl, cl = run(decl, stepsl, 1)
        out.setdefault("also", {})["layered"] = {
            "cn_schedule": "layered", "cn_update": args.cn_update, "num_iter": max(1, args.num_iter // 2),
            "engine": "on-chip layered" if _ffi.lib().samd_ldpc5g_decode_layered_supported(
                enc._handle(decl._nb_pruned_nodes), decl._cn_mode) else "generic-hbm scheduled",
            "value": round(B * world * stepsl / tl, 1), "unit": "codewords/s", "steps": stepsl,
            "ms_per_step": round(tl / stepsl * 1e3, 3), "ber": float(cl[0] / max(cl[2], 1)), "bler": float(cl[1] / max(cl[3], 1))}
        del decl

    if args.also and args.also != "none" and world == 1:
        # return_state / msg_v2c (the iterative detection-and-decoding use): the same 20 iterations as four calls
        # of five with the decoder state handed from call to call; a quarter of the batch (an image and its deferred tensor per call)
        Bs = max(1024, B // 4)
        decs = phy.fec.ldpc.LDPC5GDecoder(enc, cn_update=args.cn_update, num_iter=max(1, args.num_iter // 4), hard_out=True,
                                          return_infobits=True, return_state=True)
        llrs = llr[:Bs]

        def idd():
            x, st = decs(llrs)
            for _ in range(3):
                x, st = decs(llrs, msg_v2c=st)
            return x
        x = idd()
        torch.cuda.synchronize()
        same = bool(torch.equal(x.as_subclass(torch.Tensor), dec(llrs).as_subclass(torch.Tensor)))
        t0 = time.perf_counter()
        for _ in range(3):
            idd()
        torch.cuda.synchronize()
        ts = (time.perf_counter() - t0) / 3
        decs._onchip_ok = False                      # the HBM-resident engine (two launches per iteration), the path of rounds 1-5
        idd()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        idd()
        torch.cuda.synchronize()
        th = time.perf_counter() - t0
        out.setdefault("also", {})["return_state"] = {
            "what": f"LDPC5GDecoder(return_state=True): 4 calls x {max(1, args.num_iter // 4)} iterations, msg_v2c handed from call to call, "
                    f"batch {Bs}", "cn_update": args.cn_update,
            "engine": "on-chip, state variant of the generated kernel (message image in / out, [num_edges, batch] tensor deferred)"
                      if getattr(decs, "_state_lay", (None, None))[1] is not None else "generic-hbm",
            "value": round(Bs / ts, 1), "unit": "codewords/s", "ms_per_call": round(ts / 4 * 1e3, 3),
            "same_decisions_as_one_call": same,
            "hbm_resident_engine": {"value": round(Bs / th, 1), "unit": "codewords/s", "ms_per_call": round(th / 4 * 1e3, 3)}}
        del decs, llrs

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_c2(llr, k, n, m, args.cn_update, args.num_iter, dec, 12.0, B)
        out["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
        if dec2 is not None:
            out["also"]["cpu_baseline"] = cpu_baseline_c2(llr, k, n, m, args.also, args.num_iter, dec2, 8.0, B)
            out["also"]["speedup_vs_cpu_baseline"] = round(out["also"]["value"] / out["also"]["cpu_baseline"]["value"], 1)

    if rank == 0 and world == 1 and not args.no_extra:
        del llr, u
        torch.cuda.empty_cache()
        out["extra"] = {}
        for name, fn in (("c4", bench_c4), ("c5", bench_c5), ("c5_bp", bench_c5_bp)):
            try:
                out["extra"][name] = fn(args, short=True)
            except Exception as e:  # pylint: disable=broad-except  (a secondary workload must not lose the headline line)
                out["extra"][name] = {"error": f"{type(e).__name__}: {e}"}

    if _grouped(world):
        seen, distinct = ranks_seen(world, rank, dev)          # a collective of THIS run: who took part, are the streams distinct
        out["collectives"] = {"backend": dist.get_backend(), "group_size": dist.get_world_size()}
        out["rccl_ranks_seen"] = seen
        out["distinct_random_streams"] = distinct
    if rank == 0:
        print(json.dumps(out))
    if _grouped(world):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
