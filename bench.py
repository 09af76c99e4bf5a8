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

`roofline`  : dominant kernel of the timed step, ALGORITHMIC bytes B_msg (SURVEY 8d:
              13,684,736 B per C2 decode) / its average duration measured here with HIP events
              on the launch stream; peak = 8 TB/s HBM3E.  For the on-chip min-sum engine the
              message traffic never reaches HBM, so frac may exceed 1 - `compulsory_io_gbps`
              (4n+4k bytes per decode) is the traffic that really crosses HBM.
`cpu_baseline`: oracle/ldpc_bp.c (plain-C restatement of the reference algorithm, OpenMP over
              codewords, all host cores) on a bounded sample of the same LLRs, rank 0, N=1.
"""
import argparse
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


def b_msg(num_iter, k_out):
    """algorithmic bytes per decode of the HBM-resident formulation."""
    return num_iter * (16 * N_EDGES + 4 * N_VN) + 4 * N_CW + 4 * k_out


def bench_c4(args):
    """Secondary workload (BASELINE config C4): OFDM 14x76, TDL-A 300 ns, 4x2, LS-NN + fused per-RE
    LMMSE, QPSK + LDPC k=768 n=1536 per stream, batch 8192.  One step = LMMSEEqualizer.call on a
    resident batch; metric resource-elements/s; roofline = 120 algorithmic bytes per RE (y 32 + H 64 in,
    x_hat 16 + no_eff 8 out) against 8 TB/s.  Also reports the end-to-end chain rate."""
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    phy.config.seed = 4
    B, k, n, m = args.batch if args.batch != 65536 else 8192, 768, 1536, 2
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
    no = phy.utils.ebnodb2no(args.ebno_db, m, k / n, rg)

    def chain():
        b = src([B, 1, 2, k])
        y, h = ch(rgm(mapper(enc(b))), no)
        h_hat, ev = est(y, no)
        x_hat, no_eff = eq(y, h_hat, ev, no)
        return b, dec(demap(x_hat, no_eff)), (y, h_hat, ev)

    b, b_hat, (y, h_hat, ev) = chain()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        eq(y, h_hat, ev, no)
    ev_t = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e0, e1 in ev_t:
        e0.record(); eq(y, h_hat, ev, no); e1.record()
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    ms = float(np.mean([a.elapsed_time(c) for a, c in ev_t]))
    n_re = B * 14 * 64                               # REs visited (pilot symbols are skipped inside)
    n_data_re = B * rg.num_data_symbols
    ach = n_data_re * 120 / (ms * 1e-3) / 1e9
    t0 = time.perf_counter()
    for _ in range(3):
        chain()
    torch.cuda.synchronize()
    t_e2e = (time.perf_counter() - t0) / 3
    # time-domain variant of the same chain: OFDMModulator (rocFFT) -> TimeChannel (TDL taps, time-varying
    # FIR) -> OFDMDemodulator (rocFFT) in place of the frequency-domain channel
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
    out = {"metric": "LMMSE-equalised resource elements/sec (4x2, config C4)", "value": round(n_data_re * args.steps / t_wall, 1),
           "unit": "resource-elements/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(t_wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "c64", "data": "synthetic",
           "config": {"workload": f"C4: OFDM 14x76 (64 eff. subcarriers, pilots at symbols 2,11), TDL-A 300 ns, 4 rx x 2 streams, "
                                  f"LS-NN + LMMSE, QPSK, LDPC k=768 n=1536 per stream, batch {B}", "batch": B,
                      "ebno_db": args.ebno_db},
           "ber": float((b != b_hat).float().mean()),
           "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": "ofdm_lmmse_kernel<4,2>",
                        "algorithmic_bytes_per_re": 120, "ms_per_launch": round(ms, 3)},
           "end_to_end": {"codewords_per_s": round(2 * B / t_e2e, 1), "ms_per_batch": round(t_e2e * 1e3, 2)},
           "end_to_end_time_domain": {"codewords_per_s": round(2 * Bt / t_td, 1), "ms_per_batch": round(t_td * 1e3, 2),
                                      "batch": Bt, "ber": float((bt != bt_hat).float().mean()),
                                      "stages": "rocFFT OFDM mod/demod, cir_to_time_channel + ApplyTimeChannel "
                                                f"(l_min={l_min}, l_max={l_max}), LS with linear interpolation"}}
    if not args.no_cpu_baseline:
        from oracle import ofdm as o
        org = o.ResourceGrid(14, 76, 15e3, num_tx=1, num_streams_per_tx=2, cyclic_prefix_length=6,
                             num_guard_carriers=[5, 6], dc_null=True, pilot_pattern="kronecker",
                             pilot_ofdm_symbol_indices=[2, 11])
        osm = o.StreamManagement([[1]], 2)
        ns = 256
        yc, hc, evc = y[:ns].cpu().numpy(), h_hat[:ns].cpu().numpy(), ev.cpu().numpy()
        t0 = time.perf_counter()
        o.ofdm_lmmse_equalize(org, osm, yc, hc, evc, float(no))
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(ns * rg.num_data_symbols / t_cpu, 1), "unit": "resource-elements/s",
                               "cores": int(os.cpu_count() or 1), "kind": "port",
                               "sample": f"{ns} batch items, oracle/ofdm.py (NumPy complex128 batched linalg), {t_cpu:.1f} s"}
    print(json.dumps(out))


def bench_c5(args):
    """Secondary workload (BASELINE config C5): Polar5G uplink k=512 n=1024 (CRC11, k_polar=523),
    CRC-aided SCL list 8, batch 32768.  One step = Polar5GDecoder.call on resident LLRs; metric
    codeword-decodes/s.  The kernel is synchronisation / latency bound ("report
    decodes/s and occupancy only"); the roofline entry states the compulsory HBM bytes."""
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    _ffi.device()
    phy.config.seed = 5
    B, k, n, m = (args.batch if args.batch != 65536 else 32768), 512, 1024, 2
    enc = phy.fec.polar.Polar5GEncoder(k, n)
    dec = phy.fec.polar.Polar5GDecoder(enc, "SCL", list_size=8)
    no = phy.utils.ebnodb2no(args.ebno_db, m, k / n)
    u = phy.mapping.BinarySource()([B, k])
    y = phy.channel.AWGN()(phy.mapping.Mapper("qam", m)(enc(u)), no)
    llr = phy.mapping.Demapper("app", "qam", m)(y, no)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        dec(llr)
    ev_t = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e0, e1 in ev_t:
        e0.record(); u_hat = dec(llr); e1.record()
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t0
    ms = float(np.mean([a.elapsed_time(c) for a, c in ev_t]))
    io = (4 * n + 4 * k) * B / (ms * 1e-3) / 1e9
    out = {"metric": "codeword-decodes/sec (Polar5G n=1024 k=512, SCL-8)", "value": round(B * args.steps / t_wall, 1),
           "unit": "codewords/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(t_wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"C5: Polar5G uplink k=512 n=1024 (CRC11), SCL list 8, QPSK AWGN, batch {B}",
                      "batch": B, "ebno_db": args.ebno_db},
           "bler": float((u_hat != u).any(dim=1).float().mean()),
           "roofline": {"bound": "hbm", "achieved": round(io, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": round(io / HBM_PEAK_GBPS, 5), "traffic": None, "kernel": "polar_scl_kernel",
                        "note": "compulsory 4n+4k bytes per codeword; the kernel is latency/synchronisation bound",
                        "ms_per_launch": round(ms, 3)}}
    if not args.no_cpu_baseline:
        from oracle import polar as op
        code = op.Polar5GCode(k, n)
        ns = 64
        t0 = time.perf_counter()
        op.polar5g_decode(code, llr[:ns].cpu().numpy(), "SCL", 8)
        t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(ns / t_cpu, 2), "unit": "codewords/s", "cores": 1, "kind": "port",
                               "sample": f"{ns} codewords, oracle/polar.py (NumPy float32 restatement of the reference's SCL), {t_cpu:.1f} s"}
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=65536, help="codewords per GPU")
    ap.add_argument("--num-iter", type=int, default=20)
    ap.add_argument("--cn-update", default="minsum", choices=["minsum", "offset-minsum", "boxplus-phi", "boxplus"])
    ap.add_argument("--engine", default="auto", choices=["auto", "generic"],
                    help="generic forces the HBM-resident decoder also for min-sum")
    ap.add_argument("--ebno-db", type=float, default=4.5)
    ap.add_argument("--also", default="boxplus-phi", help="second CN rule timed with fewer steps ('none' disables)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="codewords for the CPU baseline (0 = auto)")
    ap.add_argument("--workload", default="c2", choices=["c2", "c4", "c5"],
                    help="c2 = headline LDPC decode (default); c4 = OFDM 4x2 LMMSE pass; c5 = Polar SCL-8 (single GPU)")
    args = ap.parse_args()
    if args.workload == "c4":
        if args.ebno_db == 4.5:
            args.ebno_db = 10.0
        return bench_c4(args)
    if args.workload == "c5":
        if args.ebno_db == 4.5:
            args.ebno_db = 2.5
        return bench_c5(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs a torch.distributed.run launch", file=sys.stderr)
            sys.exit(2)
    import torch.distributed as dist
    # SAMD_BENCH_BACKEND=gloo: code-path check of the N>1 logic on a box with fewer GPUs than ranks
    # (ranks then share devices); the driver's runs use the default, RCCL.
    backend = os.environ.get("SAMD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step(dec, ev=None):
        if ev is not None:
            ev[0].record()          # HIP events on the launch stream bracket the decoder kernels only
        u_hat = dec(llr)
        if ev is not None:
            ev[1].record()
        phy.utils.metrics.count_errors_into(u, u_hat, counters[:2])
        counters[2] += u.numel()
        counters[3] += B
        if world > 1:               # the path's only collective: 4 x int64 error counters (RCCL)
            red = counters.clone()
            dist.all_reduce(red, op=dist.ReduceOp.SUM)

    def run(dec, steps, warmup):
        for _ in range(warmup):
            step(dec)
        counters.zero_()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        barrier()
        t_start = time.perf_counter()
        for i in range(steps):
            step(dec, ev[i])
        barrier()
        t_wall = time.perf_counter() - t_start
        t = torch.tensor([t_wall], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dec_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        c = counters.clone()
        if world > 1:
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
        c = c.cpu().numpy()
        return float(t.item()), dec_ms, c

    dec = make_dec(args.cn_update)
    t_wall, dec_ms, c = run(dec, args.steps, args.warmup)
    onchip = bool(dec._onchip_ok)
    total_cw = B * world * args.steps
    value = total_cw / t_wall
    bytes_alg = b_msg(args.num_iter, k) * B
    achieved = bytes_alg / (dec_ms * 1e-3) / 1e9
    kernel = ("ldpc5g_decode_ms_kernel (on-chip min-sum, one float per edge in LDS, channel LLRs in an L2 workspace row)"
              if dec._cn_mode in (2, 3)
              else "ldpc5g_decode_bp_kernel (on-chip, one float per edge in LDS)") if onchip else \
             "cn_pass_kernel + vn_pass_kernel (HBM-resident, 2 launches per iteration)"
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": kernel,
                "algorithmic_bytes_per_decode": b_msg(args.num_iter, k),
                "decoder_ms_per_launch_set": round(dec_ms, 3),
                "compulsory_io_gbps": round((4 * n + 4 * k) * B / (dec_ms * 1e-3) / 1e9, 1)}
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(prof):                      # PMC-derived HBM bytes per launch, if recorded
        try:
            with open(prof) as f:
                tr = json.load(f)
            key = f"{args.cn_update}:{'onchip' if onchip else 'generic'}:B{B}"
            roofline["traffic"] = tr.get(key)
        except Exception:  # pylint: disable=broad-except
            pass
    if onchip:
        roofline["note"] = ("messages stay in LDS: frac is relative to the HBM-resident formulation's "
                            "algorithmic bytes and may exceed 1; real HBM traffic = compulsory_io "
                            "(+ write-backs of the L2 workspace rows, see traffic)")

    out = {
        "metric": "codeword-decodes/sec (n=8448, BP iters=20)", "value": round(value, 1),
        "unit": "codewords/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(t_wall / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "C2: 5G LDPC BG1 k=2816 n=8448 (rate 1/3, Z=128), 64-QAM AWGN LLRs, flooding BP "
                               f"{args.num_iter} iterations, batch {B} per GPU",
                   "cn_update": args.cn_update, "engine": "on-chip" if onchip else "generic-hbm",
                   "batch_per_gpu": B, "ebno_db": args.ebno_db, "parallelism": f"dp{world}"},
        "ber": float(c[0] / max(c[2], 1)), "bler": float(c[1] / max(c[3], 1)),
        "input_generation_s": round(t_gen, 3),
        "roofline": roofline,
    }

    # whole chain source -> encoder -> mapper -> AWGN -> demapper -> decoder -> counters (SURVEY 8d "e2e")
    def chain_step():
        ub = src([B, k])
        ll = demap(awgn(mapper(enc(ub)), no), no)
        phy.utils.metrics.count_errors_into(ub, dec(ll), counters[:2])
    chain_step()
    barrier()
    t0 = time.perf_counter()
    n_e2e = 3
    for _ in range(n_e2e):
        chain_step()
    barrier()
    t_e2e = (time.perf_counter() - t0) / n_e2e
    out["end_to_end"] = {"codewords_per_s_per_gpu": round(B / t_e2e, 1), "ms_per_batch": round(t_e2e * 1e3, 2),
                         "stages": "BinarySource, LDPC5GEncoder, Mapper, AWGN, Demapper(app), LDPC5GDecoder, count_errors"}

    if args.also and args.also != "none" and args.also != args.cn_update:
        dec2 = make_dec(args.also)
        steps2 = max(2, args.steps // 3)
        t2, ms2, c2 = run(dec2, steps2, 1)
        on2 = bool(dec2._onchip_ok)
        ach2 = bytes_alg / (ms2 * 1e-3) / 1e9
        out["also"] = {"cn_update": args.also, "engine": "on-chip" if on2 else "generic-hbm",
                       "value": round(B * world * steps2 / t2, 1), "unit": "codewords/s",
                       "decoder_ms": round(ms2, 3), "ber": float(c2[0] / max(c2[2], 1)),
                       "bler": float(c2[1] / max(c2[3], 1)),
                       "roofline": {"bound": "hbm", "achieved": round(ach2, 1), "peak": HBM_PEAK_GBPS,
                                    "unit": "GB/s", "frac": round(ach2 / HBM_PEAK_GBPS, 4),
                                    "note": "relative to the HBM-resident formulation's algorithmic bytes; "
                                            "on-chip engines keep the messages in LDS"}}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cbind, ldpc_bp as obp
        from oracle.ldpc5g import LDPC5GCode
        odec = obp.LDPC5GDecoder(LDPC5GCode(k, n, m, BG), cn_update=args.cn_update, num_iter=args.num_iter)
        cores = cbind.num_threads()
        sample = llr[:min(B, 4 * cores)].cpu().numpy()
        l5 = odec.rate_recover(sample)
        t0 = time.perf_counter()
        cbind.bp_decode(odec, l5)
        t_probe = max(time.perf_counter() - t0, 1e-3)
        ns = args.cpu_sample or int(min(B, max(4 * cores, 15.0 / (t_probe / len(l5)))))
        ns = max(cores, ns // cores * cores)
        sample = llr[:ns].cpu().numpy()
        l5 = odec.rate_recover(sample)
        t0 = time.perf_counter()
        ref = cbind.bp_decode(odec, l5)
        t_cpu = time.perf_counter() - t0
        agree = float(np.mean(ref[:, :k] == dec(llr[:ns]).cpu().numpy()))
        out["cpu_baseline"] = {"value": round(ns / t_cpu, 2), "unit": "codewords/s", "cores": cores,
                               "kind": "port",
                               "sample": f"{ns} codewords of the same C2 LLR batch, oracle/ldpc_bp.c "
                                         f"({args.cn_update}, {args.num_iter} iterations, OpenMP over codewords), "
                                         f"{t_cpu:.1f} s",
                               "hard_decision_agreement_with_gpu": agree}
        out["speedup_vs_cpu_baseline"] = round(value / (ns / t_cpu), 1)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
