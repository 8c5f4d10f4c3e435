#!/usr/bin/env python3
"""bench.py -- headline measurement of the Viterbi update()+chainback() hot path on MI355X.

One "step" = one pass of the hot path (reset -> update -> chainback) over one batch of synthetic AWGN frames already
resident in HBM.  Workload at N=1: BASELINE.json configs[1] -- Voyager K=7 R=1/2, u16 error metrics / s16 soft symbols,
65536 frames x 8192 info bits.  With --gpus N the frames of every rank are independent (weak scaling: each GPU decodes its
own 65536 frames); the only collective is the broadcast of the branch-table/config blob from rank 0 at set-up (RCCL).

Prints ONE JSON line on rank 0 (see the contract in the round prompt): value = decoded info Mbit/s over all GPUs.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=65536, help="frames per GPU")
    ap.add_argument("--bits", type=int, default=8192, help="info bits per frame (L)")
    ap.add_argument("--code", type=int, default=2, help="index into COMMON_CODES (2 = Voyager K=7 R=1/2)")
    ap.add_argument("--decode-type", default="SOFT16")
    ap.add_argument("--ebn0", type=float, default=3.0)
    ap.add_argument("--plan", default="auto", choices=["auto", "lds", "reg", "lds2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (nccl = RCCL; gloo only to exercise the N>1 path on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="test only: every rank uses cuda:0")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2],
                    help="2: double-buffered decision workspaces, chainback of step i on a second HIP stream beside the "
                         "update of step i+1; 1: both kernels back to back on one stream")
    return ap.parse_args()


def cpu_baseline(code_id, code, pc, decode_type, sym_host, tx_host, L, target_seconds):
    """The reference's own AVX2 strategy (oracle/_ref, kind "reference") on all host cores over a bounded sample of the
    same frames; falls back to the C restatement (kind "port", scalar) where _ref was never built."""
    import numpy as np
    from oracle import pyoracle

    pyoracle.ensure_built()
    dt = {"SOFT16": pyoracle.SOFT16, "SOFT8": pyoracle.SOFT8, "HARD8": pyoracle.HARD8}[decode_type]
    ocfg = pyoracle.stock_config(dt, code.R)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    frames = sym_host.shape[0]
    bits = frames * L
    res = {}
    if pyoracle.RefLib.available():
        ref = pyoracle.RefLib()
        simd = pyoracle.SIMD_AVX if ref.is_valid(code_id, pc.soft_bytes, pyoracle.SIMD_AVX) else pyoracle.SCALAR
        t1, out = ref.bench(code_id, ocfg, sym_host, frames, L, simd=simd, threads=cores, reps=1)
        reps = max(1, min(2000, int(target_seconds / max(t1, 1e-3))))
        t, out = ref.bench(code_id, ocfg, sym_host, frames, L, simd=simd, threads=cores, reps=reps)
        # single-thread scalar (the parity reference) on a slice, for context
        ns = max(1, min(frames, 64))
        ts, out_s = ref.bench(code_id, ocfg, sym_host[:ns], ns, L, simd=pyoracle.SCALAR, threads=1, reps=1)
        res = {"value": bits / t / 1e6, "unit": "Mbit/s", "cores": cores, "kind": "reference",
               "strategy": {pyoracle.SIMD_AVX: "ViterbiDecoder_AVX_u16/u8", pyoracle.SCALAR: "ViterbiDecoder_Scalar"}[simd],
               "sample": f"{frames} frames x {L} bits of the GPU batch, best of {reps} passes, {cores} threads "
                         f"(one decoder per thread, shared branch table)",
               "scalar_1thread_Mbit_s": ns * L / ts / 1e6}
        scalar_bytes = out_s
    else:
        oracle = pyoracle.Oracle()
        t0 = time.perf_counter()
        scalar_bytes, _, _ = oracle.decode_frames(code.K, code.R, code.G, ocfg, sym_host, L, threads=cores)
        t = time.perf_counter() - t0
        ns = frames
        res = {"value": bits / t / 1e6, "unit": "Mbit/s", "cores": cores, "kind": "port",
               "strategy": "oracle/viterbi_oracle.c (scalar restatement)",
               "sample": f"{frames} frames x {L} bits of the GPU batch, 1 pass, {cores} threads"}
    return res, scalar_bytes, ns


def main():
    args = parse()
    if not os.path.exists(os.path.join(ROOT, "viterbidecodercpp_amd", "libvit_hip.so")):
        # clean checkout: compile the HIP extension first (there is no other decode path to fall back to)
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            while not os.path.exists(os.path.join(ROOT, "viterbidecodercpp_amd", "libvit_hip.so")):
                time.sleep(1.0)
            time.sleep(2.0)
    import numpy as np
    import torch
    import torch.distributed as dist

    from viterbidecodercpp_amd import (COMMON_CODES, BatchDecoder, ViterbiBranchTable, ViterbiDecoder_Config, _lib,
                                       dist as vdist, get_decoding_config, pack_blob, synth)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where collective payloads live

    code = COMMON_CODES[args.code]
    pc = get_decoding_config(args.decode_type, code.R)
    F, L = args.frames, args.bits
    S = L + code.K - 1
    W = code.decision_words

    # ---- the shared branch table: built on rank 0, broadcast as one small blob (RCCL over xGMI), rebuilt everywhere ----
    blob = None
    if rank == 0:
        table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
        config = ViterbiDecoder_Config.from_decoder_config(pc)
        blob = pack_blob(table, config)
    blob = vdist.broadcast_blob(blob, src=0, device=coll_dev)
    plan = {"auto": _lib.PLAN_AUTO, "lds": _lib.PLAN_LDS, "reg": _lib.PLAN_REG, "lds2": _lib.PLAN_LDS2}[args.plan]
    dec = BatchDecoder(device=local_rank, blob=blob, plan=plan)

    # ---- synthetic frames, generated directly in HBM (not timed) ----
    tx, sym = synth.make_frames_torch(code, pc, F, L, args.ebn0, seed=1 + rank, device=dev)
    out = torch.empty((F, L // 8), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    # Two decision workspaces and two HIP streams: update() is VALU-issue bound, chainback() is a latency/HBM-bound
    # bit chase, so step i's chainback runs beside step i+1's update.  Every step still does all of its work on the same
    # resident batch; only the schedule overlaps.
    NWS = args.pipeline
    wss = [dec.new_workspace(F, L) for _ in range(NWS)]
    s_upd = torch.cuda.current_stream(dev)
    cb_prio = int(os.environ.get("VIT_BENCH_CB_PRIORITY", "-1"))  # high priority: the short bit chase gets out of the way of the update
    s_cb = torch.cuda.Stream(device=dev, priority=cb_prio) if NWS == 2 else s_upd
    cb_done = [None] * NWS

    def one_step(k, ev=None):
        ws = wss[k % NWS]
        if cb_done[k % NWS] is not None:
            s_upd.wait_event(cb_done[k % NWS])          # the chainback that last read this workspace has finished
        if ev is not None:
            ev[0].record(s_upd)
        dec.update(sym, L, want_metrics=False, workspace=ws)
        upd_done = torch.cuda.Event(enable_timing=ev is not None) if ev is None else ev[1]
        upd_done.record(s_upd)
        s_cb.wait_event(upd_done)
        with torch.cuda.stream(s_cb):
            if ev is not None and NWS == 2:
                ev[3].record(s_cb)
            dec.chainback(F, L, out=out, workspace=ws)
            done = torch.cuda.Event(enable_timing=ev is not None) if ev is None else ev[2]
            done.record(s_cb)
        cb_done[k % NWS] = done

    for k in range(args.warmup):
        one_step(k)
    torch.cuda.synchronize()

    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(args.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(args.warmup + k, evs[k])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    upd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    cb_ms = float(np.mean([(e[3] if NWS == 2 else e[1]).elapsed_time(e[2]) for e in evs]))

    # ---- size-independent parity property at full size: decoded bits vs transmitted bits ----
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
    bit_errors = int(lut[torch.bitwise_xor(out, tx).long()].sum().item())
    ber = bit_errors / float(F * L)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_bits = float(F) * L * world * args.steps
    value = total_bits / elapsed / 1e6
    sb = pc.soft_bytes
    upd_bytes = F * (S * code.R * sb + S * W * 8)            # symbols read + decision words written
    cb_bytes = F * (L * 8 + L // 8)                          # one decision word read per decoded bit + bytes out
    achieved = upd_bytes / (upd_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"{code.name}|{args.decode_type}|{F}x{L}|{_lib.PLAN_NAMES[dec.plan]}"
            traffic = tj.get(key, {}).get("update_kernel_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    result = {
        "metric": "decoded Mbit/s (= ACS trellis steps/s), update()+chainback(), bit-exact vs scalar reference",
        "value": value, "unit": "Mbit/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16" if pc.error_bytes == 2 else "u8", "data": "synthetic",
        "config": {"workload": f"{code.name} K={code.K} R=1/{code.R} {args.decode_type} "
                               f"({'u16/s16' if sb == 2 else 'u8/s8'}), {F} frames x {L} info bits per GPU, "
                               f"AWGN Eb/N0={args.ebn0} dB", "frames_per_gpu": F, "bits_per_frame": L,
                   "plan": _lib.PLAN_NAMES[dec.plan], "pipeline": f"{NWS} workspace(s), {'2 HIP streams' if NWS == 2 else '1 stream'}",
                   "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
        "per_gpu_Mbit_s": value / world, "Msym_s": value * code.R,
        "update_ms": upd_ms, "chainback_ms": cb_ms,
        "roofline": {"bound": "hbm", "kernel": "update (ACS + decision writeback)", "achieved": achieved,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": upd_bytes,
                     "limiter": "integer VALU issue, not HBM: the kernel's measured HBM bytes equal the algorithmic bytes "
                                "(traffic) and its waves issue VALU back to back (DESIGN.md 4.4, profiles/*_summary.md)"},
        "roofline_end_to_end": {"achieved": (upd_bytes + cb_bytes) / ((upd_ms + cb_ms) * 1e-3) / 1e9, "unit": "GB/s",
                                "bytes_per_info_bit": (upd_bytes + cb_bytes) / float(F * L)},
        "ber": ber,
    }

    if world == 1 and not args.no_cpu_baseline:
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        nf = min(F, max(256, 16 * cores))
        sym_host = sym[:nf].cpu().numpy()
        tx_host = tx[:nf].cpu().numpy()
        base, scalar_bytes, ns = cpu_baseline(args.code, code, pc, args.decode_type, sym_host, tx_host, L, args.cpu_seconds)
        result["cpu_baseline"] = base
        gpu_bytes = out[:ns].cpu().numpy()
        result["parity"] = {"frames_checked_vs_scalar_reference": int(ns),
                            "bit_exact": bool(np.array_equal(gpu_bytes, scalar_bytes))}
        result["speedup_vs_cpu_baseline"] = value / base["value"]
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
