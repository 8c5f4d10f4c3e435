#!/usr/bin/env python3
"""bench.py -- headline measurement of the Viterbi update()+chainback() hot path on MI355X.

One "step" = one pass of the hot path (reset -> update -> chainback) over one batch of synthetic AWGN frames already
resident in HBM.  Workload at N=1: BASELINE.json configs[1] -- Voyager K=7 R=1/2, u16 error metrics / s16 soft symbols,
65536 frames x 8192 info bits.  With --gpus N the frames of every rank are independent (weak scaling: each GPU decodes its
own 65536 frames); the only collective is the broadcast of the branch-table/config blob from rank 0 at set-up (RCCL).

Launch forms:
  python bench.py [--gpus 1]                       one rank on cuda:0
  torchrun --nproc-per-node N bench.py --gpus N    the driver's form: RANK/LOCAL_RANK/WORLD_SIZE come from the environment
  python bench.py --gpus N                         this process only spawns N fresh children (one rank per GPU) BEFORE
                                                   touching the GPU, waits for them and exits with their status

Prints ONE JSON line on rank 0 (see the contract in the round prompt): value = decoded info Mbit/s over all GPUs.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
N_SIMD = 1024            # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4          # peak engine clock (the measured clock under load is read by vit_hip_shader_clock_mhz in every run)


def issue_rates():
    """measured issue cost of the update kernels' instruction class (packed 16-bit / VOP3), in ns per wave64 instruction per
    SIMD by waves per SIMD: profiles/issue_rates.json, written from profiles/r3_valu_issue_rates.txt (wall-clock timed on the
    card: no clock assumption)"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "issue_rates.json")))
    except (OSError, ValueError):
        return None


# BASELINE.json configs[i] -> (index into COMMON_CODES, decode type, frames per GPU, info bits per frame, Eb/N0 dB)
BASELINE_CONFIGS = {
    0: (2, "SOFT16", 1, 4096, 3.0),
    1: (2, "SOFT16", 65536, 8192, 3.0),
    2: (5, "SOFT16", 65536, 8192, 3.0),
    3: (2, "HARD8", 32768, 8192, 5.0),     # 262144 frames sharded over 8 GPUs: weak scaling, one GPU's share per rank
    4: (7, "SOFT16", 4096, 8192, 3.0),
}


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
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline's timed passes")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for --gpus > 1 (nccl = RCCL; gloo only to exercise the N>1 path on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="test only: every rank uses cuda:0")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the N-rank set-up without a GPU: rendezvous, broadcast of the table blob, the all-reduce that proves "
                         "N ranks hold the same decoder, per-rank pinning -- then rank 0 prints the JSON skeleton of the N-rank record "
                         "(value null) and every rank exits.  With --backend gloo this runs on a CPU-only host.")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1, 2],
                    help="2: double-buffered decision workspaces, chainback of step i on a second HIP stream beside the "
                         "update of step i+1; 1: both kernels back to back on one stream; 0 (default): the rule of "
                         "vit_hip_pipeline_submit -- 2 for the register plan with at most two update waves per SIMD (the "
                         "headline configuration), else 1")
    ap.add_argument("--via", default=None, choices=["pipeline", "python", "host"],
                    help="pipeline (default): the timed loop calls the shipped C API only -- vit_hip_pipeline_submit per step, "
                         "vit_hip_pipeline_sync at the end; the library picks the schedule (--pipeline is ignored).  python: the "
                         "same schedule rebuilt from torch streams around vit_hip_update_batch / vit_hip_chainback_batch (A/B).  host "
                         "(the default of --config 0): ONE decoder object with host-resident state, the header-level drop-in's call "
                         "pattern reset -> update -> chainback per step (vit_hip_update_host / vit_hip_chainback_host: the single-frame "
                         "latency route)")
    ap.add_argument("--config", type=int, default=None, choices=[0, 1, 2, 3, 4],
                    help="BASELINE.json configs[i]: 0 = K7 soft16 1 frame x 4096; 1 = K7 soft16 65536 x 8192 (the default "
                         "workload); 2 = K9 soft16 65536 x 8192; 3 = K7 hard8, 262144 frames over 8 GPUs = 32768 per GPU "
                         "(--gpus 8 --config 3 is the whole run); 4 = K15 soft16 4096 x 8192.  Sets --code/--decode-type/"
                         "--frames/--bits/--ebn0")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="after the contract's timed region: an UNTIMED-for-`value` leg of back-to-back submits lasting at least this "
                         "long with per-batch timing on -> value_sustained, the decimated step series, first/last-100 medians (0: skip)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="the default workload (BASELINE configs[1]) at N=1 is followed by configs[2], [3] (one GPU's share) and [4] "
                         "through the same pipeline API, reported as `configs: [...]` in the same JSON line (the reference runs its "
                         "whole matrix in one invocation: examples/run_benchmark.cpp:168-179); this flag leaves them out")
    ap.add_argument("--parity", action="store_true", help="run the in-run parity check against the scalar reference even with --no-cpu-baseline")
    ap.add_argument("--synth", default="hip", choices=["hip", "torch"],
                    help="frame synthesis (untimed): hip = vit_hip_synth_batch (one HIP kernel), torch = ATen elementwise ops")
    args = ap.parse_args()
    if args.config is not None:
        code, dt, frames, bits, ebn0 = BASELINE_CONFIGS[args.config]
        args.code, args.decode_type, args.frames, args.bits, args.ebn0 = code, dt, frames, bits, ebn0
    if args.via is None:
        args.via = "host" if args.config == 0 else "pipeline"
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh child processes, one rank each.  This parent never
    touches the GPU (torch.cuda.device_count() does not initialise it on this image) and never exec()s."""
    import torch

    n = args.gpus
    have = torch.cuda.device_count()
    if not args.share_gpu and not args.dry_run and have < n:
        sys.exit(f"bench.py: --gpus {n} but only {have} GPU(s) are visible (use --share-gpu only to rehearse the N>1 path)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:          # one rank failed: the others would wait in a collective for ever
                    q.terminate()
        time.sleep(0.05)
    sys.exit(rc if rc >= 0 else 1)


def pin_to_gpu_numa_node(index):
    """Before any GPU call: restrict this rank's CPU affinity to the CPUs local to ITS GPU (the NUMA node the card hangs off), so
    that eight ranks of one node do not all queue their launches from socket 0.  GPU `index` is the index-th GPU node of the KFD
    topology (HIP's order; an integer HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES list is honoured); its PCI function's
    local_cpulist comes from the DRM render node.  Best effort: returns what it did (reported per rank in the JSON line)."""
    info = {"gpu": index, "numa_node": None, "cpus_local": None, "cpus_used": None, "pinned": False}
    try:
        top = "/sys/class/kfd/kfd/topology/nodes"
        gpus = []
        for n in sorted(int(x) for x in os.listdir(top)):
            props = {}
            try:
                for line in open(f"{top}/{n}/properties"):
                    kv = line.split()
                    if len(kv) == 2:
                        props[kv[0]] = kv[1]
            except OSError:
                continue            # a GPU of the host that this container may not open: not one of ours (HIP does not list it either)
            if int(props.get("simd_count", "0")) > 0:
                gpus.append(props)
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis and all(v.strip().isdigit() for v in vis.split(",")):
            gpus = [gpus[int(v)] for v in vis.split(",") if int(v) < len(gpus)]
        minor = gpus[index]["drm_render_minor"]
        devdir = f"/sys/class/drm/renderD{minor}/device"
        info["numa_node"] = int(open(f"{devdir}/numa_node").read())
        local = set()
        for part in open(f"{devdir}/local_cpulist").read().strip().split(","):
            if part:
                a, _, b = part.partition("-")
                local.update(range(int(a), int(b or a) + 1))
        info["cpus_local"] = len(local)
        use = sorted(local & set(os.sched_getaffinity(0)))
        if use:
            os.sched_setaffinity(0, use)
            info["pinned"] = True
        info["cpus_used"] = len(os.sched_getaffinity(0))
    except (OSError, ValueError, IndexError, KeyError) as e:
        info["error"] = f"{type(e).__name__}: {e}"
    return info


def host_cpu_topology():
    """One logical CPU per PHYSICAL core of ONE socket, restricted to what this process may use (affinity mask and cgroup
    CPU quota).  north_star's comparator is the single-socket AVX2 reference."""
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    per_pkg = {}
    for c in allowed:
        base = f"/sys/devices/system/cpu/cpu{c}/topology/"
        try:
            pkg = int(open(base + "physical_package_id").read())
            core = int(open(base + "core_id").read())
        except (OSError, ValueError):
            pkg, core = 0, c
        per_pkg.setdefault(pkg, {}).setdefault(core, c)     # first hardware thread of each core
    sockets = sorted(per_pkg)
    socket0 = sorted(per_pkg[sockets[0]].values())
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    use = socket0
    if quota is not None and quota >= 1 and len(use) > int(quota):
        use = use[:int(quota)]       # more runnable threads than the quota only adds throttling
    return {"model": model, "sockets_visible": len(sockets), "physical_cores_socket0": len(socket0),
            "logical_cpus_allowed": len(allowed), "cgroup_cpu_quota": quota, "cpus": use}


def cpu_baseline(code_id, code, pc, decode_type, sym_dev, L, target_seconds):
    """The reference's own AVX2 strategy (oracle/_ref, kind "reference") on the physical cores of one socket of this host,
    over a bounded sample of the GPU batch; falls back to the C restatement (kind "port", scalar) where _ref was never
    built.  Persistent pinned threads, barrier-to-barrier passes of >= 0.25 s, median (oracle/ref_shim.cpp: bench_one)."""
    import numpy as np
    from oracle import pyoracle

    pyoracle.ensure_built()
    dt = {"SOFT16": pyoracle.SOFT16, "SOFT8": pyoracle.SOFT8, "HARD8": pyoracle.HARD8}[decode_type]
    ocfg = pyoracle.stock_config(dt, code.R)
    topo = host_cpu_topology()
    cpus = topo["cpus"]
    T = len(cpus)
    F = sym_dev.shape[0]
    if not pyoracle.RefLib.available():
        nf = min(F, max(8, 2 * T))
        sym_host = sym_dev[:nf].cpu().numpy()
        oracle = pyoracle.Oracle()
        t0 = time.perf_counter()
        oracle.decode_frames(code.K, code.R, code.G, ocfg, sym_host, L, threads=T)
        t = time.perf_counter() - t0
        return {"value": nf * L / t / 1e6, "unit": "Mbit/s", "cores": T, "kind": "port",
                "strategy": "oracle/viterbi_oracle.c (scalar restatement)",
                "sample": f"{nf} frames x {L} bits of the GPU batch, 1 pass, {T} threads", "host": topo}
    ref = pyoracle.RefLib()
    simd = pyoracle.SIMD_AVX if ref.is_valid(code_id, pc.soft_bytes, pyoracle.SIMD_AVX) else pyoracle.SCALAR
    strategy = {pyoracle.SIMD_AVX: "ViterbiDecoder_AVX_u16/u8", pyoracle.SCALAR: "ViterbiDecoder_Scalar"}
    # one thread first: its rate sizes everything else (and is reported: the per-thread rate of the T-thread run must be
    # within 2x of it, else the T-thread figure is measuring the host's scheduling, not decoding)
    n1 = max(1, min(F, 64 if code.K < 11 else 2))
    sym1 = sym_dev[:n1].cpu().numpy()
    s1, _ = ref.bench(code_id, ocfg, sym1, n1, L, simd=simd, threads=1, passes=1, sweeps=1, cpus=cpus[:1])
    sweeps1 = max(1, int(0.25 / max(s1[0], 1e-6)))
    s1, _ = ref.bench(code_id, ocfg, sym1, n1, L, simd=simd, threads=1, passes=3, sweeps=sweeps1, cpus=cpus[:1])
    rate1 = n1 * L * sweeps1 / float(np.median(s1))             # bit/s, one thread
    # T threads: >= 256 frames per thread (K < 11), each pass >= 0.25 s
    per_thread = 256 if code.K < 11 else 2
    nf = max(T, min(F, per_thread * T))
    sym_host = sym_dev[:nf].cpu().numpy()
    est_pass = (nf / T) * L / rate1
    sweeps = max(1, int(0.25 / est_pass + 0.999))
    passes = max(3, min(41, int(target_seconds / (est_pass * sweeps))))
    sT, _ = ref.bench(code_id, ocfg, sym_host, nf, L, simd=simd, threads=T, passes=passes, sweeps=sweeps, cpus=cpus)
    med = float(np.median(sT))
    rateT = nf * L * sweeps / med
    # single-thread scalar (the parity reference) for context
    ns = max(1, min(F, 16 if code.K < 11 else 1))
    ss, _ = ref.bench(code_id, ocfg, sym_dev[:ns].cpu().numpy(), ns, L, simd=pyoracle.SCALAR, threads=1, passes=3, sweeps=1,
                      cpus=cpus[:1])
    res = {"value": rateT / 1e6, "unit": "Mbit/s", "cores": T, "kind": "reference", "strategy": strategy[simd],
           "sample": f"{nf} frames x {L} bits of the GPU batch ({nf // T} frames per thread), median of {passes} passes of "
                     f"{sweeps} sweep(s) ({med * 1e3:.0f} ms per pass, min {float(np.min(sT)) * 1e3:.0f} / max "
                     f"{float(np.max(sT)) * 1e3:.0f}), {T} persistent threads pinned one per physical core of socket 0, "
                     f"barrier to barrier (no thread creation inside a pass), one decoder per thread, shared branch table",
           "per_thread_Mbit_s": rateT / T / 1e6, "simd_1thread_Mbit_s": rate1 / 1e6,
           # the box's cgroup may grant fewer CPUs than the socket has cores: a labelled linear extrapolation to the whole
           # socket (optimistic for the CPU: all-core clocks are lower than the clocks of a partly loaded socket)
           "single_socket_extrapolated_Mbit_s": rateT / T * topo["physical_cores_socket0"] / 1e6,
           "scalar_1thread_Mbit_s": ns * L / float(np.median(ss)) / 1e6, "host": {k: v for k, v in topo.items() if k != "cpus"}}
    return res


def collective_library(backend):
    """what carried the N > 1 run's collectives: torch's "nccl" backend IS RCCL on ROCm -- its version as the runtime reports it"""
    import torch

    if backend != "nccl":
        return {"name": backend, "version": None}
    try:
        v = torch.cuda.nccl.version()
        v = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:
        v = f"unknown ({type(e).__name__})"
    return {"name": "RCCL (torch.distributed backend \"nccl\")", "version": v, "hip": getattr(torch.version, "hip", None)}


def reference_parity(code_id, code, pc, decode_type, last_decisions, sym_dev, out_dev, F, L, n=512):
    """n frames of the timed batch against the reference SCALAR decoder (oracle/_ref; the C restatement where that is
    absent): chainback bytes AND every decision word."""
    import numpy as np
    import torch
    from oracle import pyoracle

    pyoracle.ensure_built()
    dt = {"SOFT16": pyoracle.SOFT16, "SOFT8": pyoracle.SOFT8, "HARD8": pyoracle.HARD8}[decode_type]
    ocfg = pyoracle.stock_config(dt, code.R)
    n = max(1, min(F, n if code.K < 11 else 8))            # (the scalar reference decodes ~12 Mbit/s at K = 7, ~0.1 at K = 15: a second or so)
    S = L + code.K - 1
    # the workspace layout is tile-major: export the first frames of the last launch only (with sub-batches that launch holds
    # frames f0 .. of the batch)
    f0, got_dec = last_decisions(n)
    got_dec = got_dec.cpu().numpy().view(np.uint64)
    n = got_dec.shape[0]
    got_bytes = out_dev[f0:f0 + n].cpu().numpy()
    sym = sym_dev[f0:f0 + n].cpu().numpy()
    ref = pyoracle.RefLib() if pyoracle.RefLib.available() else None
    oracle = None if ref else pyoracle.Oracle()
    ok_b = ok_d = True
    for f in range(n):
        if ref:
            w = ref.run(code_id, ocfg, sym[f], L, simd=pyoracle.SCALAR)
        else:
            w = oracle.decode(code.K, code.R, code.G, ocfg, sym[f], L)
        ok_b = ok_b and np.array_equal(got_bytes[f], w["bytes"])
        ok_d = ok_d and np.array_equal(got_dec[f], w["decisions"][:S])
    torch.cuda.synchronize()
    return {"frames_checked_vs_scalar_reference": int(n), "first_frame_checked": int(f0), "checker": "reference (oracle/_ref)" if ref else "port (oracle/)",
            "chainback_bytes_bit_exact": bool(ok_b), "decision_words_bit_exact": bool(ok_d), "bit_exact": bool(ok_b and ok_d)}


def run_case(cfg_index, steps, warmup, local_rank, dev, sustain_seconds=1.0):
    """One more BASELINE config after the headline, through the same shipped pipeline API (vit_hip_pipeline_submit per step): its own
    decoder, synthetic batch and workspaces, freed before the next one.  Returns the compact record that goes into `configs: [...]`:
    value (wall clock over the K timed steps), value_steady (median step), value_sustained (median step of a further untimed
    leg), per-launch kernel times from the pipeline's HIP events, the HBM roofline of the update kernel, the vector-issue roofline
    where profiles/traffic.json has an instruction count for this launch, and parity of the last timed launch against the scalar
    reference (bytes and every decision word; 64 frames, 8 at K >= 11)."""
    import gc

    import numpy as np
    import torch

    from viterbidecodercpp_amd import COMMON_CODES, BatchDecoder, DecodePipeline, ViterbiBranchTable, ViterbiDecoder_Config, _lib, get_decoding_config

    t_case = time.perf_counter()
    code_id, decode_type, F, L, ebn0 = BASELINE_CONFIGS[cfg_index]
    code = COMMON_CODES[code_id]
    pc = get_decoding_config(decode_type, code.R)
    S, W, sb = L + code.K - 1, code.decision_words, pc.soft_bytes
    table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    dec = BatchDecoder(table, ViterbiDecoder_Config.from_decoder_config(pc), device=local_rank)
    tx, sym = dec.synth(F, L, ebn0, seed=1, first_frame=0)
    out = torch.empty((F, L // 8), dtype=torch.uint8, device=dev)
    pipe = DecodePipeline(dec, F, L)
    sch = pipe.schedule
    NUPD = int(sch.update_streams)
    F_launch = min(F, int(sch.sub_batch_frames))
    per_step = -(-F // F_launch)
    pipe.set_timing(True)
    for _ in range(warmup):
        pipe.submit(sym, out)
    pipe.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit(sym, out)
    pipe.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t_upd, t_cb, t_done = pipe.timing()
    n_warm = warmup * per_step
    t_done = t_done.astype(np.float64)
    upd_ms, cb_ms = float(np.mean(t_upd[n_warm:])), float(np.mean(t_cb[n_warm:]))
    t_start = t_done[n_warm - 1] if n_warm else 0.0
    step_times = np.diff(np.concatenate([[t_start], t_done[n_warm:][per_step - 1::per_step]]))
    step_median = float(np.median(step_times))
    # parity of the last TIMED launch, before anything else writes the workspaces
    parity = reference_parity(code_id, code, pc, decode_type, pipe.export_last_decisions, sym, out, F, L, n=64)
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
    ber = int(lut[torch.bitwise_xor(out, tx).long()].sum().item()) / float(F * L)
    # sustained leg (untimed for `value`): back-to-back submits for >= sustain_seconds, median completion-to-completion step
    sustained_median = None
    if sustain_seconds > 0:
        chunk = max(8, int(250.0 / max(step_median, 0.05)))
        pipe.set_timing(True)
        t_begin, n = time.perf_counter(), 0
        while time.perf_counter() - t_begin < sustain_seconds or n < 2 * chunk:
            for _ in range(chunk):
                pipe.submit(sym, out)
            n += chunk
            pipe.sync()
        _, _, d = pipe.timing()
        st = np.diff(d.astype(np.float64)[per_step - 1::per_step])
        keep = np.ones(len(st), dtype=bool)
        keep[chunk - 1::chunk] = False                      # the first step after each sync carries the pipeline refill
        sustained_median = float(np.median(st[keep]))
    upd_bytes_launch = F_launch * (S * code.R * sb + S * W * 8)
    upd_bytes = F * (S * code.R * sb + S * W * 8)
    cb_bytes = F * (L * 8 + L // 8)
    achieved = upd_bytes_launch / (upd_ms * 1e-3) / 1e9
    step_ms = elapsed / steps * 1e3
    traffic = valu_insts = traffic_src = None
    key = f"{code.name}|{decode_type}|{F_launch}x{L}|{_lib.PLAN_NAMES[dec.plan]}"
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key, {})
        traffic, valu_insts = tj.get("update_kernel_hbm_bytes_per_launch"), tj.get("update_kernel_valu_insts_per_launch")
        traffic_src = f"{tj.get('source')} (builder's rocprofv3 --pmc run of `bench.py --config {cfg_index}`; not measured by this run)" if tj else None
    except (OSError, ValueError):
        pass
    rec = {
        "baseline_config": cfg_index,
        "config": {"workload": f"{code.name} K={code.K} R=1/{code.R} {decode_type} ({'u16/s16' if sb == 2 else 'u8/s8'}), {F} frames x {L} info bits"
                               + (" (one GPU's share of the 8-GPU run of 262144 frames)" if cfg_index == 3 else "") + f", AWGN Eb/N0={ebn0} dB",
                   "frames_per_gpu": F, "bits_per_frame": L, "plan": _lib.PLAN_NAMES[dec.plan], "via": "pipeline",
                   "pipeline": f"{int(sch.workspaces)} workspaces, {NUPD} update stream(s), chainback "
                               f"{'beside the next update' if sch.chainback_overlapped else 'back to back'}"
                               + (f", {per_step} sub-batches of {F_launch} frames" if per_step > 1 else "")},
        "dtype": "u16" if pc.error_bytes == 2 else "u8", "unit": "Mbit/s", "steps": steps, "warmup": warmup,
        "value": float(F) * L * steps / elapsed / 1e6, "value_steady": float(F) * L / (step_median * 1e-3) / 1e6,
        "value_sustained": (float(F) * L / (sustained_median * 1e-3) / 1e6) if sustained_median else None,
        "ms_per_step": step_ms, "ms_per_step_median": step_median, "ms_per_step_series": [round(float(x), 3) for x in step_times[:64]],
        "update_ms": upd_ms, "chainback_ms": cb_ms, "update_launches_in_flight": NUPD,
        "roofline": {"bound": "hbm", "kernel": "update (ACS + decision writeback)", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": upd_bytes_launch, "frames_per_launch": F_launch, "launches_in_flight": NUPD,
                     "achieved_all_launches_over_wall_clock": upd_bytes * steps / elapsed / 1e9},
        "roofline_end_to_end": {"bound": "hbm", "achieved": (upd_bytes + cb_bytes) / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                "frac": (upd_bytes + cb_bytes) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        "state_updates_per_s": float(F_launch) * S * code.num_states / (upd_ms * 1e-3),
        "parity": parity, "ber": ber,
    }
    rates = issue_rates()
    if valu_insts and rates:
        ns = rates["packed16_ns_per_wave_instr_per_simd"]
        dec._handle.refresh()
        waves = (-(-F_launch // dec._handle.info.workspace_tile_frames) / float(N_SIMD) * NUPD) if dec.plan == _lib.PLAN_REG else None
        wkey = "4" if waves is None else str(int(min(4, max(1, -(-waves // 1)))))
        ach = valu_insts / (upd_ms * 1e-3) / 1e9 * NUPD
        rec["roofline_valu"] = {"bound": "valu", "kernel": "update", "achieved": ach, "peak": N_SIMD / ns[wkey], "unit": "G wave-instr/s",
                                "frac": ach / (N_SIMD / ns[wkey]), "peak_spec": N_SIMD * CLOCK_GHZ / 2.0, "frac_of_spec": ach / (N_SIMD * CLOCK_GHZ / 2.0),
                                "valu_insts_per_launch": valu_insts, "launches_in_flight": NUPD, "update_waves_per_simd": waves,
                                "peak_source": rates["source"], "insts_source": traffic_src}
    pipe.close()
    del pipe, dec, sym, tx, out
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    rec["seconds_spent"] = time.perf_counter() - t_case
    return rec


def dry_run(args, world, rank, affinity):
    """--dry-run: everything of the N-rank path that needs no GPU -- process group, blob broadcast, the all-reduce proof, the
    per-rank gather -- and the skeleton of the record the real run prints (tests/test_dist_cpu.py drives it with 8 gloo ranks)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from viterbidecodercpp_amd import COMMON_CODES, ViterbiBranchTable, ViterbiDecoder_Config, dist as vdist, get_decoding_config, pack_blob

    if args.backend != "gloo":
        sys.exit("bench.py --dry-run rehearses on the CPU: use --backend gloo")
    if world > 1:
        dist.init_process_group("gloo")
    code = COMMON_CODES[args.code]
    pc = get_decoding_config(args.decode_type, code.R)
    blob = None
    if rank == 0:
        table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
        blob = pack_blob(table, ViterbiDecoder_Config.from_decoder_config(pc))
    blob = vdist.broadcast_blob(blob, src=0, device=torch.device("cpu"))
    blob_sum = int(np.frombuffer(blob, dtype=np.uint8).astype(np.int64).sum())
    ranks_seen, blob_sum_all = 1, blob_sum
    per_rank_info = None
    if world > 1:
        chk = torch.tensor([1, blob_sum], dtype=torch.int64)
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        ranks_seen, blob_sum_all = int(chk[0].item()), int(chk[1].item())
        mine = torch.tensor([float(rank), float(affinity["numa_node"]) if affinity and affinity["numa_node"] is not None else -1.0,
                             float(affinity["cpus_used"] or 0) if affinity else 0.0, 1.0 if affinity and affinity["pinned"] else 0.0], dtype=torch.float64)
        allr = [torch.zeros(4, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_info = [{"rank": int(t[0].item()), "numa_node": int(t[1].item()), "cpus": int(t[2].item()),
                          "pinned_to_gpu_local_cpus": bool(t[3].item() > 0)} for t in allr]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        F, L = args.frames, args.bits
        print(json.dumps({
            "dry_run": True, "metric": "decoded Mbit/s (= ACS trellis steps/s), update()+chainback(), bit-exact vs scalar reference",
            "value": None, "unit": "Mbit/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16" if pc.error_bytes == 2 else "u8", "data": "synthetic",
            "config": {"workload": f"{code.name} K={code.K} R=1/{code.R} {args.decode_type}, {F} frames x {L} info bits per GPU",
                       "frames_per_gpu": F, "bits_per_frame": L, "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
            "global_frames": F * world, "frame_ranges": [[r * F, (r + 1) * F] for r in range(world)],
            "ranks": {"world_size": world, "ranks_in_blob_allreduce": ranks_seen, "backend": args.backend if world > 1 else None,
                      "blob_bytes": len(blob), "blob_checksum_identical_on_all_ranks": blob_sum_all == blob_sum * world, "per_rank": per_rank_info}}))


def host_route(args):
    print(json.dumps(host_route_result(args)))


def host_route_result(args):
    """BASELINE configs[0]: one frame through the drop-in's own call pattern (examples/run_simple.cpp:67-80) -- a single decoder
    object with the reference's public state; update() runs ONE GPU launch that also chains the completed frame back
    (vit_hip_update_host_lazy: symbols read from host-mapped memory, rows kept on the device, results polled), chainback() then
    copies the bytes out.  A step = reset -> update(whole frame) -> chainback; the host-pointer boundary is part of the route (it IS
    the product here), so this line is a latency figure, not the throughput headline."""
    import numpy as np
    from oracle import pyoracle
    from viterbidecodercpp_amd import (COMMON_CODES, ViterbiBranchTable, ViterbiDecoder_Config, ViterbiDecoder_Core, ViterbiDecoder_HIP,
                                       get_decoding_config, synth)

    code = COMMON_CODES[args.code]
    pc = get_decoding_config(args.decode_type, code.R)
    L = args.bits
    table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    vitdec = ViterbiDecoder_Core(table, ViterbiDecoder_Config.from_decoder_config(pc))
    tx, sym = synth.make_frames_numpy(code, pc, 1, L, args.ebn0, seed=1)
    flat = sym[0].reshape(-1)
    vitdec.set_traceback_length(L)
    t_upd, t_cb = [], []

    def one():
        vitdec.reset()
        t0 = time.perf_counter()
        acc = ViterbiDecoder_HIP.update(vitdec, flat)
        t1 = time.perf_counter()
        out = vitdec.chainback(L)
        t2 = time.perf_counter()
        t_upd.append(t1 - t0)
        t_cb.append(t2 - t1)
        return acc, out

    for _ in range(args.warmup):
        one()
    del t_upd[:], t_cb[:]
    # the interpreter's cyclic collector runs on allocation counts: with torch imported a full collection takes ~40 ms and lands,
    # deterministically, in the 186th timed step (one 1.4 ms young-generation pass in the 57th) -- 130 frame times charged to one
    # frame.  It is the host's housekeeping, not the route's: collected once here, off inside the timed loop
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        acc, out = one()
    elapsed = time.perf_counter() - t0
    gc.enable()
    S = L + code.K - 1
    dt = {"SOFT16": pyoracle.SOFT16, "SOFT8": pyoracle.SOFT8, "HARD8": pyoracle.HARD8}[args.decode_type]
    ocfg = pyoracle.stock_config(dt, code.R)
    pyoracle.ensure_built()
    ref = pyoracle.RefLib() if pyoracle.RefLib.available() else None
    want = ref.run(args.code, ocfg, sym[0], L, simd=pyoracle.SCALAR) if ref else pyoracle.Oracle().decode(code.K, code.R, code.G, ocfg, sym[0], L)
    ok_b = bool(np.array_equal(out, want["bytes"]))
    ok_d = bool(np.array_equal(np.asarray(vitdec.m_decisions).reshape(S, -1), np.asarray(want["decisions"])[:S].reshape(S, -1)))
    step_ms = elapsed / args.steps * 1e3
    sb = pc.soft_bytes
    frame_bytes = S * code.R * sb + S * 8 + L * 8 + L // 8          # symbols in, decision rows out, one word per step back, bytes out
    result = {
        "metric": "decoded Mbit/s (= ACS trellis steps/s), update()+chainback(), bit-exact vs scalar reference",
        "value": L * args.steps / elapsed / 1e6, "unit": "Mbit/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16" if pc.error_bytes == 2 else "u8", "data": "synthetic",
        "config": {"workload": f"{code.name} K={code.K} R=1/{code.R} {args.decode_type}, ONE frame of {L} info bits through the header-level drop-in "
                               f"(ViterbiDecoder_HIP::update + ViterbiDecoder_Core::chainback, host-resident state), AWGN Eb/N0={args.ebn0} dB",
                   "frames_per_gpu": 1, "bits_per_frame": L, "via": "host", "plan": ("in-place trellis on one wavefront + two helper wavefronts (csrc/kernels_one.hpp: one_frame7_kernel)" if code.K == 7 and code.R <= 4
                            else "one wavefront, lane == state (csrc/kernels_one.hpp: one_frame_kernel)" if code.K <= 7 else "lds")
                           + "; update + the chainback behind it in ONE launch, rows kept on the device"},
        "update_ms": float(np.median(t_upd)) * 1e3, "chainback_ms": float(np.median(t_cb)) * 1e3,
        "ms_per_step_median": float(np.median(np.asarray(t_upd) + np.asarray(t_cb))) * 1e3,
        # the spread of the two calls themselves (a step's reset() and the loop around them are the rest of ms_per_step)
        "update_ms_mean": float(np.mean(t_upd)) * 1e3, "update_ms_max": float(np.max(t_upd)) * 1e3,
        "update_ms_p99": float(np.percentile(t_upd, 99)) * 1e3, "update_ms_series_head": [round(float(x) * 1e3, 4) for x in t_upd[:32]],
        "slow_update_calls_over_2x_median": int(np.sum(np.asarray(t_upd) > 2 * np.median(t_upd))),
        "slow_update_calls": [(int(i), round(float(t_upd[i]) * 1e3, 3)) for i in np.nonzero(np.asarray(t_upd) > 2 * np.median(t_upd))[0][:16]],
        "ns_per_trellis_step_update": float(np.median(t_upd)) / S * 1e9,
        # a dependent chain of S steps on one wavefront: the HBM roofline does not bound it; quoted for the record's shape only
        "roofline": {"bound": "hbm", "kernel": "single-frame update (latency route)", "achieved": frame_bytes / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": frame_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                     "note": "latency-bound: one frame is a chain of L + K - 1 dependent steps on ONE wavefront; the figure of merit is ns per step"},
        "parity": {"frames_checked_vs_scalar_reference": 1, "checker": "reference (oracle/_ref)" if ref else "port (oracle/)",
                   "chainback_bytes_bit_exact": ok_b, "decision_words_bit_exact": ok_d, "bit_exact": ok_b and ok_d},
        "ber": float(np.unpackbits(out ^ tx[0]).mean()),
    }
    if ref and not args.no_cpu_baseline:
        topo = host_cpu_topology()
        res = {}
        for name, simd in (("scalar", pyoracle.SCALAR), ("avx", pyoracle.SIMD_AVX)):
            if simd != pyoracle.SCALAR and not ref.is_valid(args.code, pc.soft_bytes, simd):
                continue
            s1, _ = ref.bench(args.code, ocfg, sym, 1, L, simd=simd, threads=1, passes=1, sweeps=1, cpus=topo["cpus"][:1])
            sweeps = max(1, int(0.2 / max(s1[0], 1e-6)))
            s1, _ = ref.bench(args.code, ocfg, sym, 1, L, simd=simd, threads=1, passes=5, sweeps=sweeps, cpus=topo["cpus"][:1])
            res[name] = L * sweeps / float(np.median(s1)) / 1e6
        result["cpu_baseline"] = {"value": res.get("scalar"), "unit": "Mbit/s", "cores": 1, "kind": "reference", "strategy": "ViterbiDecoder_Scalar",
                                  "sample": f"the same frame, reset -> update -> chainback on ONE core, median of 5 passes of >= 0.2 s",
                                  "simd_1thread_Mbit_s": res.get("avx"), "ms_per_frame_scalar": L / res["scalar"] / 1e3 if res.get("scalar") else None,
                                  "ms_per_frame_avx": L / res["avx"] / 1e3 if res.get("avx") else None, "host": {k: v for k, v in topo.items() if k != "cpus"}}
        result["speedup_vs_cpu_baseline"] = result["value"] / res["scalar"] if res.get("scalar") else None
    return result


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)          # never returns
    if args.via == "host":
        if args.gpus != 1 or args.frames != 1:
            sys.exit("bench.py --via host is the single-frame, single-GPU latency route (--config 0)")
        import torch  # noqa: F401  (one HIP runtime per process: torch's)
        return host_route(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world} from the launcher")
    rank = int(os.environ.get("RANK", "0"))

    # clean checkout: local rank 0 compiles the HIP extension first (there is no other decode path to fall back to).  The other
    # local ranks wait for a MARKER that rank 0 writes after the build has RETURNED (the .so exists long before the linker has
    # finished writing it); the marker is keyed by this launch's rendezvous port, so a stale one from another run is never taken
    lib_path = os.path.join(ROOT, "viterbidecodercpp_amd", "libvit_hip.so")
    # torchrun's defaults (port 29500, run id "none") repeat from run to run: a marker left by an EARLIER launch must never satisfy
    # this one's wait.  The marker therefore carries rank 0's pid and start time, waiters accept it only while that very process is
    # alive and younger than the marker, and rank 0 removes it at exit.
    marker = os.path.join("/tmp", f"vit_hip_built.{os.getuid()}.{os.environ.get('MASTER_PORT', '0')}.{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}")

    def proc_start(pid):
        try:
            return open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[19]     # starttime, in clock ticks since boot
        except (OSError, IndexError):
            return None

    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        if world > 1:
            try:
                os.unlink(marker)               # a stale one from an earlier run with the same port / run id
            except OSError:
                pass
        if not os.path.exists(lib_path) and not args.dry_run:      # --dry-run rehearses on a CPU-only host: nothing to build there
            import __graft_entry__
            __graft_entry__.build()
        if world > 1:
            import atexit
            tmp = f"{marker}.{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(f"{os.getpid()} {proc_start(os.getpid())}")
            os.rename(tmp, marker)
            atexit.register(lambda: os.path.exists(marker) and os.unlink(marker))
    elif (not os.path.exists(lib_path) and not args.dry_run) or world > 1:
        t_wait = time.time()
        while True:
            try:
                pid, start = open(marker).read().split()
                if proc_start(int(pid)) == start:
                    break                       # written by a rank 0 that is still running: this launch's
            except (OSError, ValueError):
                pass
            if time.time() - t_wait > 1800:
                sys.exit("bench.py: local rank 0 never finished building libvit_hip.so")
            time.sleep(0.2)
    import numpy as np
    import torch
    import torch.distributed as dist

    from viterbidecodercpp_amd import (COMMON_CODES, BatchDecoder, ViterbiBranchTable, ViterbiDecoder_Config, _lib,
                                       dist as vdist, get_decoding_config, pack_blob, synth)

    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    affinity = pin_to_gpu_numa_node(local_rank) if world > 1 else None      # before the first GPU call of this rank
    if args.dry_run:
        return dry_run(args, world, rank, affinity)
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
    # proof that `world` ranks hold the same decoder: sum over ranks of a checksum of the received blob
    blob_sum = int(np.frombuffer(blob, dtype=np.uint8).astype(np.int64).sum())
    ranks_seen, blob_sum_all = 1, blob_sum
    if world > 1:
        chk = torch.tensor([1, blob_sum], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        ranks_seen, blob_sum_all = int(chk[0].item()), int(chk[1].item())

    # ---- synthetic frames, generated directly in HBM (not timed) ----
    if args.synth == "hip":
        tx, sym = dec.synth(F, L, args.ebn0, seed=1, first_frame=rank * F)    # rank r owns frames [r*F, (r+1)*F) of one global batch
    else:
        tx, sym = synth.make_frames_torch(code, pc, F, L, args.ebn0, seed=1 + rank, device=dev)
    out = torch.empty((F, L // 8), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    lib = _lib.load()
    import ctypes as C

    def shader_clock(light=False):
        mhz, cyc = C.c_double(0), C.c_double(0)
        _lib.check(lib.vit_hip_shader_clock_mhz(local_rank, C.byref(mhz), None if light else C.byref(cyc)))
        return float(mhz.value), (None if light else float(cyc.value))

    simds = 4 * torch.cuda.get_device_properties(dev).multi_processor_count
    dec._handle.refresh()
    tile_frames = dec._handle.info.workspace_tile_frames
    if args.via == "pipeline":
        # ---- the shipped API: one vit_hip_pipeline_submit per step, one vit_hip_pipeline_sync at the end.  The library owns
        # the workspaces, the streams and the schedule (include/vit_hip.h); the per-kernel durations come from its own HIP
        # events on the streams the kernels run on (vit_hip_pipeline_set_timing).
        from viterbidecodercpp_amd import DecodePipeline
        pipe = DecodePipeline(dec, F, L)
        sch = pipe.schedule
        NWS, NUPD = int(sch.workspaces), int(sch.update_streams)
        sched_desc = (f"vit_hip_pipeline: {NWS} workspaces, {NUPD} update stream(s), chainback "
                      f"{'on its own stream beside the next update' if sch.chainback_overlapped else 'back to back on the update stream'}")
        # The clock probe (a chip full of packed adds, no memory traffic) runs BEFORE the warm-up, not between warm-up and timed
        # region: nothing but the synchronisation the contract asks for separates the W warm-up steps from the K timed ones, so the
        # card enters the timed region in the power state the decode itself put it in (with the probe in between, the first eight
        # timed steps ramped from 3.64 to 3.35 ms).  Timing records run through both; the timed region's are the last K.
        clock_cold = shader_clock()
        pipe.set_timing(True)
        for _ in range(args.warmup):
            pipe.submit(sym, out)
        pipe.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.submit(sym, out)
        pipe.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed_local = time.perf_counter() - t0
        clock_after = shader_clock()
        t_upd, t_cb, t_done = pipe.timing()
        # one record per SUB-batch (the library may feed a batch to the kernels in several launches: sch.sub_batch_frames)
        F_launch = min(F, int(sch.sub_batch_frames))
        per_step = -(-F // F_launch)
        assert len(t_upd) == (args.warmup + args.steps) * per_step
        n_warm = args.warmup * per_step
        t_upd, t_cb, t_done = t_upd[n_warm:], t_cb[n_warm:], t_done.astype(np.float64)
        upd_ms, cb_ms = float(np.mean(t_upd)), float(np.mean(t_cb))        # per launch
        # completion to completion; the first timed step is measured from the completion of the last warm-up batch (the pipeline
        # was drained there: it carries the fill and the host's synchronisation gap) or, without warm-up, from the first update's start
        t_start = t_done[n_warm - 1] if n_warm else 0.0
        step_times = np.diff(np.concatenate([[t_start], t_done[n_warm:][per_step - 1::per_step]]))
        if per_step > 1:
            sched_desc += f", each batch in {per_step} sub-batches of {F_launch} frames"
        if sch.chainback_wave_priority:
            sched_desc += ", chainback at the higher wave priority"
        if sch.chainback_small_kernel:
            sched_desc += ", small-footprint (LDS-ring) chainback kernel"

        def last_decisions(n):            # decision rows of the LAST timed launch, straight from the pipeline's workspace
            return pipe.export_last_decisions(n)

        def sustained_leg(seconds):
            """>= `seconds` of back-to-back submits (never part of `value`): what a long stream of batches runs at once the card has
            settled into the decode's own power state.  Submits go out in chunks of ~0.25 s so that the host never queues more than
            that ahead; per-batch completion times come from the pipeline's own events."""
            step_ms = float(np.median(step_times))
            chunk = max(8, int(250.0 / max(step_ms, 0.05)))
            pipe.set_timing(True)                     # clears the records of the timed region (already read)
            t_begin = time.perf_counter()
            n = 0
            while time.perf_counter() - t_begin < seconds or n < 2 * chunk:
                for _ in range(chunk):
                    pipe.submit(sym, out)
                n += chunk
                pipe.sync()
            wall = time.perf_counter() - t_begin
            u, c, d = pipe.timing()
            d = d.astype(np.float64)[per_step - 1::per_step]
            steps_s = np.diff(d)
            # each chunk ends in a sync: the first step of the next chunk carries the pipeline refill -- excluded from the medians
            keep = np.ones(len(steps_s), dtype=bool)
            keep[chunk - 1::chunk] = False
            st = steps_s[keep]
            dec = max(1, len(steps_s) // 200)
            med = float(np.median(st))
            return {"seconds": wall, "batches": int(n), "submit_chunk": int(chunk),
                    "ms_per_step_median": med, "ms_per_step_first100_median": float(np.median(st[:100])),
                    "ms_per_step_last100_median": float(np.median(st[-100:])), "ms_per_step_p10": float(np.percentile(st, 10)),
                    "ms_per_step_p90": float(np.percentile(st, 90)),
                    "update_ms_median": float(np.median(u)), "chainback_ms_median": float(np.median(c)),
                    "ms_per_step_series_decimated": [round(float(x), 3) for x in steps_s[::dec][:256]], "series_decimation": int(dec),
                    "value_wall_clock": float(F) * L * n / wall / 1e6}

        def update_kernel_clock():
            """measurement builds only (-DVIT_HIP_CLOCK_STAMPS, VIT_HIP_LIB_PATH): the shader clock the UPDATE WAVES themselves saw
            (s_memtime against s_memrealtime, lane 0 of every wave, entry to exit) over a few more untimed batches -- beside the
            chainback of the previous batch, as in the timed region -- and for one update launched alone"""
            if not hasattr(lib, "vit_hip_experiment_clock_stamps"):
                return None
            lib.vit_hip_experiment_clock_stamps.argtypes = [C.c_void_p]
            tiles = -(-F_launch // tile_frames)
            wall_khz = 100000.0                        # s_memrealtime: the constant 100 MHz reference clock (hipDeviceAttributeWallClockRate)
            stamps = torch.zeros((tiles, 6), dtype=torch.int64, device=dev)

            def read(tag):
                torch.cuda.synchronize()
                a = stamps.cpu().numpy().astype(np.float64)
                ok = (a[:, 3] > a[:, 1])
                a = a[ok]
                mhz = (a[:, 2] - a[:, 0]) / (a[:, 3] - a[:, 1]) * (wall_khz / 1e3)
                xcc = a[:, 4].astype(np.int64) & 15
                t0 = a[:, 1].min()
                return {"what": tag, "waves": int(len(a)), "mhz_median": float(np.median(mhz)), "mhz_p10": float(np.percentile(mhz, 10)),
                        "mhz_p90": float(np.percentile(mhz, 90)),
                        "mhz_median_per_xcd": {int(x): float(np.median(mhz[xcc == x])) for x in sorted(set(xcc.tolist()))},
                        "wave_lifetime_us_median": float(np.median(a[:, 3] - a[:, 1]) / (wall_khz / 1e3)),
                        "kernel_span_us": float((a[:, 3].max() - t0) / (wall_khz / 1e3))}

            res = []
            _lib.check(lib.vit_hip_experiment_clock_stamps(C.c_void_p(stamps.data_ptr())))
            for _ in range(6):                         # the last update of these runs beside the chainback of the one before
                pipe.submit(sym, out)
            pipe.sync()
            res.append(read("update beside the previous batch's chainback (pipeline, 6 batches; stamps of the last update)"))
            dec.update(sym[:F_launch], L, want_metrics=False)
            res.append(read("update alone (one launch, nothing else on the card)"))
            _lib.check(lib.vit_hip_experiment_clock_stamps(None))
            return res

        def clock_under_load():
            # UNTIMED extra batches with the clock probe launched in their middle: the shader clock while the decode kernels
            # (and whatever memory traffic they overlap with) run -- lower than the probe's own before / after figures where a
            # memory-heavy chainback shares the card with the update (K = 9: ~2.25 GHz against 2.39 GHz beside the update alone)
            step_ms = float(np.median(step_times))
            n = int(np.ceil(30.0 / max(step_ms, 0.1))) + 3
            for _ in range(n):
                pipe.submit(sym, out)
            time.sleep(1.5 * step_ms / 1e3)
            t_probe = time.perf_counter()
            c = shader_clock(light=True)                       # one wave per CU: does not load the SIMDs it measures
            t_probe = time.perf_counter() - t_probe
            pipe.sync()
            return c[0], {"probe_ms": t_probe * 1e3, "batches_in_flight_during_probe": n}
    else:
        # ---- A/B: the same schedule from Python.  Decision workspaces and HIP streams as vit_hip_pipeline_create picks them:
        # update() is VALU-issue bound, chainback() is a latency/HBM-bound bit chase, so step i's chainback runs beside step
        # i+1's update; a batch of at most one update wave per SIMD also gets a second update stream.
        NWS = args.pipeline
        NUPD = 1
        if NWS == 0:
            if dec.plan == _lib.PLAN_REG and F <= simds * tile_frames:
                NWS, NUPD = 3, 2
            elif dec.plan == _lib.PLAN_REG and F <= 2 * simds * tile_frames:
                NWS = 2
            else:
                NWS = 1
        wss = [dec.new_workspace(F, L) for _ in range(NWS)]
        cb_prio = int(os.environ.get("VIT_BENCH_CB_PRIORITY", "-1"))  # high priority: the short bit chase gets out of the way of the update
        s_upds = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(NUPD - 1)]
        s_cb = torch.cuda.Stream(device=dev, priority=cb_prio) if NWS >= 2 else s_upds[0]
        cb_done = [None] * NWS
        sched_desc = f"python streams: {NWS} workspace(s), {NUPD} update stream(s), {'separate chainback stream' if NWS >= 2 else '1 stream'}"

        def one_step(k, ev=None):
            ws = wss[k % NWS]
            s_upd = s_upds[k % NUPD]
            if cb_done[k % NWS] is not None:
                s_upd.wait_event(cb_done[k % NWS])          # the chainback that last read this workspace has finished
            with torch.cuda.stream(s_upd):
                if ev is not None:
                    ev[0].record(s_upd)
                dec.update(sym, L, want_metrics=False, workspace=ws)
                upd_done = torch.cuda.Event(enable_timing=ev is not None) if ev is None else ev[1]
                upd_done.record(s_upd)
            s_cb.wait_event(upd_done)
            with torch.cuda.stream(s_cb):
                if ev is not None and NWS >= 2:
                    ev[3].record(s_cb)
                dec.chainback(F, L, out=out, workspace=ws)
                done = torch.cuda.Event(enable_timing=ev is not None) if ev is None else ev[2]
                done.record(s_cb)
            cb_done[k % NWS] = done

        clock_cold = shader_clock()             # no probe between warm-up and timed region (see the pipeline route above)
        evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(args.steps)]
        for k in range(args.warmup):
            one_step(k)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            one_step(args.warmup + k, evs[k])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed_local = time.perf_counter() - t0
        clock_after = shader_clock()
        upd_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
        cb_ms = float(np.mean([(e[3] if NWS >= 2 else e[1]).elapsed_time(e[2]) for e in evs]))
        t_done = np.asarray([evs[0][0].elapsed_time(e[2]) for e in evs], dtype=np.float64)
        step_times = np.diff(np.concatenate([[0.0], t_done]))

        F_launch = F
        clock_under_load = None
        sustained_leg = None
        update_kernel_clock = None

        def last_decisions(n):
            return 0, dec.export_decisions(n, L, workspace=wss[(args.warmup + args.steps - 1) % NWS])

    elapsed = elapsed_local
    step_median_ms = float(np.median(step_times))             # steady state: completion to completion, fill and drain excluded
    per_rank = [float(F) * L * args.steps / elapsed_local / 1e6]
    if world > 1:
        tmax = torch.tensor([elapsed, step_median_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed, step_median_ms = float(tmax[0].item()), float(tmax[1].item())
        rates = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(rates, torch.tensor(per_rank, dtype=torch.float64, device=coll_dev))
        per_rank = [float(r.item()) for r in rates]

    # ---- size-independent parity property at full size: decoded bits vs transmitted bits ----
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=dev)
    bit_errors = int(lut[torch.bitwise_xor(out, tx).long()].sum().item())
    ber = bit_errors / float(F * L)

    # every rank reads its own shader clock under load (untimed extra batches) and says where it was pinned
    clock_load = (None, None)
    if clock_under_load is not None:
        try:
            clock_load = clock_under_load()
        except Exception as e:              # a rank whose probe fails still takes part in the all_gather below
            clock_load = (None, {"error": f"{type(e).__name__}: {e}"})
    sustained = None
    if sustained_leg is not None and args.sustain_seconds > 0:
        try:
            sustained = sustained_leg(args.sustain_seconds)
        except Exception as e:
            sustained = {"error": f"{type(e).__name__}: {e}"}
    kernel_clock = None
    if update_kernel_clock is not None and dec.plan == _lib.PLAN_REG:
        try:
            kernel_clock = update_kernel_clock()
        except Exception as e:
            kernel_clock = [{"error": f"{type(e).__name__}: {e}"}]
    per_rank_info = None
    if world > 1:
        if affinity and affinity.get("error") and rank == 0:
            print(f"bench.py: rank 0 could not pin itself to its GPU's CPUs: {affinity['error']}", file=sys.stderr)
        mine = torch.tensor([clock_load[0] or 0.0, float(affinity["numa_node"]) if affinity and affinity["numa_node"] is not None else -1.0,
                             float(affinity["cpus_used"] or 0) if affinity else 0.0, 1.0 if affinity and affinity["pinned"] else 0.0],
                            dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros(4, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_info = [{"clock_mhz_under_load": float(t[0].item()), "numa_node": int(t[1].item()), "cpus": int(t[2].item()),
                          "pinned_to_gpu_local_cpus": bool(t[3].item() > 0)} for t in allr]
    sustained_median_all = sustained.get("ms_per_step_median") if sustained else None
    if world > 1:
        sm = torch.tensor([sustained_median_all or 0.0], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(sm, op=dist.ReduceOp.MAX)
        sustained_median_all = float(sm[0].item()) or None
    # ---- parity of the last timed launch, on EVERY rank: each one checks frames of ITS OWN shard against the scalar reference
    # (bytes and every decision word); the pass flags and counts are summed over the ranks.  N = 1: 512 frames; N > 1: 64 per rank
    parity = None
    if args.parity or not args.no_cpu_baseline or world > 1:
        from oracle import pyoracle
        if world > 1:
            if int(os.environ.get("LOCAL_RANK", "0")) == 0:
                pyoracle.ensure_built()          # one builder per node; the others wait at the barrier
            dist.barrier()
        try:
            parity = reference_parity(args.code, code, pc, args.decode_type, last_decisions, sym, out, F, L, n=512 if world == 1 else 64)
        except Exception as e:                   # a rank whose checker fails still enters the all_reduce below -- as a failure
            parity = {"frames_checked_vs_scalar_reference": 0, "bit_exact": False, "chainback_bytes_bit_exact": False,
                      "decision_words_bit_exact": False, "error": f"{type(e).__name__}: {e}"}
        if world > 1:
            pv = torch.tensor([1.0 if parity["bit_exact"] else 0.0, float(parity["frames_checked_vs_scalar_reference"]), 1.0,
                               1.0 if parity["chainback_bytes_bit_exact"] else 0.0, 1.0 if parity["decision_words_bit_exact"] else 0.0],
                              dtype=torch.float64, device=coll_dev)
            dist.all_reduce(pv, op=dist.ReduceOp.SUM)
            parity = {"ranks_checked": int(pv[2].item()), "ranks_bit_exact": int(pv[0].item()), "frames_checked": int(pv[1].item()),
                      "frames_checked_vs_scalar_reference": int(pv[1].item()), "frames_per_rank": parity["frames_checked_vs_scalar_reference"],
                      "checker": parity.get("checker"), "chainback_bytes_bit_exact": int(pv[3].item()) == world,
                      "decision_words_bit_exact": int(pv[4].item()) == world, "bit_exact": int(pv[0].item()) == world,
                      "what": "every rank: the first frames of ITS shard of the last timed launch (rank r owns frames [r*F, (r+1)*F) of one global batch)"}
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_bits = float(F) * L * world * args.steps
    value = total_bits / elapsed / 1e6
    step_ms = elapsed / args.steps * 1e3
    sb = pc.soft_bytes
    upd_bytes = F * (S * code.R * sb + S * W * 8)            # symbols read + decision words written, one step
    cb_bytes = F * (L * 8 + L // 8)                          # one decision word read per decoded bit + bytes out
    upd_bytes_launch = F_launch * (S * code.R * sb + S * W * 8)     # ... and one LAUNCH (a step may be several: sub-batches)
    achieved = upd_bytes_launch / (upd_ms * 1e-3) / 1e9
    traffic, traffic_src, valu_insts = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    key = f"{code.name}|{args.decode_type}|{F_launch}x{L}|{_lib.PLAN_NAMES[dec.plan]}"
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath)).get(key, {})
            traffic = tj.get("update_kernel_hbm_bytes_per_launch")
            valu_insts = tj.get("update_kernel_valu_insts_per_launch")
            traffic_src = f"{tj.get('source')} (builder's rocprofv3 --pmc run of this command; not measured by this run)" if tj else None
        except Exception:
            traffic = None

    state_updates = float(F_launch) * S * code.num_states     # add-compare-select results per launch
    result = {
        "metric": "decoded Mbit/s (= ACS trellis steps/s), update()+chainback(), bit-exact vs scalar reference",
        "value": value, "unit": "Mbit/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        # `value` is wall clock over the K timed steps: it carries the pipeline's fill (the first update has no chainback beside
        # it) and drain (the last chainback runs alone) and the card's clock ramp.  `value_steady` prices the same batches at the
        # MEDIAN completion-to-completion step time (max over ranks): what a long stream of batches runs at
        "value_steady": float(F) * L * world / (step_median_ms * 1e-3) / 1e6,
        "ms_per_step": step_ms, "ms_per_step_min": float(np.min(step_times)), "ms_per_step_median": float(np.median(step_times)),
        "ms_per_step_max": float(np.max(step_times)), "ms_per_step_series": [round(float(x), 3) for x in step_times[:64]],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u16" if pc.error_bytes == 2 else "u8", "data": "synthetic",
        "config": {"workload": f"{code.name} K={code.K} R=1/{code.R} {args.decode_type} "
                               f"({'u16/s16' if sb == 2 else 'u8/s8'}), {F} frames x {L} info bits per GPU, "
                               f"AWGN Eb/N0={args.ebn0} dB", "frames_per_gpu": F, "bits_per_frame": L,
                   "plan": _lib.PLAN_NAMES[dec.plan], "via": args.via, "pipeline": sched_desc,
                   "synth": args.synth,
                   "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
        "per_gpu_Mbit_s": value / world, "per_rank_Mbit_s": per_rank, "Msym_s": value * code.R,
        "ranks": {"world_size": world, "ranks_in_blob_allreduce": ranks_seen, "backend": args.backend if world > 1 else None,
                  "collective_library": collective_library(args.backend) if world > 1 else None,
                  "blob_bytes": len(blob), "blob_checksum_identical_on_all_ranks": blob_sum_all == blob_sum * world,
                  "per_rank": per_rank_info},
        "update_ms": upd_ms, "chainback_ms": cb_ms, "update_launches_in_flight": NUPD,
        # the clock the SIMDs sustained under a packed-integer load right before / after the timed region (s_memtime against
        # s_memrealtime around ~2 ms of v_pk_add_u16 on every SIMD: vit_hip_shader_clock_mhz), and the nominal maximum
        # "under_load": a ONE-wave-per-CU probe launched WHILE untimed extra batches run on the pipeline's streams (the
        # batches submitted outlast it: 30 ms of work and more against a probe of a millisecond or two)
        # "before_warmup": the heavy probe on a cold card, before the warm-up (nothing is probed between warm-up and timed region);
        # "update_kernel": measurement builds only -- the clock the update waves themselves read (s_memtime / s_memrealtime stamps)
        "clock_mhz": {"before_warmup": clock_cold[0], "after": clock_after[0],
                      "under_load": clock_load[0], "nominal_max_spec": CLOCK_GHZ * 1e3,
                      "cycles_per_pk_instr_4_waves": [clock_cold[1], clock_after[1]],
                      "under_load_probe": clock_load[1], "update_kernel": kernel_clock},
        # >= --sustain-seconds of back-to-back batches AFTER the timed region (not part of `value`): median step of the whole leg
        # (max over ranks), its first / last hundred steps, the decimated series
        "value_sustained": (float(F) * L * world / (sustained_median_all * 1e-3) / 1e6) if sustained_median_all else None,
        "sustained": sustained,
        "roofline": {"bound": "hbm", "kernel": "update (ACS + decision writeback)", "achieved": achieved,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": upd_bytes_launch, "frames_per_launch": F_launch,
                     # `achieved` is per launch as the contract defines it.  With two update launches in flight (the small-batch
                     # schedule) each one takes about twice as long while the SIMDs do the work of both: the kernel's rate over
                     # the timed region, all launches together, is the second figure
                     "launches_in_flight": NUPD,
                     "achieved_all_launches_over_wall_clock": upd_bytes * args.steps / elapsed_local / 1e9,
                     "tighter_bound": "valu (roofline_valu below): the kernel's HBM bytes equal the algorithmic bytes and its "
                                      "waves issue packed integer VALU back to back (DESIGN.md 4.4)"},
        # wall-clock of a whole step (the kernels overlap on separate streams where the schedule says so: durations do not add)
        "roofline_end_to_end": {"bound": "hbm", "achieved": (upd_bytes + cb_bytes) / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                                "unit": "GB/s", "frac": (upd_bytes + cb_bytes) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                "bytes_per_info_bit": (upd_bytes + cb_bytes) / float(F * L)},
        "state_updates_per_s": state_updates / (upd_ms * 1e-3),
        "ber": ber,
    }
    rates = issue_rates()
    if valu_insts and rates:
        # The ceiling of the update kernel's vector-instruction rate: what a pure stream of packed 16-bit instructions issues
        # at, timed on the card by wall clock (no clock assumed), with as many waves per SIMD as this batch gives the update
        # kernel.  (The guide's 2-clock figure is reached by a handful of 32-bit integer opcodes in homogeneous streams only --
        # profiles/r4_op_rates.txt; with ALL of the kernel's adds in that form it ran 1.5 - 2 % faster, DESIGN.md 4.4 -- so it is
        # not quoted as a peak any more.)
        ns = rates["packed16_ns_per_wave_instr_per_simd"]
        waves = None
        if dec.plan == _lib.PLAN_REG:
            waves = -(-F_launch // tile_frames) / float(N_SIMD) * NUPD
        wkey = "4" if waves is None else str(int(min(4, max(1, -(-waves // 1)))))
        peak_measured = N_SIMD / ns[wkey]                                  # G wave-instructions per second over the chip
        ach = valu_insts / (upd_ms * 1e-3) / 1e9 * NUPD                    # NUPD launches share the SIMDs while each one runs
        result["roofline_valu"] = {
            "bound": "valu", "kernel": "update", "achieved": ach, "peak": peak_measured, "unit": "G wave-instr/s",
            "frac": ach / peak_measured,
            # the measured peak is that of a PACKED 16-bit stream; a handful of 32-bit opcodes the kernels also use (v_bitop3 with
            # VGPR operands, v_xor) issue faster, so it is no strict upper bound for the mix: the guide's 2 clocks per wave64
            # instruction at the nominal clock is, and is quoted beside it
            "peak_spec": N_SIMD * CLOCK_GHZ / 2.0, "frac_of_spec": ach / (N_SIMD * CLOCK_GHZ / 2.0),
            "valu_insts_per_launch": valu_insts, "launches_in_flight": NUPD,
            "valu_insts_per_state_update_pair": valu_insts * 64.0 / (state_updates / 2.0),
            "update_waves_per_simd": waves, "ns_per_packed_instr_per_simd_at_that_occupancy": ns[wkey],
            "shader_clock_mhz_measured_this_run": [clock_cold[0], clock_after[0]],
            "peak_source": rates["source"], "insts_source": traffic_src}

    if parity is not None:
        result["parity"] = parity
    def headline_cpu_baseline():
        # (the CPU leg: 10 - 20 s in which the card idles and drops its clocks -- it runs LAST, behind the other configs' GPU legs)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.code, code, pc, args.decode_type, sym, L, args.cpu_seconds)
            result["speedup_vs_cpu_baseline"] = value / result["cpu_baseline"]["value"]
            if "single_socket_extrapolated_Mbit_s" in result["cpu_baseline"]:
                result["speedup_vs_single_socket_extrapolated"] = value / result["cpu_baseline"]["single_socket_extrapolated_Mbit_s"]
    # ---- the other BASELINE configs, driver-timed in the same line (the reference runs its whole matrix in one invocation:
    # examples/run_benchmark.cpp:168-179).  Only behind the DEFAULT workload at N = 1; the headline's numbers above are complete
    # before any of this runs, and its buffers are freed first.
    default_workload = (args.code, args.decode_type, args.frames, args.bits) == BASELINE_CONFIGS[1][:4] and args.plan == "auto"
    if world == 1 and args.via == "pipeline" and default_workload and not args.no_extra_configs:
        import gc
        pipe.close()
        del pipe, tx, out                   # (the symbols stay for the CPU leg: 2 GB of 288)
        gc.collect()
        torch.cuda.empty_cache()
        extra = []
        for idx in (2, 3, 4):
            try:
                extra.append(run_case(idx, args.steps, args.warmup, local_rank, dev, sustain_seconds=min(1.0, args.sustain_seconds)))
            except Exception as e:
                extra.append({"baseline_config": idx, "error": f"{type(e).__name__}: {e}"})
        # ... and configs[0], the reference's own first example (examples/run_simple.cpp): ONE 4096-bit frame through the drop-in's
        # reset -> update -> chainback (a latency figure; `--config 0` prints the same record alone)
        try:
            import argparse
            t0c = time.perf_counter()
            c0 = BASELINE_CONFIGS[0]
            r0 = host_route_result(argparse.Namespace(code=c0[0], decode_type=c0[1], bits=c0[3], ebn0=c0[4], steps=max(args.steps, 200), warmup=max(args.warmup, 20),
                                                      no_cpu_baseline=args.no_cpu_baseline))
            r0["baseline_config"] = 0
            r0["seconds_spent"] = time.perf_counter() - t0c
            extra.insert(0, r0)
        except Exception as e:
            extra.insert(0, {"baseline_config": 0, "error": f"{type(e).__name__}: {e}"})
        result["configs"] = extra
    headline_cpu_baseline()
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
