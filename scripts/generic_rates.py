#!/usr/bin/env python3
"""Rates of the GENERIC register-plan kernels (polynomials read at run time, RegSpec::GENERIC) beside kernels specialised for a set of
the same (K, R): the stock code where the reference has one, else a set of tests/jit_codes.txt from tests/_jit_cache.
usage: generic_rates.py [frames] [L] > profiles/r6_generic_rates.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("VIT_HIP_CACHE_DIR", os.path.join(ROOT, "tests", "_jit_cache"))
import numpy as np
import torch
from viterbidecodercpp_amd import BatchDecoder, DecodePipeline, ViterbiBranchTable, ViterbiDecoder_Config, _lib, get_decoding_config

F = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8192

# (K, R, polynomials nobody compiled, polynomials with specialised kernels, where those come from)
CASES = [
    (3, 2, (0o7, 0o7), (0o7, 0o5), "stock Basic K=3"),
    (4, 2, (0o13, 0o17), (0o15, 0o17), "package cache (best K=4)"),
    (5, 2, (0o37, 0o21), (0o27, 0o31), "stock Basic K=5"),
    (5, 3, (0o27, 0o31, 0o35), (0o25, 0o33, 0o37), "package cache (best K=5 R=1/3)"),
    (6, 2, (0o73, 0o45), (0o65, 0o57), "package cache (IS-54)"),
    (7, 2, (0o147, 0o135), (0o155, 0o117), "stock Voyager"),
    (7, 3, (0o133, 0o145, 0o175), (0o133, 0o171, 0o165), "stock LTE"),
    (7, 4, (0o117, 0o133, 0o155, 0o171), (0o155, 0o117, 0o123, 0o155), "stock DAB (8 of 16 patterns occur)"),
    (8, 2, (0o247, 0o365), (0o371, 0o247), "tests/_jit_cache"),
    (8, 3, (0o225, 0o331, 0o357), (0o367, 0o331, 0o225), "tests/_jit_cache"),
    (9, 2, (0o515, 0o677), (0o753, 0o561), "stock IS-95A"),
    (9, 3, (0o435, 0o567, 0o715), (0o557, 0o663, 0o711), "package cache (UMTS R = 1/3)"),
    (9, 4, (0o463, 0o535, 0o733, 0o745), (0o765, 0o671, 0o513, 0o473), "stock CDMA 2000"),
]


def rate(K, R, G, dt, want_generic):
    pc = get_decoding_config(dt, R)
    table = ViterbiBranchTable(K, R, G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    dec = BatchDecoder(table, ViterbiDecoder_Config.from_decoder_config(pc))
    if not want_generic and "GENERIC" in dec.plan_note:
        dec.set_plan(_lib.PLAN_REG)              # kernels specialised for the set, from the user cache
    assert dec.plan == _lib.PLAN_REG and ("GENERIC" in dec.plan_note) == want_generic, dec.plan_note
    tx, sym = dec.synth(F, L, 3.0, seed=1)
    out = torch.empty((F, L // 8), dtype=torch.uint8, device="cuda")
    pipe = DecodePipeline(dec, F, L)
    s = pipe.schedule
    steps = 12 if K == 7 else 6
    for _ in range(3):
        pipe.submit(sym, out)
    pipe.sync()
    pipe.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        pipe.submit(sym, out)
    pipe.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    u, c, d = pipe.timing()
    ber = int(dec.count_bit_errors(out, tx).item()) / float(F * L)
    upd = dec.kernel_resources(_lib.KERNEL_UPDATE)
    res = dict(gbit=F * L / ms / 1e6, ms=ms, upd=float(np.median(u)), cb=float(np.median(c)), ber=ber, vgpr=upd["vgpr_alloc"],
               scratch=upd.get("scratch_bytes", 0), sched=f"ws={s.workspaces} upd={s.update_streams} overlapped={s.chainback_overlapped} sub={s.sub_batch_frames}")
    del pipe, out, sym, tx, dec
    torch.cuda.empty_cache()
    return res


print(f"# GENERIC kernels (polynomials at run time) against kernels specialised for a set of the same (K, R); {F} x {L} through vit_hip_pipeline_*, one MI355X")
for K, R, Gg, Gs, origin in CASES:
    for dt in ("SOFT16", "SOFT8"):
        if K == 8 and R == 3 and dt == "SOFT8":
            continue                             # (tests/_jit_cache holds the 16-bit object of this set only)
        g = rate(K, R, Gg, dt, True)
        s = rate(K, R, Gs, dt, False)
        print(f"K{K} R{R} {dt:6s} generic {'/'.join(oct(x) for x in Gg)}: {g['gbit']:7.2f} Gbit/s (update {g['upd']:.3f} ms, {g['vgpr']} VGPRs, {g['sched']}, BER {g['ber']:.1e}) | "
              f"specialised {'/'.join(oct(x) for x in Gs)} [{origin}]: {s['gbit']:7.2f} Gbit/s (update {s['upd']:.3f} ms, {s['vgpr']} VGPRs, {s['sched']}) | ratio {g['gbit'] / s['gbit']:.3f}",
              flush=True)
