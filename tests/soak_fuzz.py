#!/usr/bin/env python3
"""Randomised soak on the GPU box: tests/test_gpu_fuzz.py's comparison (decision words, metrics, renormalisation sums,
chainback bytes against the oracle) with fresh seeds, for a wall-clock budget.  Every third case also feeds the decoders in
two chunks through the resumed update (vit_hip_update_batch_resume) and must land on the same results.
    python tests/soak_fuzz.py [seconds] [first_seed] [chainback kernel: 1 | 2]   (tests/test_gpu_soak.py runs a 45-second slice under pytest;
the third argument forces one of the two chainback kernels of the K = 7 / 9 codes through vit_hip_chainback_batch_ex)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np


def soak(budget_seconds: float, first_seed: int = 1, progress=None, chainback_kernel=None) -> int:
    """returns the number of (code, width, config) cases checked; raises AssertionError on the first mismatch"""
    import torch
    from oracle import pyoracle
    from viterbidecodercpp_amd import COMMON_CODES, BatchDecoder, Code, ViterbiBranchTable, ViterbiDecoder_Config, _lib
    from tests.test_gpu_fuzz import random_config

    pyoracle.ensure_built()
    oracle = pyoracle.Oracle()
    cases = [(COMMON_CODES[2], _lib.PLAN_REG), (COMMON_CODES[3], _lib.PLAN_REG), (COMMON_CODES[4], _lib.PLAN_REG),
             (COMMON_CODES[5], _lib.PLAN_REG), (COMMON_CODES[6], _lib.PLAN_REG), (COMMON_CODES[0], _lib.PLAN_REG),
             (COMMON_CODES[1], _lib.PLAN_REG), (Code("K11", 11, 2, (0o3345, 0o3613)), _lib.PLAN_LDS2),
             (Code("K10", 10, 3, (0o1117, 0o1365, 0o1633)), _lib.PLAN_LDS2),
             (Code("K12", 12, 3, (0o4335, 0o5723, 0o7221)), _lib.PLAN_LDS2), (COMMON_CODES[7], _lib.PLAN_LDS2),
             (Code("K6", 6, 2, (0o65, 0o57)), _lib.PLAN_LDS), (Code("K16", 16, 2, (46749, 58851)), _lib.PLAN_LDS2),
             # polynomials drawn per seed, PLAN_AUTO: the GENERIC register-plan kernels (K = 3 .. 9; R = 1 .. 4), which read them at run time
             (7, _lib.PLAN_AUTO), (8, _lib.PLAN_AUTO), (9, _lib.PLAN_AUTO), (5, _lib.PLAN_AUTO), (4, _lib.PLAN_AUTO), (6, _lib.PLAN_AUTO), (3, _lib.PLAN_AUTO)]
    t_end = time.time() + budget_seconds
    seed, n = first_seed, 0
    while time.time() < t_end:
        for ci, (code0, plan) in enumerate(cases):
            for width in (2, 1):
                rng = np.random.default_rng(100000 * seed + 10 * ci + width)
                code = code0
                if isinstance(code0, int):
                    K, R = code0, int(rng.integers(1, 5))
                    if K == 6 and R % 2:
                        R += 1                      # (K = 6 at an odd rate has no generic kernels: an 80- to 240-step unrolled block)
                    code = Code(f"K{K} generic", K, R, tuple(int(x) | 1 | (1 << (K - 1)) for x in rng.integers(0, 1 << K, R)))
                trial = int(rng.integers(0, 4))
                cfg = random_config(rng, width, trial)
                sdt = np.int16 if width == 2 else np.int8
                table = ViterbiBranchTable(code.K, code.R, code.G, cfg.high, cfg.low, sdt)
                config = ViterbiDecoder_Config(cfg.max_error, cfg.initial_start_error, cfg.initial_non_start_error,
                                               cfg.renormalisation_threshold, np.uint16 if width == 2 else np.uint8)
                F = int(rng.integers(1, 70 if code.K >= 7 else 200)) if code.K < 11 else int(rng.integers(1, 5))
                L = int(rng.integers(1, 200)) if code.K < 11 else (int(rng.integers(1, 60)) if code.K < 15 else int(rng.integers(1, 24)) if code.K == 15 else int(rng.integers(1, 12)))
                S = L + code.K - 1
                lim = 1 << (8 * width - 1)
                if rng.integers(0, 2):
                    sym = rng.integers(cfg.low, cfg.high + 1, size=(F, S, code.R)).astype(sdt)
                else:
                    sym = rng.integers(-lim, lim, size=(F, S, code.R)).astype(sdt)
                N = code.num_states
                ss = rng.integers(0, N, F).astype(np.int32)
                es = rng.integers(0, N, F).astype(np.int32)
                want = [oracle.decode(code.K, code.R, code.G, cfg, sym[f], L, start_state=int(ss[f]), end_state=int(es[f]))
                        for f in range(F)]
                d_sym = torch.from_numpy(sym).cuda()
                dec = BatchDecoder(table, config, plan=plan)
                if plan == _lib.PLAN_AUTO:
                    assert dec.plan == _lib.PLAN_REG, dec.plan_note
                if n % 3 == 2:          # streamed: two chunks through the resumed update
                    cut = int(rng.integers(1, S)) if S > 1 else 1
                    met = dec.reset_batch(F, start_state=ss)
                    rs = dec.update_resume(d_sym[:, :cut].contiguous(), L, 0, met)
                    if cut < S:
                        rs = rs + dec.update_resume(d_sym.reshape(-1)[cut * code.R:], L, cut, met, n_steps=S - cut,
                                                    symbol_frame_stride=S * code.R)
                else:
                    met, rs = dec.update(d_sym, L, start_state=ss)
                got_dec = dec.export_decisions(F, L).cpu().numpy().view(np.uint64)
                out = dec.chainback(F, L, end_state=es, kernel=chainback_kernel).cpu().numpy()
                met = met.cpu().numpy()
                met = met.view(np.uint16) if width == 2 else met
                for f in range(F):
                    tag = (seed, code.name, code.G, plan, width, f, L, cfg)
                    assert np.array_equal(got_dec[f], want[f]["decisions"]), ("decisions", tag)
                    assert np.array_equal(met[f].astype(np.uint32), want[f]["metrics"]), ("metrics", tag)
                    assert int(rs[f].item()) == want[f]["renorm_sum"], ("renorm", tag)
                    assert np.array_equal(out[f], want[f]["bytes"]), ("bytes", tag)
                n += 1
        seed += 1
        if progress and seed % 200 == 0:
            progress(f"  ... {n} cases, seed {seed}")
    return n


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    kern = int(sys.argv[3]) if len(sys.argv) > 3 else None
    total = soak(budget, seed0, progress=lambda m: print(m, flush=True), chainback_kernel=kern)
    print(f"soak ok: {total} random (code, width, config) cases from seed {seed0}")
