#!/usr/bin/env python3
"""Randomised soak of the single-decoder host route (ViterbiDecoder_HIP::update + ViterbiDecoder_Core::chainback on ONE frame; csrc/kernels_one.hpp:
the in-place K = 7 kernel with its two helper wavefronts, the lane == state kernel for the other K <= 7 codes): random polynomials, configurations
(thresholds 0 / a few steps' error / type-max / random), in-range or full-range symbols, start / end states, frame lengths 1 .. 3000 bits, calls cut
into random pieces.  Decision rows, metrics, renormalisation sums, error and bytes against the oracle.
    python tests/soak_host_route.py [seconds] [first_seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np


def soak(budget_seconds, first_seed=1, progress=None):
    from oracle import pyoracle
    from tests.test_gpu_fuzz import random_config
    from viterbidecodercpp_amd import ViterbiBranchTable, ViterbiDecoder_Config, ViterbiDecoder_Core, ViterbiDecoder_HIP

    pyoracle.ensure_built()
    oracle = pyoracle.Oracle()
    t_end = time.time() + budget_seconds
    seed, n = first_seed, 0
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        K = int(rng.choice([7, 7, 7, 7, 5, 3, 6, 4, 2]))
        R = int(rng.integers(1, 5)) if K == 7 else int(rng.integers(1, 7))
        G = tuple(int(g) | 1 | (1 << (K - 1)) for g in rng.integers(0, 1 << K, R))
        width = int(rng.integers(1, 3))
        M = (1 << (8 * width)) - 1
        cfg = random_config(rng, width, int(rng.integers(0, 4)))
        if rng.integers(0, 3) == 0:          # a renormalisation every few steps
            me = (cfg.high - cfg.low) * R & M
            cfg = type(cfg)(width, width, cfg.high, cfg.low, me, 0, min(M, 3 * me), min(M, int(rng.integers(2, 9)) * me + 1))
        sdt, edt = (np.int16, np.uint16) if width == 2 else (np.int8, np.uint8)
        lim = 1 << (8 * width - 1)
        L = int(rng.integers(1, 3000)) if rng.integers(0, 4) else int(rng.integers(1, 120))
        S = L + K - 1
        sym = (rng.integers(cfg.low, cfg.high + 1, size=(S, R)) if rng.integers(0, 2) else rng.integers(-lim, lim, size=(S, R))).astype(sdt)
        N = 1 << (K - 1)
        ss, es = int(rng.integers(0, N)), int(rng.integers(0, N))
        want = oracle.decode(K, R, G, cfg, sym, L, start_state=ss, end_state=es)
        table = ViterbiBranchTable(K, R, G, cfg.high, cfg.low, sdt)
        config = ViterbiDecoder_Config(cfg.max_error, cfg.initial_start_error, cfg.initial_non_start_error, cfg.renormalisation_threshold, edt)
        vitdec = ViterbiDecoder_Core(table, config)
        vitdec.set_traceback_length(L)
        vitdec.reset(ss)
        flat = np.ascontiguousarray(sym.reshape(-1))
        acc, t = 0, 0
        whole = rng.integers(0, 3) == 0
        while t < S:
            n_steps = S - t if whole else min(int(rng.integers(1, 200)), S - t)
            acc += ViterbiDecoder_HIP.update(vitdec, flat[t * R:(t + n_steps) * R])
            t += n_steps
        acc += vitdec.take_unreported_renormalisation()
        tag = (seed, K, R, G, width, L, cfg)
        assert vitdec.get_error(es) == want["error"], ("error", tag)
        acc += vitdec.take_unreported_renormalisation()
        assert acc == want["renorm_sum"], ("renorm", tag)
        assert np.array_equal(vitdec.m_metrics.astype(np.uint32), want["metrics"]), ("metrics", tag)
        assert np.array_equal(np.asarray(vitdec.m_decisions).reshape(S, -1), want["decisions"].reshape(S, -1)), ("decisions", tag)
        assert np.array_equal(vitdec.chainback(L, es), want["bytes"]), ("bytes", tag)
        n += 1
        seed += 1
        if progress and n % 500 == 0:
            progress(f"  ... {n} cases, seed {seed}")
    return n


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    total = soak(budget, seed0, progress=lambda m: print(m, flush=True))
    print(f"host-route soak ok: {total} random single-frame cases from seed {seed0}")
