"""Fuzz: random decoder configurations (not just the stock ones) on every kernel plan.  The reference's semantics hold for
ANY ViterbiDecoder_Config and soft-decision range -- thresholds of 0 (renormalise every step) or type-max (never),
asymmetric high/low, max_error unrelated to (high-low)*R, arbitrary initial metrics, symbols far outside [low, high] (wrapping
error_t arithmetic) -- and so must ours, bit for bit."""
import numpy as np
import pytest

from oracle import pyoracle
from viterbidecodercpp_amd import (COMMON_CODES, BatchDecoder, Code, ViterbiBranchTable, ViterbiDecoder_Config, _lib)

pytestmark = pytest.mark.gpu

CASES = [
    # (code, plans that serve it)
    (COMMON_CODES[0], [_lib.PLAN_REG, _lib.PLAN_LDS]),
    (COMMON_CODES[1], [_lib.PLAN_REG, _lib.PLAN_LDS]),
    (COMMON_CODES[2], [_lib.PLAN_REG, _lib.PLAN_LDS]),
    (COMMON_CODES[3], [_lib.PLAN_REG, _lib.PLAN_LDS]),
    (COMMON_CODES[4], [_lib.PLAN_REG]),
    (COMMON_CODES[5], [_lib.PLAN_REG, _lib.PLAN_LDS]),
    (COMMON_CODES[6], [_lib.PLAN_REG]),
    (Code("K10", 10, 2, (0o1167, 0o1545)), [_lib.PLAN_LDS2, _lib.PLAN_LDS]),
    (Code("K11", 11, 2, (0o3345, 0o3613)), [_lib.PLAN_LDS2, _lib.PLAN_LDS]),
    (Code("K10R5", 10, 5, (0o1167, 0o1545, 0o1117, 0o1365, 0o1633)), [_lib.PLAN_LDS2]),      # two table passes per block
    (Code("K11R6", 11, 6, (0o3345, 0o3613, 0o2671, 0o3175, 0o2353, 0o3661)), [_lib.PLAN_LDS2]),   # four
    (COMMON_CODES[7], [_lib.PLAN_LDS2]),
    # K = 16: one 1024-thread workgroup per CU, 128 KiB of metrics updated in place; the LDS plan reads its patterns from L2
    (Code("K16", 16, 2, (46749, 58851)), [_lib.PLAN_LDS2, _lib.PLAN_LDS]),
]


def random_config(rng, width, trial):
    M = (1 << (8 * width)) - 1
    lim = 1 << (8 * width - 1)
    low = int(rng.integers(-lim, lim - 1))
    high = int(rng.integers(low + 1, lim))
    thr = [0, M, int(rng.integers(0, M + 1)), int(rng.integers(M // 2, M + 1))][trial % 4]
    return pyoracle.DecodeConfig(width, width, high, low, int(rng.integers(0, M + 1)), int(rng.integers(0, M + 1)),
                                 int(rng.integers(0, M + 1)), thr)


@pytest.mark.parametrize("width", [2, 1])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_random_configs_all_plans(oracle, case, width):
    import torch

    code, plans = CASES[case]
    rng = np.random.default_rng(1000 * case + width)
    N = code.num_states
    for trial in range(4):
        cfg = random_config(rng, width, trial)
        sdt = np.int16 if width == 2 else np.int8
        table = ViterbiBranchTable(code.K, code.R, code.G, cfg.high, cfg.low, sdt)
        config = ViterbiDecoder_Config(cfg.max_error, cfg.initial_start_error, cfg.initial_non_start_error,
                                       cfg.renormalisation_threshold, np.uint16 if width == 2 else np.uint8)
        F = int(rng.integers(1, 40)) if code.K < 11 else 3
        L = int(rng.integers(1, 30)) * 8 if code.K < 15 else 16 if code.K == 15 else 8
        S = L + code.K - 1
        lim = 1 << (8 * width - 1)
        if trial % 2 == 0:      # in-range symbols
            sym = rng.integers(cfg.low, cfg.high + 1, size=(F, S, code.R)).astype(sdt)
        else:                   # anything the type can hold
            sym = rng.integers(-lim, lim, size=(F, S, code.R)).astype(sdt)
        ss = rng.integers(0, N, F).astype(np.int32)
        es = rng.integers(0, N, F).astype(np.int32)
        want = [oracle.decode(code.K, code.R, code.G, cfg, sym[f], L, start_state=int(ss[f]), end_state=int(es[f]))
                for f in range(F)]
        d_sym = torch.from_numpy(sym).cuda()
        for plan in plans:
            dec = BatchDecoder(table, config, plan=plan)
            assert dec.plan == plan
            met, rs = dec.update(d_sym, L, start_state=ss)
            got_dec = dec.export_decisions(F, L).cpu().numpy().view(np.uint64)
            out = dec.chainback(F, L, end_state=es).cpu().numpy()
            met = met.cpu().numpy()
            met = met.view(np.uint16) if width == 2 else met
            for f in range(F):
                tag = (code.name, plan, trial, f, cfg)
                assert np.array_equal(got_dec[f], want[f]["decisions"]), tag
                assert np.array_equal(met[f].astype(np.uint32), want[f]["metrics"]), tag
                assert int(rs[f].item()) == want[f]["renorm_sum"], tag
                assert np.array_equal(out[f], want[f]["bytes"]), tag
