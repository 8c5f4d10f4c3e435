"""The GENERIC register-plan kernels: polynomials as a run-time argument, as in the reference.

`ViterbiBranchTable(K, R, G, ...)` takes G at run time (include/viterbi/viterbi_branch_table.h:34-55) and any set runs at the compiled
<K, R>'s speed.  The register plan's kernels are specialised per set; for a set that neither the library nor the package cache has
kernels for, `vit_hip_create` loads the GENERIC code object of (K, R) (RegSpec::GENERIC, kernels_reg.hpp: each butterfly fetches its
branch-metric pair through a run-time address formed from the polynomials in the kernel arguments) -- no compiler on the host, no
fall to the compatibility plan.  Everything is checked bit for bit against the oracle.
"""
import os

import numpy as np
import pytest

from viterbidecodercpp_amd import BatchDecoder, _lib, synth
from viterbidecodercpp_amd.codes import Code

from tests.helpers import check_batch_against_oracle, gpu_metrics_to_u32, make_table_config, oracle_cfg

pytestmark = pytest.mark.gpu


def _no_compiler(monkeypatch, tmp_path):
    monkeypatch.setenv("VIT_HIP_HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("VIT_HIP_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("VIT_HIP_JIT", raising=False)


# polynomial sets nobody compiled: arbitrary taps (outer taps set, as every table the reference builds implies), a set that repeats
# a polynomial, one whose lane bits contribute the same pattern, a catastrophic pair (decodes like any other trellis)
SETS = [
    (7, 2, (0o147, 0o135)), (7, 2, (0o165, 0o127)), (7, 2, (0o101, 0o177)), (7, 2, (0o155, 0o155)), (7, 2, (0o163, 0o115)),
    (7, 3, (0o133, 0o145, 0o175)), (7, 3, (0o117, 0o127, 0o155)), (7, 3, (0o171, 0o171, 0o133)),
    (7, 4, (0o135, 0o135, 0o147, 0o163)), (7, 4, (0o117, 0o133, 0o155, 0o171)), (7, 4, (0o101, 0o103, 0o105, 0o111)),
    # K = 8, 9: the sub-chunk fetch (a butterfly's pair arrives one sub-chunk of four butterflies ahead)
    (9, 2, (0o515, 0o677)), (9, 2, (0o401, 0o777)), (9, 3, (0o435, 0o567, 0o715)), (9, 4, (0o463, 0o535, 0o733, 0o745)),
    (8, 2, (0o247, 0o371)), (8, 3, (0o225, 0o331, 0o367)), (8, 4, (0o235, 0o275, 0o313, 0o357)),
    # below K = 7: all states of a frame pair in one lane, the 2^R pairs parked in LDS and read back per butterfly
    (5, 2, (0o37, 0o21)), (5, 3, (0o27, 0o31, 0o35)), (5, 4, (0o25, 0o27, 0o33, 0o37)),
    (6, 2, (0o73, 0o45)), (6, 4, (0o53, 0o67, 0o71, 0o75)), (4, 2, (0o13, 0o17)), (4, 3, (0o11, 0o15, 0o17)), (4, 4, (0o13, 0o15, 0o15, 0o17)),
    (3, 2, (0o7, 0o7)), (3, 3, (0o5, 0o5, 0o7)), (3, 4, (0o5, 0o7, 0o7, 0o7)),
    # rate 1 (no redundancy, but a valid trellis: one symbol per step, two branch patterns)
    (7, 1, (0o165,)), (9, 1, (0o753,)), (5, 1, (0o27,)), (8, 1, (0o371,)),
]


@pytest.mark.parametrize("K,R,G", SETS)
@pytest.mark.parametrize("decode_type", ["SOFT16", "SOFT8", "HARD8"])
def test_generic_kernels_match_oracle(oracle, monkeypatch, tmp_path, K, R, G, decode_type):
    """decision words, final metrics, renormalisation sums and chainback bytes of sets only the generic kernels serve: 70 frames (a
    partial tile and a partial pair), a length that ends inside a decision row and inside the unrolled block"""
    _no_compiler(monkeypatch, tmp_path)
    code = Code(f"K{K}R{R}", K, R, G)
    ebn0 = {"SOFT16": 2.0, "SOFT8": 3.0, "HARD8": 4.0}[decode_type]
    pc, _, _ = make_table_config(code, decode_type)
    F = 70 if K >= 7 else 150                   # (below K = 7 a wavefront holds 128 frames)
    _, sym = synth.make_frames_numpy(code, pc, F, 432, ebn0, seed=sum(G) + R)
    dec = check_batch_against_oracle(oracle, code, decode_type, F, 427, ebn0, seed=0, sym=sym)     # plan=None: PLAN_AUTO
    assert dec.plan == _lib.PLAN_REG, dec.plan_note
    assert "GENERIC" in dec.plan_note and "_0_0_0_0_0_0_" in dec.plan_note and "package cache" in dec.plan_note, dec.plan_note
    assert not any(f.endswith(".hsaco") for f in os.listdir(tmp_path))          # nothing was compiled


@pytest.mark.parametrize("K,R,G,width", [(7, 2, (0o147, 0o135), 2), (7, 2, (0o165, 0o127), 1), (7, 3, (0o133, 0o145, 0o175), 2),
                                         (7, 4, (0o117, 0o133, 0o155, 0o171), 1), (7, 4, (0o135, 0o135, 0o147, 0o163), 2),
                                         (9, 2, (0o515, 0o677), 2), (9, 3, (0o435, 0o567, 0o715), 1), (9, 4, (0o463, 0o535, 0o733, 0o745), 2),
                                         (8, 2, (0o247, 0o371), 1), (8, 3, (0o225, 0o331, 0o367), 2),
                                         (5, 2, (0o37, 0o21), 2), (5, 3, (0o27, 0o31, 0o35), 1), (6, 2, (0o73, 0o45), 2), (6, 4, (0o53, 0o67, 0o71, 0o75), 1),
                                         (4, 3, (0o11, 0o15, 0o17), 2), (3, 2, (0o7, 0o7), 1)])
def test_generic_kernels_resume_and_states(oracle, monkeypatch, tmp_path, K, R, G, width):
    """batched streaming (vit_hip_update_batch_resume enters the unrolled block in the middle: the offset-table rows in flight must
    belong to the steps that follow), non-zero start / end states, and a threshold state 0 crosses every two to six steps -- chunks
    of every length 1..9 in turn, as tests/test_gpu_resume.py::test_chunked_update_with_frequent_renormalisation does for the
    specialised kernels"""
    import torch

    from oracle import pyoracle
    from viterbidecodercpp_amd import ViterbiBranchTable, ViterbiDecoder_Config

    _no_compiler(monkeypatch, tmp_path)
    code = Code(f"K{K}R{R}", K, R, G)
    F, L = (45 if K >= 7 else 150), 150          # (below K = 7 a wavefront holds 128 frames)
    S = L + K - 1
    rng = np.random.default_rng(4300 + R + width)
    if width == 2:
        cfg = pyoracle.DecodeConfig(2, 2, 127, -127, 254 * R, 0, 300, 700)
    else:
        cfg = pyoracle.DecodeConfig(1, 1, 3, -3, 6 * R, 0, 10, 25)
    sdt = np.int16 if width == 2 else np.int8
    table = ViterbiBranchTable(K, R, G, cfg.high, cfg.low, sdt)
    config = ViterbiDecoder_Config(cfg.max_error, cfg.initial_start_error, cfg.initial_non_start_error,
                                   cfg.renormalisation_threshold, np.uint16 if width == 2 else np.uint8)
    sym = rng.integers(cfg.low, cfg.high + 1, size=(F, S, R)).astype(sdt)
    start = rng.integers(0, code.num_states, F).astype(np.int32)
    end = rng.integers(0, code.num_states, F).astype(np.int32)
    want = [oracle.decode(K, R, G, cfg, sym[f], L, start_state=int(start[f]), end_state=int(end[f])) for f in range(F)]
    assert min(w["renorm_sum"] for w in want) > 0, "the configuration must renormalise in every frame"
    d_sym = torch.from_numpy(sym).cuda()
    dec = BatchDecoder(table, config)
    assert dec.plan == _lib.PLAN_REG and "GENERIC" in dec.plan_note, dec.plan_note
    # one call
    met1, rs1 = dec.update(d_sym, L, start_state=start)
    dec1 = dec.export_decisions(F, L).cpu().numpy().view(np.uint64)
    out1 = dec.chainback(F, L, end_state=end).cpu().numpy()
    # chunks
    ws = dec.new_workspace(F, L)
    ws.fill_(0x5A)
    met = dec.reset_batch(F, start_state=start)
    rs = torch.zeros(F, dtype=torch.int64, device="cuda")
    t, k = 0, 0
    while t < S:
        n = min(1 + k % 9, S - t)
        rs += dec.update_resume(d_sym.reshape(-1)[t * R:], L, t, met, n_steps=n, symbol_frame_stride=S * R, workspace=ws)
        t += n
        k += 1
    got_dec = dec.export_decisions(F, L, workspace=ws).cpu().numpy().view(np.uint64)
    out = dec.chainback(F, L, end_state=end, workspace=ws).cpu().numpy()
    m, m1 = gpu_metrics_to_u32(met, width), gpu_metrics_to_u32(met1, width)
    for f in range(F):
        assert np.array_equal(got_dec[f], want[f]["decisions"]) and np.array_equal(dec1[f], want[f]["decisions"]), f
        assert np.array_equal(m[f], want[f]["metrics"]) and np.array_equal(m1[f], want[f]["metrics"]), f
        assert int(rs[f].item()) == want[f]["renorm_sum"] and int(rs1[f].item()) == want[f]["renorm_sum"], f
        assert np.array_equal(out[f], want[f]["bytes"]) and np.array_equal(out1[f], want[f]["bytes"]), f


@pytest.mark.parametrize("K,G,stock_id,floor", [(7, (0o147, 0o135), 2, 0.85), (9, (0o515, 0o677), 5, 0.75)])
def test_generic_kernels_rate(oracle, monkeypatch, tmp_path, K, G, stock_id, floor):
    """the point of the generic kernels: a set nobody compiled runs near the stock code's rate instead of the compatibility plan's
    1/18 (65536 x 8192 through the pipeline; the stock code of the same K -- Voyager, IS-95A -- on the same box as the yardstick)"""
    import time

    import torch

    from viterbidecodercpp_amd import COMMON_CODES, DecodePipeline

    _no_compiler(monkeypatch, tmp_path)
    F, L = 65536, 8192
    rates = {}
    for name, code in (("generic", Code("custom", K, 2, G)), ("stock", COMMON_CODES[stock_id])):
        assert code.K == K and code.R == 2
        pc, table, config = make_table_config(code, "SOFT16")
        dec = BatchDecoder(table, config)
        assert dec.plan == _lib.PLAN_REG, dec.plan_note
        assert ("GENERIC" in dec.plan_note) == (name == "generic"), dec.plan_note
        tx, sym = dec.synth(F, L, 3.0, seed=5)
        out = torch.empty((F, L // 8), dtype=torch.uint8, device="cuda")
        pipe = DecodePipeline(dec, F, L)
        n_warm, n_timed = (4, 12) if K == 7 else (2, 5)
        for _ in range(n_warm):
            pipe.submit(sym, out)
        pipe.sync()
        best = 0.0
        for _ in range(3):                      # the best of three timed loops: one slow step of the box must not decide a ratio
            t0 = time.perf_counter()
            for _ in range(n_timed):
                pipe.submit(sym, out)
            pipe.sync()
            best = max(best, n_timed * F * L / (time.perf_counter() - t0) / 1e9)
        rates[name] = best
        if name == "generic":
            n = 64
            want, _, _ = oracle.decode_frames(K, 2, code.G, oracle_cfg("SOFT16", 2), sym[:n].cpu().numpy(), L, threads=8)
            assert np.array_equal(out[:n].cpu().numpy(), want)
        del pipe, out, sym, tx
        torch.cuda.empty_cache()
    print(f"generic K{K} R2 {oct(G[0])}/{oct(G[1])}: {rates['generic']:.1f} Gbit/s, stock: {rates['stock']:.1f} Gbit/s "
          f"({rates['generic'] / rates['stock']:.3f})")
    assert rates["generic"] >= floor * rates["stock"], rates
