"""GPU parity: the HIP path (through the C ABI) against the oracle on identical inputs -- bit-exact decision words,
final metrics, renormalisation sum and chainback bytes.  Mirrors the reference's test matrix
(examples/run_tests.cpp:118-191: 8 codes x {SOFT16, SOFT8, HARD8}) but on NOISY frames, which the reference never checks."""
import numpy as np
import pytest

from viterbidecodercpp_amd import COMMON_CODES, _lib
from tests.helpers import DECODE_TYPES, check_batch_against_oracle, default_ebn0

pytestmark = pytest.mark.gpu


def _size(code):
    # sizes the oracle finishes in seconds
    if code.K >= 15:
        return 3, 64
    if code.K >= 9:
        return 5, 512
    return 9, 1024


@pytest.mark.parametrize("decode_type", DECODE_TYPES)
@pytest.mark.parametrize("code", COMMON_CODES, ids=lambda c: c.name.replace(" ", "_"))
def test_lds_plan_matrix(oracle, code, decode_type):
    F, L = _size(code)
    check_batch_against_oracle(oracle, code, decode_type, F, L, default_ebn0(code, decode_type), seed=7,
                               plan=_lib.PLAN_LDS)


@pytest.mark.parametrize("decode_type", DECODE_TYPES)
@pytest.mark.parametrize("code", COMMON_CODES, ids=lambda c: c.name.replace(" ", "_"))
def test_auto_plan_matrix(oracle, code, decode_type):
    F, L = _size(code)
    check_batch_against_oracle(oracle, code, decode_type, F + 2, L, default_ebn0(code, decode_type), seed=11)


@pytest.mark.parametrize("decode_type", DECODE_TYPES)
@pytest.mark.parametrize("K,R,G", [(10, 2, (0o1167, 0o1545)), (10, 3, (0o1117, 0o1365, 0o1633)), (11, 2, (0b10011011001, 0b11110110101)), (13, 3, (0o10533, 0o10675, 0o17661)),
                                   (12, 2, (0o4335, 0o5723)), (14, 2, (0o21645, 0o35661)), (15, 6, COMMON_CODES[7].G),
                                   # one-wavefront workgroups build their tables in passes over (step, pattern): one pass at R <= 4, two at
                                   # R = 5, four at R = 6 (the later ones fetch their symbols when they run)
                                   (10, 4, (0o1167, 0o1545, 0o1117, 0o1365)), (10, 5, (0o1167, 0o1545, 0o1117, 0o1365, 0o1633)),
                                   (11, 6, (0o3345, 0o3613, 0o2671, 0o3175, 0o2353, 0o3661)), (11, 5, (0o3345, 0o3613, 0o2671, 0o3175, 0o2353)),
                                   # K = 12: two wavefronts of two steps each -- one pass up to R = 5, two at R = 6
                                   (12, 5, (0o4335, 0o5723, 0o6265, 0o7173, 0o5537)), (12, 6, (0o4335, 0o5723, 0o6265, 0o7173, 0o5537, 0o6747)),
                                   # K = 13: the same with two groups per thread (a permuted table B beside every table)
                                   (13, 6, (0o10533, 0o10675, 0o17661, 0o13271, 0o15353, 0o16475)), (13, 2, (0o10533, 0o17661))])
def test_lds2_plan_large_k(oracle, K, R, G, decode_type):
    """PLAN_LDS2 (packed frame pair in LDS) on K = 10..15 (K = 10: 32 group slots, lanes 32 - 63 mirror them), including
    non-stock polynomials; odd frame count; long enough to renormalise."""
    from viterbidecodercpp_amd import Code
    code = Code(f"K{K}R{R}", K, R, tuple(G))
    dec = check_batch_against_oracle(oracle, code, decode_type, 3, 48 if K >= 14 else 1600 if K == 10 and R <= 3 else 200 if R >= 4 else 128, -2.0 if K >= 14 else 2.0, seed=K,
                                     plan=_lib.PLAN_LDS2)
    assert dec.plan == _lib.PLAN_LDS2
    dec2 = check_batch_against_oracle(oracle, code, decode_type, 2, 40, 1.0, seed=K + 1, plan=_lib.PLAN_LDS)
    assert dec2.plan == _lib.PLAN_LDS


@pytest.mark.parametrize("code_id", [2, 5])
def test_noise_free_round_trip(oracle, code_id):
    """the reference's own test property (run_tests.cpp:184-186): clean frame decodes with 0 bit errors."""
    import torch
    from viterbidecodercpp_amd import BatchDecoder, synth
    from tests.helpers import make_table_config

    code = COMMON_CODES[code_id]
    pc, table, config = make_table_config(code, "SOFT16")
    tx, sym = synth.make_frames_numpy(code, pc, 6, 512, None, seed=3)
    dec = BatchDecoder(table, config)
    out = dec.decode(torch.from_numpy(sym).cuda(), 512).cpu().numpy()
    assert np.array_equal(out, tx)


@pytest.mark.parametrize("plan", [_lib.PLAN_LDS, _lib.PLAN_AUTO])
def test_ragged_lengths_and_states(oracle, plan):
    """L not a multiple of 8 (partial last byte), non-zero start/end states, odd frame counts, partial updates."""
    code = COMMON_CODES[2]
    rng = np.random.default_rng(5)
    for F, L in [(1, 8), (3, 13), (33, 100), (17, 257)]:
        ss = rng.integers(0, 64, F).astype(np.int32)
        es = rng.integers(0, 64, F).astype(np.int32)
        check_batch_against_oracle(oracle, code, "SOFT16", F, ((L + 7) // 8) * 8, 3.0, seed=L, plan=plan,
                                   start_state=ss, end_state=es)
    # partial update: fewer steps than the traceback length allows
    check_batch_against_oracle(oracle, code, "SOFT16", 4, 256, 3.0, seed=9, plan=plan, n_steps=100)


@pytest.mark.parametrize("K,R,G,decode_type,plan,cases", [
    (7, 2, (0o155, 0o117), "SOFT16", _lib.PLAN_REG, [(33, 13), (70, 100), (17, 257), (130, 1001), (5, 4099)]),
    (7, 2, (0o155, 0o117), "HARD8", _lib.PLAN_REG, [(33, 1029), (3, 7)]),
    (9, 2, (0o753, 0o561), "SOFT16", _lib.PLAN_REG, [(33, 13), (35, 257), (130, 1001), (5, 2601)]),
    (9, 4, (0o765, 0o671, 0o513, 0o473), "SOFT8", _lib.PLAN_REG, [(9, 101), (34, 1027)]),
    (5, 2, (0o27, 0o31), "SOFT16", _lib.PLAN_REG, [(130, 13), (70, 1001)]),
    (3, 2, (0o7, 0o5), "SOFT8", _lib.PLAN_REG, [(130, 9), (70, 1003)]),
    (11, 2, (0o3345, 0o3613), "SOFT16", _lib.PLAN_LDS2, [(3, 13), (5, 257), (4, 1001)]),
    (10, 2, (0o1167, 0o1545), "SOFT8", _lib.PLAN_LDS2, [(3, 13), (7, 257), (5, 1001)]),
    (13, 2, (0o10533, 0o17661), "HARD8", _lib.PLAN_LDS2, [(3, 101), (2, 515)]),
    (15, 6, (0o42631, 0o47245, 0o56507, 0o73363, 0o77267, 0o64537), "SOFT16", _lib.PLAN_LDS2, [(3, 13), (2, 257), (3, 1001)]),
    (16, 2, (46749, 58851), "SOFT16", _lib.PLAN_LDS2, [(2, 101)]),
    (6, 2, (0o65, 0o57), "SOFT16", _lib.PLAN_LDS, [(3, 13), (5, 257)]),
])
def test_ragged_bit_lengths_on_every_plan(oracle, K, R, G, decode_type, plan, cases):
    """total_bits % 8 != 0 on the BATCHED route of every kernel plan (pinned by one K = 7 fixture only): the partial last
    byte is filled from the end state as ViterbiTracebackBuffer does (core.h:87-153), per-frame start / end states, frame counts that
    leave partial tiles, lengths on both sides of the chainback kernels' 32- / 1024-step iterations.  Noisy symbols of a frame
    generated at the next whole-byte length, cut to L + K - 1 steps."""
    from viterbidecodercpp_amd import Code, synth
    from tests.helpers import make_table_config

    code = Code(f"K{K}R{R}", K, R, tuple(G))
    pc, _, _ = make_table_config(code, decode_type)
    rng = np.random.default_rng(K * 100 + R)
    for F, L in cases:
        assert L % 8 != 0
        L8 = (L + 7) // 8 * 8
        _, sym = synth.make_frames_numpy(code, pc, F, L8, 2.0, seed=L + F)
        sym = np.ascontiguousarray(sym[:, :L + K - 1])
        ss = rng.integers(0, code.num_states, F).astype(np.int32)
        es = rng.integers(0, code.num_states, F).astype(np.int32)
        dec = check_batch_against_oracle(oracle, code, decode_type, F, L, None, seed=0, plan=plan, sym=sym, start_state=ss, end_state=es)
        assert dec.plan == plan


@pytest.mark.parametrize("plan", [_lib.PLAN_LDS, _lib.PLAN_AUTO])
@pytest.mark.parametrize("decode_type", ["SOFT16", "HARD8"])
def test_out_of_range_symbols_wrap(oracle, plan, decode_type):
    """symbols outside [low, high] make error_t arithmetic wrap (scalar.h:30-34 NOTE): still bit-exact."""
    code = COMMON_CODES[2]
    rng = np.random.default_rng(17)
    F, L = 6, 256
    S = L + code.K - 1
    if decode_type == "SOFT16":
        sym = rng.integers(-32768, 32768, size=(F, S, code.R)).astype(np.int16)
    else:
        sym = rng.integers(-128, 128, size=(F, S, code.R)).astype(np.int8)
    check_batch_against_oracle(oracle, code, decode_type, F, L, None, seed=0, plan=plan, sym=sym)


def test_lds2_block_boundaries_and_mid_block_renormalisation(oracle):
    """PLAN_LDS2 runs four trellis steps per barrier.  Cover every length of the last partial block (n_steps % 4, including
    frames shorter than one block) and renormalisations that fall after the 1st, 2nd, 3rd and 4th step of a block: with a
    threshold a little above the initial metrics state 0 reaches it every few steps, at drifting block positions."""
    import torch
    from oracle import pyoracle
    from viterbidecodercpp_amd import BatchDecoder, Code, ViterbiBranchTable, ViterbiDecoder_Config

    code = Code("K11R2", 11, 2, (0o3345, 0o3613))
    # (a) stock configuration, partial updates of 1..9 steps and whole frames with S % 4 = 0, 1, 2, 3
    for n in range(1, 10):
        check_batch_against_oracle(oracle, code, "SOFT16", 3, 64, 2.0, seed=n, plan=_lib.PLAN_LDS2, n_steps=n)
    for L in (8, 16, 24, 32, 40, 48):   # S = L + 10
        check_batch_against_oracle(oracle, code, "SOFT16", 5, L, 1.0, seed=L, plan=_lib.PLAN_LDS2)
    # (b) frequent renormalisation at every block position, both widths
    rng = np.random.default_rng(2024)
    for width, thr_list in ((2, (700, 1100, 2500, 6000)), (1, (20, 33, 47, 90))):
        sdt = np.int16 if width == 2 else np.int8
        high, low = (127, -127) if width == 2 else (3, -3)
        max_error = (high - low) * code.R
        for thr in thr_list:
            cfg = pyoracle.DecodeConfig(width, width, high, low, max_error, 0, max_error * (3 if width == 2 else 1), thr)
            table = ViterbiBranchTable(code.K, code.R, code.G, cfg.high, cfg.low, sdt)
            config = ViterbiDecoder_Config(cfg.max_error, cfg.initial_start_error, cfg.initial_non_start_error,
                                           cfg.renormalisation_threshold, np.uint16 if width == 2 else np.uint8)
            F, L = 3, 120 + int(rng.integers(0, 4))
            S = L + code.K - 1
            sym = rng.integers(low, high + 1, size=(F, S, code.R)).astype(sdt)
            want = [oracle.decode(code.K, code.R, code.G, cfg, sym[f], L) for f in range(F)]
            assert min(w["renorm_sum"] for w in want) > 0, "the case must renormalise"
            dec = BatchDecoder(table, config, plan=_lib.PLAN_LDS2)
            met, rs = dec.update(torch.from_numpy(sym).cuda(), L)
            got_dec = dec.export_decisions(F, L).cpu().numpy().view(np.uint64)
            out = dec.chainback(F, L).cpu().numpy()
            met = met.cpu().numpy()
            met = met.view(np.uint16) if width == 2 else met
            for f in range(F):
                assert np.array_equal(got_dec[f], want[f]["decisions"]), (width, thr, f)
                assert np.array_equal(met[f].astype(np.uint32), want[f]["metrics"]), (width, thr, f)
                assert int(rs[f].item()) == want[f]["renorm_sum"], (width, thr, f)
                assert np.array_equal(out[f], want[f]["bytes"]), (width, thr, f)


@pytest.mark.parametrize("code_id,F,L", [(2, 70, 4104), (2, 33, 2056), (5, 35, 2600), (5, 5, 1032)])
def test_long_frames_chainback_flushes(oracle, code_id, F, L):
    """The K = 7 / K = 9 chainback kernels park output bytes in LDS and flush them every 1024 steps: frames long enough for
    several flushes, lengths that leave a partial last flush and byte offsets that are not dword aligned, non-zero end
    states, frame counts that leave partial tiles."""
    code = COMMON_CODES[code_id]
    rng = np.random.default_rng(L)
    es = rng.integers(0, code.num_states, F).astype(np.int32)
    check_batch_against_oracle(oracle, code, "SOFT16", F, L, 2.0, seed=L + 1, end_state=es)
    check_batch_against_oracle(oracle, code, "HARD8", F, L + 8, 5.0, seed=L + 2)
