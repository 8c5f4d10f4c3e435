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
@pytest.mark.parametrize("K,R,G", [(11, 2, (0b10011011001, 0b11110110101)), (13, 3, (0o10533, 0o10675, 0o17661)),
                                   (12, 2, (0o4335, 0o5723)), (14, 2, (0o21645, 0o35661)), (15, 6, COMMON_CODES[7].G)])
def test_lds2_plan_large_k(oracle, K, R, G, decode_type):
    """PLAN_LDS2 (packed frame pair in LDS) on K = 11..15, including non-stock polynomials; odd frame count."""
    from viterbidecodercpp_amd import Code
    code = Code(f"K{K}R{R}", K, R, tuple(G))
    dec = check_batch_against_oracle(oracle, code, decode_type, 3, 48 if K >= 14 else 128, -2.0 if K >= 14 else 2.0, seed=K,
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
