"""Batched streaming: a batch of decoders fed in chunks through vit_hip_update_batch_resume (the reference's update() is
incrementally callable with a cursor: viterbi_decoder_scalar.h:29-55, puncture_code_helpers.h:51).  Any split of the steps
must give the same decision rows, metrics, renormalisation sums and bytes as one call -- and as the oracle."""
import numpy as np
import pytest

from viterbidecodercpp_amd import COMMON_CODES, BatchDecoder, Code, ViterbiBranchTable, ViterbiDecoder_Config, _lib, synth
from tests.helpers import gpu_metrics_to_u32, make_table_config, oracle_frames

pytestmark = pytest.mark.gpu

K11 = Code("K11 R=1/2", 11, 2, (0o3345, 0o3613))
K6 = Code("K6 R=1/2", 6, 2, (0o65, 0o57))

CASES = [
    (COMMON_CODES[2], "SOFT16", _lib.PLAN_REG, 67, 336),      # K7: 24-step blocks, 4-step decision rows
    (COMMON_CODES[2], "HARD8", _lib.PLAN_REG, 33, 200),
    (COMMON_CODES[3], "SOFT8", _lib.PLAN_REG, 40, 152),       # K7 R3: odd symbol strides
    (COMMON_CODES[4], "SOFT16", _lib.PLAN_REG, 35, 144),      # K7 R4: 8-step LDS ring
    (COMMON_CODES[5], "SOFT16", _lib.PLAN_REG, 37, 120),      # K9
    (COMMON_CODES[6], "SOFT16", _lib.PLAN_REG, 20, 96),       # K9 R4
    (COMMON_CODES[0], "SOFT16", _lib.PLAN_REG, 130, 104),     # K3
    (COMMON_CODES[1], "SOFT8", _lib.PLAN_REG, 130, 104),      # K5
    (K11, "SOFT16", _lib.PLAN_LDS2, 5, 72),
    (Code("K10", 10, 2, (0o1167, 0o1545)), "SOFT8", _lib.PLAN_LDS2, 5, 96),    # K10: half-filled wavefront
    (COMMON_CODES[7], "SOFT16", _lib.PLAN_LDS2, 3, 40),       # K15
    (Code("K16 R=1/2", 16, 2, (46749, 58851)), "SOFT16", _lib.PLAN_LDS2, 3, 24),
    (K6, "SOFT16", _lib.PLAN_LDS, 4, 88),
    # run-time compiled register-plan instantiations carry their own resume kernels (same polynomials as tests/test_gpu_api.py:
    # the code objects come from the on-disk cache when that file ran first)
    (Code("custom K7", 7, 2, (0o171, 0o133)), "SOFT16", _lib.PLAN_REG, 40, 200),
    (K6, "SOFT16", _lib.PLAN_REG, 70, 104),
    (Code("custom K8", 8, 2, (0o371, 0o247)), "SOFT8", _lib.PLAN_REG, 37, 120),
    (COMMON_CODES[2], "SOFT16", _lib.PLAN_LDS, 3, 88),
]


def _splits(S, rng, style):
    if style == "halves":
        return [S // 2, S - S // 2]
    if style == "ones_then_rest":
        return [1, 1, 1, 2, 3, S - 8]
    cuts, left = [], S
    while left > 0:
        n = int(rng.integers(1, min(left, 61) + 1))
        cuts.append(n)
        left -= n
    return cuts


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("style", ["halves", "ones_then_rest", "random"])
def test_chunked_update_equals_one_call_and_the_oracle(oracle, case, style):
    import torch

    code, decode_type, plan, F, L = CASES[case]
    pc, table, config = make_table_config(code, decode_type)
    S = L + code.K - 1
    rng = np.random.default_rng(1000 * case + len(style))
    ebn0 = 1.0 if code.K < 15 else -4.0
    _, sym = synth.make_frames_numpy(code, pc, F, L, ebn0, seed=case)
    # stock configuration here; test_chunked_update_with_frequent_renormalisation below lowers the threshold
    start = rng.integers(0, code.num_states, F).astype(np.int32)
    end = rng.integers(0, code.num_states, F).astype(np.int32)
    d_sym = torch.from_numpy(sym).cuda()

    dec = BatchDecoder(table, config, plan=plan)
    met1, rs1 = dec.update(d_sym, L, start_state=start)
    dec1 = dec.export_decisions(F, L).clone()
    out1 = dec.chainback(F, L, end_state=end).clone()

    dec2 = BatchDecoder(table, config, plan=plan)
    ws = dec2.new_workspace(F, L)
    ws.fill_(0xA5)                                   # stale contents must not leak into the rows
    met2 = dec2.reset_batch(F, start_state=start)
    rs2 = torch.zeros(F, dtype=torch.int64, device="cuda")
    t = 0
    for i, n in enumerate(_splits(S, rng, style)):
        if i % 2 == 0:      # packed chunk
            chunk = d_sym[:, t:t + n].contiguous()
            rs2 += dec2.update_resume(chunk, L, t, met2, workspace=ws)
        else:               # a view into the whole batch: frame stride S*R
            view = d_sym.reshape(-1)[t * code.R:]
            rs2 += dec2.update_resume(view, L, t, met2, n_steps=n, symbol_frame_stride=S * code.R, workspace=ws)
        t += n
    assert t == S
    assert torch.equal(met1, met2), "metrics differ between one call and the chunked calls"
    assert torch.equal(rs1, rs2), "renormalisation sums differ"
    assert torch.equal(dec2.export_decisions(F, L, workspace=ws), dec1), "decision rows differ"
    assert torch.equal(dec2.chainback(F, L, end_state=end, workspace=ws), out1)
    # and both are the reference's answer
    want = oracle_frames(oracle, code, decode_type, sym, L, S, start, end)
    assert np.array_equal(dec1.cpu().numpy().view(np.uint64), want["decisions"])
    assert np.array_equal(gpu_metrics_to_u32(met2, pc.error_bytes), want["metrics"])
    assert np.array_equal(rs2.cpu().numpy().astype(np.uint64), want["renorm_sum"])
    assert np.array_equal(out1.cpu().numpy(), want["bytes"])


LOW_THR_CASES = [
    (COMMON_CODES[2], 2, _lib.PLAN_REG, 45, 150),      # K7 soft16
    (COMMON_CODES[2], 1, _lib.PLAN_REG, 33, 150),      # K7 u8
    (COMMON_CODES[5], 2, _lib.PLAN_REG, 34, 90),       # K9
    (COMMON_CODES[1], 2, _lib.PLAN_REG, 130, 90),      # K5: all states in one lane
    (K11, 2, _lib.PLAN_LDS2, 4, 60),
    (Code("K10", 10, 2, (0o1167, 0o1545)), 2, _lib.PLAN_LDS2, 4, 60),
    (COMMON_CODES[7], 2, _lib.PLAN_LDS2, 3, 33),       # K15: careful blocks predicted at every block position
    (K6, 2, _lib.PLAN_LDS, 4, 80),
]


@pytest.mark.parametrize("case", range(len(LOW_THR_CASES)))
def test_chunked_update_with_frequent_renormalisation(oracle, case):
    """A threshold a few steps' worth of error above the initial metrics: state 0 crosses it every two to six steps, so
    renormalisation lands on chunk boundaries, inside resumed entry blocks and (PLAN_LDS2) at every position of a radix-16
    block.  Chunks of every length 1..9 in turn; results must equal one call and the oracle."""
    import torch
    from oracle import pyoracle

    code, width, plan, F, L = LOW_THR_CASES[case]
    S = L + code.K - 1
    rng = np.random.default_rng(4200 + case)
    if width == 2:
        cfg = pyoracle.DecodeConfig(2, 2, 127, -127, 254 * code.R, 0, 300, 700)
    else:
        cfg = pyoracle.DecodeConfig(1, 1, 3, -3, 6 * code.R, 0, 10, 25)
    sdt = np.int16 if width == 2 else np.int8
    table = ViterbiBranchTable(code.K, code.R, code.G, cfg.high, cfg.low, sdt)
    config = ViterbiDecoder_Config(cfg.max_error, cfg.initial_start_error, cfg.initial_non_start_error,
                                   cfg.renormalisation_threshold, np.uint16 if width == 2 else np.uint8)
    sym = rng.integers(cfg.low, cfg.high + 1, size=(F, S, code.R)).astype(sdt)
    start = rng.integers(0, code.num_states, F).astype(np.int32)
    end = rng.integers(0, code.num_states, F).astype(np.int32)
    want = [oracle.decode(code.K, code.R, code.G, cfg, sym[f], L, start_state=int(start[f]), end_state=int(end[f]))
            for f in range(F)]
    assert min(w["renorm_sum"] for w in want) > 0, "the configuration must renormalise in every frame"
    d_sym = torch.from_numpy(sym).cuda()
    dec = BatchDecoder(table, config, plan=plan)
    ws = dec.new_workspace(F, L)
    ws.fill_(0x5A)
    met = dec.reset_batch(F, start_state=start)
    rs = torch.zeros(F, dtype=torch.int64, device="cuda")
    t, k = 0, 0
    while t < S:
        n = min(1 + k % 9, S - t)
        rs += dec.update_resume(d_sym.reshape(-1)[t * code.R:], L, t, met, n_steps=n, symbol_frame_stride=S * code.R, workspace=ws)
        t += n
        k += 1
    got_dec = dec.export_decisions(F, L, workspace=ws).cpu().numpy().view(np.uint64)
    out = dec.chainback(F, L, end_state=end, workspace=ws).cpu().numpy()
    m = gpu_metrics_to_u32(met, width)
    for f in range(F):
        assert np.array_equal(got_dec[f], want[f]["decisions"]), f
        assert np.array_equal(m[f], want[f]["metrics"]), f
        assert int(rs[f].item()) == want[f]["renorm_sum"], f
        assert np.array_equal(out[f], want[f]["bytes"]), f


def test_resume_argument_checks():
    import ctypes as C
    import torch

    code = COMMON_CODES[2]
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    F, L = 4, 64
    S = L + code.K - 1
    sym = torch.zeros((F, 10, code.R), dtype=torch.int16, device="cuda")
    met = dec.reset_batch(F)
    with pytest.raises(_lib.VitHipError):           # runs past the traceback length
        dec.update_resume(sym, L, S - 5, met)
    with pytest.raises(_lib.VitHipError):           # stride shorter than a chunk
        dec.update_resume(sym, L, 0, met, symbol_frame_stride=3)
    with pytest.raises(ValueError):                 # a view too short for the last frame's chunk
        dec.update_resume(sym.reshape(-1)[:F * 10 * code.R - 8], L, 0, met, n_steps=10, symbol_frame_stride=10 * code.R)
    with pytest.raises(ValueError):                 # packed chunks of the wrong rate
        dec.update_resume(torch.zeros((F, 10, code.R + 1), dtype=torch.int16, device="cuda"), L, 0, met)
    lib = _lib.load()
    ws = dec.new_workspace(F, L)
    rc = lib.vit_hip_update_batch_resume(dec._handle._h, C.c_void_p(sym.data_ptr()), 0, F, 0, 10, L, C.c_void_p(ws.data_ptr()),
                                         ws.numel(), None, None, None)
    assert rc == _lib.ERR_INVALID_ARG
