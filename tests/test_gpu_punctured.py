"""Punctured (DAB fast-information-channel) decode through the streaming interface: the scenario of the reference's
examples/run_punctured_decoder.cpp.  K=7 R=1/4 mother code, puncturing vectors PI_16 x 21 blocks,
PI_15 x 3 blocks and the 24-bit tail vector PI_X (ETSI EN 300 401 clause 11.1.2 / 11.2; vectors as listed in
run_punctured_decoder.cpp:39-76), punctured positions re-inserted as the erasure value 0, update() called with R symbols at
a time exactly as examples/helpers/puncture_code_helpers.h:17-55 does.  Pass = 0 bit errors on a clean channel (the
reference's criterion) AND bit-exact agreement with the oracle, clean and noisy."""
import numpy as np
import pytest

from viterbidecodercpp_amd import (COMMON_CODES, BatchDecoder, ViterbiBranchTable, ViterbiDecoder_Config, ViterbiDecoder_Core,
                                   ViterbiDecoder_HIP, get_decoding_config, synth)
from tests.helpers import oracle_cfg

pytestmark = pytest.mark.gpu

PI_16 = [1, 1, 1, 0] * 8
PI_15 = [1, 1, 1, 0] * 7 + [1, 1, 0, 0]
PI_X = [1, 1, 0, 0] * 6
TOTAL_DATA_BITS = 768                       # one FIC block of three 256-bit FIBs


def puncture_mask(K=7, R=4):
    mask = (PI_16 * 4) * 21 + (PI_15 * 4) * 3 + PI_X        # 128-symbol blocks, then the tail
    assert len(mask) == (TOTAL_DATA_BITS + K - 1) * R
    return np.asarray(mask, dtype=bool)


@pytest.mark.parametrize("decode_type", ["SOFT16", "SOFT8", "HARD8"])
@pytest.mark.parametrize("ebn0", [None, 6.0])
def test_dab_fic_punctured_streaming(oracle, decode_type, ebn0):
    import torch

    code = COMMON_CODES[4]                   # DAB Radio K=7 R=1/4
    pc = get_decoding_config(decode_type, code.R)
    table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    config = ViterbiDecoder_Config.from_decoder_config(pc)
    tx, sym = synth.make_frames_numpy(code, pc, 1, TOTAL_DATA_BITS, ebn0, seed=42)
    mask = puncture_mask()
    transmitted = sym[0].reshape(-1)[mask]                  # what goes over the air: 2/3 of the mother-code symbols
    depunctured = np.zeros(mask.size, dtype=pc.soft_dtype)  # erasure value 0 (run_punctured_decoder.cpp:165)
    depunctured[mask] = transmitted
    assert transmitted.size == 21 * 4 * 24 + 3 * (4 * 21 + 8) + 12

    want = oracle.decode(code.K, code.R, code.G, oracle_cfg(decode_type, code.R), depunctured, TOTAL_DATA_BITS)

    vitdec = ViterbiDecoder_Core(table, config)
    vitdec.set_traceback_length(TOTAL_DATA_BITS)
    vitdec.reset()
    acc = 0
    for t in range(0, depunctured.size, code.R):            # R symbols per update() call
        acc += ViterbiDecoder_HIP.update(vitdec, depunctured[t:t + code.R])
    rx = vitdec.chainback(TOTAL_DATA_BITS)
    assert acc == want["renorm_sum"] and vitdec.get_error() == want["error"]
    assert np.array_equal(vitdec.m_decisions, want["decisions"])
    assert np.array_equal(rx, want["bytes"])
    if ebn0 is None:
        assert np.array_equal(rx, tx[0])                    # 0 incorrect bits, the reference's pass criterion

    # the batched route on the same depunctured stream (many FIC blocks at once in a receiver)
    dec = BatchDecoder(table, config)
    batch = np.ascontiguousarray(np.broadcast_to(depunctured.reshape(1, -1, code.R), (40, depunctured.size // code.R, code.R)))
    out = dec.decode(torch.from_numpy(batch).cuda(), TOTAL_DATA_BITS).cpu().numpy()
    assert all(np.array_equal(out[f], want["bytes"]) for f in range(40))


@pytest.mark.parametrize("decode_type", ["SOFT16", "SOFT8", "HARD8"])
def test_dab_fic_punctured_batch_front_end(oracle, decode_type):
    """The same scenario at receiver scale: many noisy FIC blocks, depunctured ON THE DEVICE (vit_hip_depuncture_batch) and
    decoded by the batch route; every frame bit-exact against the oracle fed with the host-depunctured stream."""
    import torch

    code = COMMON_CODES[4]
    pc = get_decoding_config(decode_type, code.R)
    table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    config = ViterbiDecoder_Config.from_decoder_config(pc)
    F = 37
    tx, sym = synth.make_frames_numpy(code, pc, F, TOTAL_DATA_BITS, 5.0, seed=7)
    mask = puncture_mask()
    flat = sym.reshape(F, -1)
    transmitted = np.ascontiguousarray(flat[:, mask])
    depunctured = np.zeros_like(flat)
    depunctured[:, mask] = transmitted

    dec = BatchDecoder(table, config)
    d_sym = dec.depuncture(torch.from_numpy(transmitted).cuda(), mask)
    assert d_sym.shape == (F, TOTAL_DATA_BITS + code.K - 1, code.R)
    assert np.array_equal(d_sym.cpu().numpy().reshape(F, -1), depunctured)
    out = dec.decode(d_sym, TOTAL_DATA_BITS).cpu().numpy()
    for f in range(F):
        want = oracle.decode(code.K, code.R, code.G, oracle_cfg(decode_type, code.R), depunctured[f], TOTAL_DATA_BITS)
        assert np.array_equal(out[f], want["bytes"]), f
    # ragged tail: a symbol count that is not a multiple of the 8-symbol store width, odd frame offsets for 8-bit symbols
    short = np.ones(5 * code.R, dtype=bool)
    short[[1, 6, 7, 13]] = False
    src = np.ascontiguousarray(flat[:, : int(short.sum())])
    got = dec.depuncture(torch.from_numpy(src).cuda(), short).cpu().numpy().reshape(F, -1)
    ref = np.zeros((F, short.size), dtype=flat.dtype)
    ref[:, short] = src
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("decode_type", ["SOFT16", "HARD8"])
def test_streaming_exact_return_mode_and_short_frames(oracle, decode_type):
    """(1) set_exact_update_return(True): every N = R call returns what the reference's update() returns for THAT call
    (examples/helpers/puncture_code_helpers.h:51 accumulates per call) -- checked call by call against the oracle streamed the
    same way.  (2) default (deferred) mode with a frame SHORTER than the traceback buffer: the flush comes from get_error(),
    its renormalisation sum is owed to THAT frame -- collected by take_unreported_renormalisation(); reset() drops it, so the next
    frame's total starts from zero (round-4 advisor finding: a carried debt inflated the next frame's error metric)."""
    code = COMMON_CODES[4]
    pc = get_decoding_config(decode_type, code.R)
    table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    config = ViterbiDecoder_Config.from_decoder_config(pc)
    L = 2048 if decode_type == "SOFT16" else 256           # long enough to renormalise several times
    tx, sym = synth.make_frames_numpy(code, pc, 1, L, 2.0, seed=11)
    flat = sym[0].reshape(-1)
    S = L + code.K - 1
    cfg = oracle_cfg(decode_type, code.R)
    whole = oracle.decode(code.K, code.R, code.G, cfg, flat, L)
    assert whole["renorm_sum"] > 0
    # the oracle's per-call values: its update() called with the R symbols of one step at a time
    otab = oracle.branch_table(code.K, code.R, code.G, cfg.high, cfg.low)
    om = oracle.reset(code.K, code.R, cfg, 0)
    per_call = [oracle.update(code.K, code.R, cfg, otab, om, np.ascontiguousarray(flat[t * code.R:(t + 1) * code.R]))[1] for t in range(S)]
    assert sum(per_call) == whole["renorm_sum"] and max(per_call) > 0

    exact = ViterbiDecoder_Core(table, config)
    exact.set_traceback_length(L)
    exact.set_exact_update_return(True)
    exact.reset()
    got = []
    for t in range(S):
        got.append(ViterbiDecoder_HIP.update(exact, flat[t * code.R:(t + 1) * code.R]))
        assert exact._pending_steps == 0
    assert got == per_call                                 # call by call, not only in total
    assert exact.get_error() == whole["error"]
    assert np.array_equal(exact.chainback(L), whole["bytes"])

    short = ViterbiDecoder_Core(table, config)
    short.set_traceback_length(L + 333)
    for collect in (True, False):
        short.reset()
        acc = 0
        for t in range(S):
            acc += ViterbiDecoder_HIP.update(short, flat[t * code.R:(t + 1) * code.R])
        err = short.get_error()                           # runs what was still queued
        assert err == whole["error"]
        if collect:
            acc += short.take_unreported_renormalisation()
            assert acc == whole["renorm_sum"]
        else:
            short.reset()                                   # the tail's sum is dropped, not carried into the next frame ...
            # ... but not silently: what reset() dropped stays readable, and with it the frame's total is the reference's
            assert acc + short.last_dropped_renormalisation() == whole["renorm_sum"]
            first = ViterbiDecoder_HIP.update(short, flat[:code.R])
            short.get_error()
            assert first + short.take_unreported_renormalisation() == per_call[0]
            assert acc <= whole["renorm_sum"]
