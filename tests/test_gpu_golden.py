"""GPU vs the committed golden fixtures (outputs of the REAL reference): batched route with both kernel plans, and the
host-pointer route that mirrors the reference's decoder interface call for call (reset -> update -> get_error -> chainback,
examples/run_simple.cpp:76-80), including streaming update() calls of R symbols (puncture_code_helpers.h:51)."""
import json
import os

import numpy as np
import pytest

from viterbidecodercpp_amd import (BatchDecoder, ViterbiBranchTable, ViterbiDecoder_Config, ViterbiDecoder_Core,
                                   ViterbiDecoder_HIP, _lib, get_decoding_config)
from tests.helpers import decisions_match_fixture

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "MANIFEST.json")))


def _setup(name):
    meta = MANIFEST[name]
    g = np.load(os.path.join(GOLD, name + ".npz"))
    pc = get_decoding_config(meta["decode_type"], meta["R"])
    table = ViterbiBranchTable(meta["K"], meta["R"], meta["G"], pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    config = ViterbiDecoder_Config.from_decoder_config(pc)
    return meta, g, pc, table, config


@pytest.mark.parametrize("plan", [_lib.PLAN_LDS, _lib.PLAN_AUTO])
@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_batched_route_matches_golden(name, plan):
    import torch

    meta, g, pc, table, config = _setup(name)
    L, S = meta["L"], meta["steps"]
    dec = BatchDecoder(table, config, plan=plan)
    # replicate the fixture frame so that lanes / tiles beyond the first are exercised too
    F = 35
    sym = np.ascontiguousarray(np.broadcast_to(g["symbols"], (F,) + g["symbols"].shape))
    d_sym = torch.from_numpy(sym).cuda()
    ss = np.full(F, meta["start_state"], dtype=np.int32)
    es = np.full(F, meta["end_state"], dtype=np.int32)
    met, rs = dec.update(d_sym, L, start_state=ss)
    got_dec = dec.export_decisions(F, L).cpu().numpy().view(np.uint64)
    out = dec.chainback(F, L, end_state=es).cpu().numpy()
    met = met.cpu().numpy()
    met = met.view(np.uint16) if pc.error_bytes == 2 else met
    for f in (0, 15, 16, 31, 32, 34):
        assert decisions_match_fixture(got_dec[f], g), (name, f)
        assert np.array_equal(met[f].astype(np.uint32), g["metrics"]), (name, f)
        assert int(rs[f].item()) == int(g["renorm_sum"])
        assert np.array_equal(out[f], g["bytes"]), (name, f)
    if "avx_decisions" in g.files:      # negative control: we implement the scalar tie rule, not the SIMD one
        assert not np.array_equal(got_dec[0], g["avx_decisions"])


@pytest.mark.parametrize("chunk", [0, 1, 5])
@pytest.mark.parametrize("name", ["k7r2_soft16_2db", "k7r2_hard8_4db", "k7r2_soft16_states", "k9r2_soft16_2db",
                                  "k5r2_soft16_3db", "k15r6_soft16_m4db", "k7r2_soft16_2db_l4096", "k15r6_soft16_0db_l256"])
def test_host_route_mirrors_reference_call_pattern(name, chunk):
    meta, g, pc, table, config = _setup(name)
    L, R = meta["L"], meta["R"]
    vitdec = ViterbiDecoder_Core(table, config)
    vitdec.set_traceback_length(L)
    assert vitdec.get_traceback_length() == L
    vitdec.reset(meta["start_state"])
    sym = g["symbols"].reshape(-1)
    acc = 0
    if chunk == 0:
        acc += ViterbiDecoder_HIP.update(vitdec, sym)
    else:
        for t in range(0, meta["steps"], chunk):
            acc += ViterbiDecoder_HIP.update(vitdec, sym[t * R:(t + chunk) * R])
    assert vitdec.m_current_decoded_bit == meta["steps"]
    assert acc == int(g["renorm_sum"])
    assert vitdec.get_error(meta["end_state"]) == int(g["error"])
    assert np.array_equal(vitdec.m_metrics.astype(np.uint32), g["metrics"])
    assert decisions_match_fixture(vitdec.m_decisions, g)
    assert np.array_equal(vitdec.chainback(L, meta["end_state"]), g["bytes"])


def test_erasure_symbols_and_reuse():
    """depunctured streams carry 0 ("erasure") symbols (run_punctured_decoder.cpp:248-286); a Core is reused across
    frames after reset() (run_tests.cpp:163-190)."""
    from oracle import pyoracle

    meta, g, pc, table, config = _setup("k7r2_soft16_2db")
    oracle = pyoracle.Oracle()
    ocfg = pyoracle.stock_config(pyoracle.SOFT16, meta["R"])
    vitdec = ViterbiDecoder_Core(table, config)
    vitdec.set_traceback_length(meta["L"])
    for rep in range(2):
        sym = g["symbols"].copy()
        sym[rep::3] = 0                                   # puncture every third step
        vitdec.reset()
        acc = ViterbiDecoder_HIP.update(vitdec, sym.reshape(-1))
        want = oracle.decode(meta["K"], meta["R"], meta["G"], ocfg, sym, meta["L"])
        assert acc == want["renorm_sum"]
        assert np.array_equal(vitdec.m_decisions, want["decisions"])
        assert np.array_equal(vitdec.chainback(meta["L"]), want["bytes"])
