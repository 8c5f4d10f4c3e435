"""CPU tests of the checker itself: the C restatement (oracle/viterbi_oracle.c) against the golden fixtures generated from
the real reference (tests/golden/make_golden.py), and -- where oracle/_ref/libvitref.so exists -- against the reference
live on fresh random inputs.  An oracle that fails here pins nothing."""
import json
import os

import numpy as np
import pytest

from oracle import pyoracle
from viterbidecodercpp_amd import COMMON_CODES, get_decoding_config, synth
from tests.helpers import decisions_match_fixture

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLD, "MANIFEST.json")))
DT = {"SOFT16": pyoracle.SOFT16, "SOFT8": pyoracle.SOFT8, "HARD8": pyoracle.HARD8}


def load_case(name):
    meta = MANIFEST[name]
    data = np.load(os.path.join(GOLD, name + ".npz"))
    return meta, data


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_oracle_matches_golden(oracle, name):
    meta, g = load_case(name)
    cfg = pyoracle.stock_config(DT[meta["decode_type"]], meta["R"])
    assert list(g["config"]) == cfg.cfg4
    table = oracle.branch_table(meta["K"], meta["R"], meta["G"], cfg.high, cfg.low)
    assert np.array_equal(table, g["table"])
    got = oracle.decode(meta["K"], meta["R"], meta["G"], cfg, g["symbols"], meta["L"], start_state=meta["start_state"],
                        end_state=meta["end_state"])
    assert decisions_match_fixture(got["decisions"], g)
    assert np.array_equal(got["metrics"], g["metrics"])
    assert got["renorm_sum"] == int(g["renorm_sum"])
    assert got["error"] == int(g["error"])
    assert np.array_equal(got["bytes"], g["bytes"])


@pytest.mark.parametrize("name", ["k7r2_soft16_2db", "k7r2_hard8_4db", "k9r2_soft16_2db", "k7r2_soft16_states"])
@pytest.mark.parametrize("chunk", [1, 7])
def test_oracle_incremental_update_equals_one_shot(oracle, name, chunk):
    """streaming: update() called with few symbols at a time (puncture_code_helpers.h:51) gives the same state."""
    meta, g = load_case(name)
    cfg = pyoracle.stock_config(DT[meta["decode_type"]], meta["R"])
    got = oracle.decode(meta["K"], meta["R"], meta["G"], cfg, g["symbols"], meta["L"], start_state=meta["start_state"],
                        end_state=meta["end_state"], chunk_steps=chunk)
    assert np.array_equal(got["decisions"], g["decisions"])
    assert np.array_equal(got["metrics"], g["metrics"])
    assert got["renorm_sum"] == int(g["renorm_sum"])
    assert np.array_equal(got["bytes"], g["bytes"])


def test_avx_tie_rule_is_not_the_parity_target():
    """the reference's SIMD strategies break ties the other way: they must NOT match the scalar decisions (negative control)."""
    differing = [m["avx_decision_words_differing"] for m in MANIFEST.values() if m["avx_decision_words_differing"] is not None]
    assert differing and all(d > 0 for d in differing)


def test_stock_codes_and_configs_match_reference(reflib):
    assert reflib.num_codes() == len(COMMON_CODES) == len(pyoracle.STOCK_CODES)
    for cid, code in enumerate(COMMON_CODES):
        name, K, R, G = reflib.stock_code(cid)
        assert (name, K, R, tuple(G)) == (code.name, code.K, code.R, tuple(code.G))
        assert (name, K, R, G) == tuple(pyoracle.STOCK_CODES[cid])
        for dt_name, dt in DT.items():
            want = reflib.stock_config(dt, R)
            assert pyoracle.stock_config(dt, R) == want
            pc = get_decoding_config(dt_name, R)
            assert pc.cfg4() == tuple(want.cfg4)
            assert (pc.soft_decision_high, pc.soft_decision_low, pc.soft_bytes) == (want.high, want.low, want.soft_bytes)


@pytest.mark.parametrize("cid", range(8))
def test_oracle_matches_reference_live(oracle, reflib, cid):
    """fresh noisy input, all three decode types, full-range garbage symbols included (wrapping arithmetic)."""
    code = COMMON_CODES[cid]
    rng = np.random.default_rng(100 + cid)
    L = 64 if code.K == 15 else 512
    S = L + code.K - 1
    for dt_name, dt in DT.items():
        cfg = pyoracle.stock_config(dt, code.R)
        pc = get_decoding_config(dt_name, code.R)
        _, sym = synth.make_frames_numpy(code, pc, 1, L, 2.0 - (6 if code.K == 15 else 0), seed=cid)
        lim = 32768 if cfg.soft_bytes == 2 else 128
        garbage = rng.integers(-lim, lim, size=(S, code.R)).astype(cfg.soft_dtype)
        for s in (sym[0], garbage):
            ss, es = int(rng.integers(0, code.num_states)), int(rng.integers(0, code.num_states))
            a = reflib.run(cid, cfg, s, L, start_state=ss, end_state=es)
            b = oracle.decode(code.K, code.R, code.G, cfg, s, L, start_state=ss, end_state=es)
            for k in ("decisions", "metrics", "bytes"):
                assert np.array_equal(a[k], b[k]), (code.name, dt_name, k)
            assert a["renorm_sum"] == b["renorm_sum"] and a["error"] == b["error"]


@pytest.mark.parametrize("cid", range(8))
def test_encoders_agree(oracle, reflib, cid):
    code = COMMON_CODES[cid]
    data = np.random.default_rng(cid).integers(0, 256, 48, dtype=np.uint8)
    want = reflib.encode(cid, data, which=0)
    assert np.array_equal(want, reflib.encode(cid, data, which=1))          # shift register == lookup (reference)
    assert np.array_equal(want, oracle.encode(code.K, code.R, code.G, data))  # oracle restatement
    assert np.array_equal(want, synth.encode_bits_numpy(code.K, code.R, code.G, data).reshape(-1))  # synthetic-data path


def test_oracle_frames_driver_threads(oracle):
    code = COMMON_CODES[2]
    pc = get_decoding_config("SOFT16", code.R)
    cfg = pyoracle.stock_config(pyoracle.SOFT16, code.R)
    _, sym = synth.make_frames_numpy(code, pc, 9, 256, 2.0, seed=5)
    a, ma, ra = oracle.decode_frames(code.K, code.R, code.G, cfg, sym, 256, threads=1, want_metrics=True)
    b, mb, rb = oracle.decode_frames(code.K, code.R, code.G, cfg, sym, 256, threads=4, want_metrics=True)
    assert np.array_equal(a, b) and np.array_equal(ma, mb) and np.array_equal(ra, rb)
    *_, ha = oracle.decode_frames(code.K, code.R, code.G, cfg, sym, 256, threads=3, want_hash=True)
    for f in range(9):
        one = oracle.decode(code.K, code.R, code.G, cfg, sym[f], 256)
        assert np.array_equal(one["bytes"], a[f]) and one["renorm_sum"] == int(ra[f])
        # the per-frame digest of the decision words (vo_decode_frames_hashed) restated in numpy
        w = one["decisions"].reshape(-1)
        with np.errstate(over="ignore"):
            h = (w * ((2 * np.arange(w.size, dtype=np.uint64) + 1) * np.uint64(oracle.HASH_MUL))).sum(dtype=np.uint64)
        assert int(ha[f]) == int(h) == oracle.hash_decisions(one["decisions"])
        w2 = w.copy()
        w2[17] ^= np.uint64(1) << np.uint64(40)     # any single flipped bit changes it (odd multipliers)
        assert oracle.hash_decisions(w2) != int(h)


def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """GPU sanitizers are not available on the pool; the checker itself at least runs clean under ASan + UBSan on the CPU:
    204 random decoders (K = 2 ... 11, both widths, thresholds 0 / type-max, ragged L, chunked updates, the threaded driver)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "oracle_sanitize")
    subprocess.run([gcc, "-std=c11", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-I" + os.path.join(root, "oracle"), "-o", exe, os.path.join(root, "tests", "cpp", "oracle_sanitize.c"),
                    os.path.join(root, "oracle", "viterbi_oracle.c"), "-lpthread"], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout and "runtime error" not in r.stderr, r.stdout + r.stderr
