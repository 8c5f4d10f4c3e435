"""C-ABI behaviour on the GPU box: error codes (no exceptions, no crashes), plan selection, workspace checks, empty and
degenerate batches, handle reuse across batches of different sizes."""
import ctypes as C
import os

import numpy as np
import pytest

from viterbidecodercpp_amd import COMMON_CODES, BatchDecoder, Code, _lib, synth
from tests.helpers import check_batch_against_oracle, make_table_config, oracle_cfg

pytestmark = pytest.mark.gpu


def test_plan_selection_and_errors():
    lib = _lib.load()
    expect = {0: _lib.PLAN_REG, 1: _lib.PLAN_REG, 2: _lib.PLAN_REG, 3: _lib.PLAN_REG, 4: _lib.PLAN_REG, 5: _lib.PLAN_REG,
              6: _lib.PLAN_REG, 7: _lib.PLAN_LDS2}
    for cid, plan in expect.items():
        pc, table, config = make_table_config(COMMON_CODES[cid], "SOFT16")
        dec = BatchDecoder(table, config)
        assert dec.plan == plan, COMMON_CODES[cid].name
        info = dec._handle.info
        assert list(info.polynomials[:COMMON_CODES[cid].R]) == list(COMMON_CODES[cid].G) and info.table_is_linear == 1
        assert (info.soft_decision_high, info.soft_decision_low) == (127, -127)
    # K=10: PLAN_LDS2 since round 5 (no register plan)
    code = Code("custom", 10, 2, (0o1755, 0o1363))
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    assert dec.plan == _lib.PLAN_LDS2
    assert lib.vit_hip_set_plan(dec._handle._h, _lib.PLAN_REG) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_set_plan(dec._handle._h, _lib.PLAN_LDS) == _lib.OK
    # K=6, R=7 has neither (no fast plan reads more than six symbols per step): served by the LDS plan, the others refused
    code = Code("custom", 6, 7, (0o65, 0o57, 0o75, 0o53, 0o71, 0o47, 0o77))
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    assert dec.plan == _lib.PLAN_LDS
    assert lib.vit_hip_set_plan(dec._handle._h, _lib.PLAN_REG) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_set_plan(dec._handle._h, _lib.PLAN_LDS2) == _lib.ERR_UNSUPPORTED
    assert lib.vit_hip_set_plan(dec._handle._h, 99) == _lib.ERR_INVALID_ARG
    assert lib.vit_hip_set_plan(dec._handle._h, _lib.PLAN_AUTO) == _lib.OK


@pytest.mark.parametrize("K,R,G,decode_type", [
    (7, 6, (0o171, 0o133, 0o165, 0o117, 0o135, 0o157), "SOFT16"),
    (7, 5, (0o171, 0o133, 0o165, 0o117, 0o135), "SOFT8"),
    (9, 6, (0o557, 0o663, 0o711, 0o561, 0o753, 0o715), "SOFT16"),
    (9, 5, (0o557, 0o663, 0o711, 0o561, 0o753), "HARD8"),
    (8, 6, (0o371, 0o247, 0o367, 0o331, 0o225, 0o313), "SOFT16"),     # K = 8: split tables + the sub-chunk fetch (253 registers, no scratch)
])
def test_rate_5_and_6_codes_on_the_register_plan(oracle, K, R, G, decode_type):
    """R = 5, 6 at K = 7, 9: run-time compiled register-plan instantiations with SPLIT pattern tables (low / high part, two packed
    instructions per butterfly to join them) -- not the compatibility plan.  Noisy frames long enough to renormalise, start / end
    states, ragged lengths; PLAN_AUTO without VIT_HIP_JIT lands on PLAN_LDS and says what to do about it."""
    code = Code(f"K{K}R{R}", K, R, tuple(G))
    pc, table, config = make_table_config(code, decode_type)
    auto = BatchDecoder(table, config)
    assert auto.plan == _lib.PLAN_LDS and "vit_hip_set_plan(h, VIT_HIP_PLAN_REG)" in auto.plan_note and "COMPATIBILITY" in auto.plan_note
    rng = np.random.default_rng(K * R)
    for F, L in ((70, 1400), (33, 104), (3, 8)):
        ss = rng.integers(0, code.num_states, F).astype(np.int32)
        es = rng.integers(0, code.num_states, F).astype(np.int32)
        dec = check_batch_against_oracle(oracle, code, decode_type, F, L, 2.0, seed=F + L, plan=_lib.PLAN_REG, start_state=ss, end_state=es)
        assert dec.plan == _lib.PLAN_REG and dec.plan_note.startswith(f"K={K} R={R}: PLAN_REG")
    # out-of-range symbols and a fuzzed configuration (the split sums wrap like the whole ones)
    from tests.test_gpu_fuzz import random_config
    from viterbidecodercpp_amd import ViterbiBranchTable, ViterbiDecoder_Config
    width = 2 if decode_type == "SOFT16" else 1
    for trial in range(3):
        cfg = random_config(rng, width, trial)
        sdt = np.int16 if width == 2 else np.int8
        tb = ViterbiBranchTable(K, R, code.G, cfg.high, cfg.low, sdt)
        cf = ViterbiDecoder_Config(cfg.max_error, cfg.initial_start_error, cfg.initial_non_start_error, cfg.renormalisation_threshold,
                                   np.uint16 if width == 2 else np.uint8)
        lim = 1 << (8 * width - 1)
        F, L = 37, 150
        sym = rng.integers(-lim, lim, size=(F, L + K - 1, R)).astype(sdt)
        dec = BatchDecoder(tb, cf, plan=_lib.PLAN_REG)
        import torch
        met, rs = dec.update(torch.from_numpy(sym).cuda(), L)
        got_dec = dec.export_decisions(F, L).cpu().numpy().view(np.uint64)
        out = dec.chainback(F, L).cpu().numpy()
        met = met.cpu().numpy()
        met = met.view(np.uint16) if width == 2 else met
        for f in range(F):
            want = oracle.decode(K, R, code.G, cfg, sym[f], L)
            assert np.array_equal(got_dec[f], want["decisions"]) and np.array_equal(out[f], want["bytes"]), (trial, f)
            assert np.array_equal(met[f].astype(np.uint32), want["metrics"]) and int(rs[f].item()) == want["renorm_sum"], (trial, f)


def test_custom_polynomials_decode(oracle):
    code = Code("NASA K=7 (171,133)", 7, 2, (0o171, 0o133))
    check_batch_against_oracle(oracle, code, "SOFT16", 5, 512, 2.0, seed=21)
    code = Code("K=9 R=1/3 custom", 9, 3, (0o557, 0o663, 0o711))
    check_batch_against_oracle(oracle, code, "HARD8", 3, 256, 4.0, seed=22)
    code = Code("K=4 R=1/2", 4, 2, (0o15, 0o17))
    check_batch_against_oracle(oracle, code, "SOFT8", 4, 256, 3.0, seed=23)
    code = Code("K=2 R=1/2", 2, 2, (0b11, 0b11))
    check_batch_against_oracle(oracle, code, "SOFT16", 2, 64, 3.0, seed=24)


def test_workspace_and_argument_checks():
    import torch

    lib = _lib.load()
    code = COMMON_CODES[2]
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    F, L = 4, 64
    _, sym = synth.make_frames_numpy(code, pc, F, L, 3.0, seed=1)
    d_sym = torch.from_numpy(sym).cuda()
    need = dec.workspace_bytes(F, L)
    ws = torch.empty(need + 512, dtype=torch.uint8, device="cuda")
    out = torch.empty((F, L // 8), dtype=torch.uint8, device="cuda")
    h = dec._handle._h
    p = lambda t, off=0: C.c_void_p(t.data_ptr() + off)  # noqa: E731
    S = L + code.K - 1
    assert lib.vit_hip_update_batch(h, p(d_sym), F, S, L, p(ws), need - 1, None, None, None, None) == _lib.ERR_WORKSPACE
    assert lib.vit_hip_update_batch(h, p(d_sym), F, S, L, p(ws, 16), need, None, None, None, None) == _lib.ERR_WORKSPACE
    assert lib.vit_hip_update_batch(h, p(d_sym), F, S + 1, L, p(ws), need, None, None, None, None) == _lib.ERR_INVALID_ARG
    assert lib.vit_hip_update_batch(h, None, F, S, L, p(ws), need, None, None, None, None) == _lib.ERR_INVALID_ARG
    assert b"NULL" in lib.vit_hip_last_error()
    assert lib.vit_hip_update_batch(h, p(d_sym), 0, S, L, p(ws), need, None, None, None, None) == _lib.OK   # empty batch
    assert lib.vit_hip_chainback_batch(h, p(ws), 0, L, p(out), None, None) == _lib.OK
    assert lib.vit_hip_decode_batch(h, p(d_sym), F, L, p(ws), need, p(out), None, None, None, None) == _lib.OK
    torch.cuda.synchronize()
    assert lib.vit_hip_destroy(None) == _lib.OK


def test_handle_reuse_across_batch_shapes(oracle):
    """one decoder, many batches: sizes change, results stay exact (workspace re-used and re-grown)."""
    import torch
    from oracle import pyoracle

    code = COMMON_CODES[2]
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    ocfg = pyoracle.stock_config(pyoracle.SOFT16, code.R)
    for F, L in [(33, 256), (1, 8), (70, 1024), (2, 64), (64, 512)]:
        tx, sym = synth.make_frames_numpy(code, pc, F, L, 2.0, seed=F + L)
        out = dec.decode(torch.from_numpy(sym).cuda(), L).cpu().numpy()
        want, _, _ = oracle.decode_frames(code.K, code.R, code.G, ocfg, sym, L, threads=4)
        assert np.array_equal(out, want), (F, L)


def test_blob_round_trip_builds_identical_decoder(oracle):
    import torch
    from viterbidecodercpp_amd import pack_blob

    code = COMMON_CODES[5]
    pc, table, config = make_table_config(code, "HARD8")
    a = BatchDecoder(table, config)
    b = BatchDecoder(blob=pack_blob(table, config))
    assert (a.K, a.R, a.plan, a.soft_bytes) == (b.K, b.R, b.plan, b.soft_bytes)
    _, sym = synth.make_frames_numpy(code, pc, 6, 256, 4.0, seed=4)
    d = torch.from_numpy(sym).cuda()
    assert torch.equal(a.decode(d, 256), b.decode(d, 256))


def test_batch_calls_capture_into_a_hip_graph(oracle):
    """the batch entry points only enqueue work (no allocation, no synchronisation): a decode captured once into a
    hipGraph (torch.cuda.CUDAGraph) replays on new symbols and stays bit-exact."""
    import torch
    from oracle import pyoracle

    code = COMMON_CODES[2]
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    F, L = 96, 512
    ocfg = pyoracle.stock_config(pyoracle.SOFT16, code.R)
    _, sym0 = synth.make_frames_numpy(code, pc, F, L, 2.0, seed=31)
    d_sym = torch.from_numpy(sym0).cuda()
    out = torch.empty((F, L // 8), dtype=torch.uint8, device="cuda")
    ws = dec.new_workspace(F, L)
    dec.update(d_sym, L, want_metrics=False, workspace=ws)       # warm-up outside capture
    dec.chainback(F, L, out=out, workspace=ws)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        dec.update(d_sym, L, want_metrics=False, workspace=ws)
        dec.chainback(F, L, out=out, workspace=ws)
    for seed in (32, 33):
        _, sym = synth.make_frames_numpy(code, pc, F, L, 2.0, seed=seed)
        d_sym.copy_(torch.from_numpy(sym))
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        want, _, _ = oracle.decode_frames(code.K, code.R, code.G, ocfg, sym, L, threads=4)
        assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("K,R,G,decode_type", [
    (7, 2, (0o171, 0o133), "SOFT16"),        # the CCSDS/NASA pair written in the other polynomial order / bit order
    (7, 3, (0o133, 0o171, 0o165), "HARD8"),
    (9, 2, (0o557, 0o663), "SOFT16"),
    (9, 3, (0o557, 0o663, 0o711), "SOFT8"),
    (5, 3, (0o25, 0o33, 0o37), "SOFT16"),
    (4, 2, (0o15, 0o17), "SOFT16"),
    (3, 4, (0o5, 0o7, 0o7, 0o5), "HARD8"),
    (6, 2, (0o65, 0o57), "SOFT16"),          # K = 6: 32 states per lane, two decision dwords per step (round 2)
    (6, 4, (0o65, 0o57, 0o75, 0o53), "HARD8"),
    (8, 2, (0o371, 0o247), "SOFT16"),        # K = 8: 32 registers per lane over 4 lanes, a decision row holds two steps
    (8, 2, (0o371, 0o247), "SOFT8"),
    (8, 3, (0o367, 0o331, 0o225), "SOFT16"),   # K = 8 with 8 / 16 patterns (round 5: PLAN_LDS before): branch metrics per sub-chunk, as at K = 9
    (8, 4, (0o371, 0o247, 0o367, 0o331), "HARD8"),
    (6, 3, (0o65, 0o57, 0o75), "SOFT8"),       # K = 6 at an odd rate: a 240-step unrolled block
    (5, 5, (0o27, 0o31, 0o33, 0o37, 0o35), "SOFT16"),   # R = 5, 6 below K = 7: every sum formed in the lane, no table to split
    (2, 2, (0o3, 0o1), "SOFT16"),              # K = 2: two states, one butterfly
    (2, 3, (0o3, 0o2, 0o3), "HARD8"),
    (6, 6, (0o65, 0o57, 0o75, 0o53, 0o71, 0o47), "HARD8"),
])
def test_plan_reg_runtime_instantiation(oracle, K, R, G, decode_type):
    """polynomials outside the ahead-of-time table: PLAN_REG is compiled for them on first use (reg_jit.hpp) and must
    agree with the oracle exactly like the stock instantiations."""
    code = Code(f"custom K{K}R{R}", K, R, tuple(G))
    pc, table, config = make_table_config(code, decode_type)
    # AUTO does not compile anything behind the caller's back, and does not look into the user cache: PLAN_REG only for the sets
    # that were precompiled into the package at install time (tools/precompile.py: COMMON_SETS)
    from viterbidecodercpp_amd.tools.precompile import COMMON_SETS
    in_package = any((K, R, tuple(G)) == (k, r, g) for _, k, r, g in COMMON_SETS)
    generic = 3 <= K <= 9 and 1 <= R <= 4 and not (K == 6 and R % 2)       # ... or, for these (K, R), the package's GENERIC kernels (tests/test_gpu_generic.py)
    auto = BatchDecoder(table, config)
    assert auto.plan == (_lib.PLAN_REG if in_package or generic else _lib.PLAN_LDS)
    assert ("GENERIC" in auto.plan_note) == (generic and not in_package), auto.plan_note
    # asking for the register plan by name gets kernels specialised for the set (here: from the session's user cache)
    dec = check_batch_against_oracle(oracle, code, decode_type, 70, 384, 3.0, seed=K * R, plan=_lib.PLAN_REG)
    assert dec.plan == _lib.PLAN_REG and "GENERIC" not in dec.plan_note, dec.plan_note
    # the run-time compiled code object's kernel descriptors are readable too (the pipeline's residency rules need them): from
    # the .hsaco in the cache directory
    upd, cb = dec.kernel_resources(_lib.KERNEL_UPDATE), dec.kernel_resources(_lib.KERNEL_CHAINBACK)
    assert 8 <= upd["vgpr_alloc"] <= 512 and upd["vgpr_alloc"] % 8 == 0 and 8 <= cb["vgpr_alloc"] <= 512, (upd, cb)
    if K == 9:
        assert cb["vgpr_alloc"] <= 32 and cb["lds_static_bytes"] == 0 and cb["lds_dynamic_bytes"] == 8 * 4096 + 8192, cb
    rng = np.random.default_rng(K)
    ss = rng.integers(0, code.num_states, 33).astype(np.int32)
    es = rng.integers(0, code.num_states, 33).astype(np.int32)
    check_batch_against_oracle(oracle, code, decode_type, 33, 104, 2.0, seed=K + R, plan=_lib.PLAN_REG, start_state=ss, end_state=es)


@pytest.mark.parametrize("alt", [_lib.KERNEL_CHAINBACK, _lib.KERNEL_CHAINBACK_ALT])
@pytest.mark.parametrize("K,R,G,decode_type", [
    (9, 2, (0o753, 0o561), "SOFT16"),        # CDMA IS-95A, ahead-of-time instantiation
    (9, 4, (0o765, 0o671, 0o513, 0o473), "HARD8"),
    (9, 3, (0o557, 0o663, 0o711), "SOFT8"),  # run-time instantiation
    (7, 2, (0o155, 0o117), "SOFT16"),        # Voyager
    (7, 4, (0o155, 0o117, 0o123, 0o155), "SOFT16"),   # DAB Radio: its LDS-ring kernel has a two-slot ring (12 KiB)
    (7, 2, (0o171, 0o133), "HARD8"),         # run-time instantiation
])
def test_chainback_bodies(oracle, K, R, G, decode_type, alt):
    """K = 7 and K = 9 have two chainback kernels each -- rows streamed through an LDS ring (K = 9: the one in use; K = 7:
    experiments only) and a register-ring / cooperative one (K = 7: the one in use; K = 9: the alternative): both are run here,
    with trace lengths that leave ragged ends above and below the 32-step iterations, frame counts that are no multiple of a
    wave's 128, and per-frame end states."""
    code = Code(f"K{K}R{R}", K, R, tuple(G))
    rng = np.random.default_rng(R)
    for F, L in ((130, 1000), (70, 384), (33, 104), (257, 40), (1, 24), (3, 8)):
        ss = rng.integers(0, code.num_states, F).astype(np.int32)
        es = rng.integers(0, code.num_states, F).astype(np.int32)
        check_batch_against_oracle(oracle, code, decode_type, F, L, 2.0, seed=F + L, plan=_lib.PLAN_REG, start_state=ss, end_state=es,
                                   chainback_kernel=alt)
    check_batch_against_oracle(oracle, code, decode_type, 200, 2048, 1.0, seed=9, plan=_lib.PLAN_REG, chainback_kernel=alt)
    check_batch_against_oracle(oracle, code, decode_type, 40, 2048 + 8 * (K + R), 1.0, seed=10, plan=_lib.PLAN_REG, chainback_kernel=alt)


def test_plan_reg_runtime_instantiation_failure_is_an_error_code(monkeypatch, tmp_path):
    lib = _lib.load()
    monkeypatch.setenv("VIT_HIP_HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("VIT_HIP_CACHE_DIR", str(tmp_path))
    code = Code("custom", 6, 5, (0o73, 0o45, 0o51, 0o67, 0o75))      # (R = 5: no generic kernels to fall back on)
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    assert lib.vit_hip_set_plan(dec._handle._h, _lib.PLAN_REG) == _lib.ERR_UNSUPPORTED
    msg = lib.vit_hip_last_error()
    assert b"hipcc" in msg, msg
    dec._handle.refresh()
    assert dec.plan == _lib.PLAN_LDS                                  # still usable on the LDS plan
    # K = 3..9, R = 1..4: the handle already runs the package's GENERIC kernels; the failed attempt at specialised ones leaves them in place
    pc, table, config = make_table_config(Code("custom", 7, 2, (0o147, 0o135)), "SOFT16")
    dec = BatchDecoder(table, config)
    assert dec.plan == _lib.PLAN_REG and "GENERIC" in dec.plan_note
    assert lib.vit_hip_set_plan(dec._handle._h, _lib.PLAN_REG) == _lib.OK
    dec._handle.refresh()
    assert dec.plan == _lib.PLAN_REG and "GENERIC" in dec.plan_note


def test_precompiled_package_cache_serves_plan_reg_without_a_compiler(oracle, monkeypatch, tmp_path):
    """the reference takes the polynomials at run time (viterbi_branch_table.h:34-55).  The world's most deployed K = 7 pair in its
    standard orientation (0o171, 0o133) is not a stock code: with NO compiler on the host and an EMPTY user cache, vit_hip_create
    (PLAN_AUTO) still runs it on the register plan, from the code object build() precompiled into the package cache -- bit-exact,
    at the stock code's rate (65536 x 8192 through the pipeline: >= 140 Gbit/s)."""
    import time

    import torch
    from viterbidecodercpp_amd import DecodePipeline

    monkeypatch.setenv("VIT_HIP_HIPCC", "/nonexistent/hipcc")
    monkeypatch.setenv("VIT_HIP_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("VIT_HIP_JIT", raising=False)
    for K, R, G, decode_type in ((7, 2, (0o171, 0o133), "SOFT16"), (7, 2, (0o171, 0o133), "HARD8"), (5, 2, (0o23, 0o33), "SOFT8"),
                                 (9, 2, (0o561, 0o753), "SOFT16"), (7, 3, (0o171, 0o165, 0o133), "SOFT8")):
        code = Code(f"K{K}R{R}", K, R, G)
        dec = check_batch_against_oracle(oracle, code, decode_type, 70, 384, 2.0, seed=K + R)           # plan=None: PLAN_AUTO
        assert dec.plan == _lib.PLAN_REG, dec.plan_note
        assert "package cache" in dec.plan_note and "/precompiled/reg_K" in dec.plan_note, dec.plan_note
    assert not any(f.endswith(".hsaco") for f in os.listdir(tmp_path))          # nothing was compiled
    # a set that was NOT precompiled runs the package's GENERIC kernels where (K, R) has them (K = 3..9, R = 1..4: tests/test_gpu_generic.py)
    # and stays on the compatibility plan where not
    pc, table, config = make_table_config(Code("custom", 7, 2, (0o147, 0o135)), "SOFT16")
    d = BatchDecoder(table, config)
    assert d.plan == _lib.PLAN_REG and "GENERIC" in d.plan_note, d.plan_note
    pc, table, config = make_table_config(Code("custom", 6, 5, (0o73, 0o45, 0o51, 0o67, 0o75)), "SOFT16")
    assert BatchDecoder(table, config).plan == _lib.PLAN_LDS
    # rate: the headline batch size
    code = Code("802.11 K7", 7, 2, (0o171, 0o133))
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    F, L = 65536, 8192
    tx, sym = dec.synth(F, L, 3.0, seed=5)
    out = torch.empty((F, L // 8), dtype=torch.uint8, device="cuda")
    pipe = DecodePipeline(dec, F, L)
    for _ in range(4):
        pipe.submit(sym, out)
    pipe.sync()
    t0 = time.perf_counter()
    for _ in range(12):
        pipe.submit(sym, out)
    pipe.sync()
    gbit = 12 * F * L / (time.perf_counter() - t0) / 1e9
    ber = float((torch.bitwise_xor(out, tx).to(torch.int32).view(-1) != 0).float().mean())
    assert ber < 2e-2
    n = 64
    want, _, _ = oracle.decode_frames(7, 2, code.G, oracle_cfg("SOFT16", 2), sym[:n].cpu().numpy(), L, threads=8)
    assert np.array_equal(out[:n].cpu().numpy(), want)
    print(f"precompiled K7 0o171/0o133 SOFT16 65536 x 8192: {gbit:.1f} Gbit/s")
    assert gbit >= 140.0, gbit


def test_runtime_compile_on_the_box(oracle, monkeypatch, tmp_path):
    """the compile path itself (every other run-time instantiated set of this suite comes precompiled from tests/_jit_cache): an
    empty user cache, hipcc on the box, a set nobody precompiled -- vit_hip_set_plan(PLAN_REG) compiles it, the note names the user
    cache, and a second handle loads the object without compiling."""
    monkeypatch.setenv("VIT_HIP_CACHE_DIR", str(tmp_path))
    code = Code("custom K4", 4, 2, (0o13, 0o17))
    dec = check_batch_against_oracle(oracle, code, "SOFT16", 70, 384, 3.0, seed=4, plan=_lib.PLAN_REG)
    assert dec.plan == _lib.PLAN_REG and "user cache" in dec.plan_note and str(tmp_path) in dec.plan_note, dec.plan_note
    objs = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    assert len(objs) == 1 and objs[0].startswith("reg_K4R2_11_15_0_0_0_0_s16_gfx950_"), objs
    st = os.stat(os.path.join(tmp_path, objs[0]))
    assert (st.st_mode & 0o077) == 0                                            # private to this user
    monkeypatch.setenv("VIT_HIP_HIPCC", "/nonexistent/hipcc")                   # the cached object needs no compiler
    pc, table, config = make_table_config(code, "SOFT16")
    assert BatchDecoder(table, config, plan=_lib.PLAN_REG).plan == _lib.PLAN_REG


@pytest.mark.parametrize("code_id,decode_type,plans", [
    (2, "SOFT16", [_lib.PLAN_REG, _lib.PLAN_LDS]), (2, "HARD8", [_lib.PLAN_REG, _lib.PLAN_LDS]),
    (3, "SOFT8", [_lib.PLAN_REG, _lib.PLAN_LDS]), (5, "HARD8", [_lib.PLAN_REG]), (1, "SOFT8", [_lib.PLAN_REG]),
    (7, "HARD8", [_lib.PLAN_LDS2]),
])
def test_symbol_buffers_at_odd_offsets(oracle, code_id, decode_type, plans):
    """a caller may hand over a view into a larger buffer: the symbol pointer is then only element-aligned."""
    import torch
    from tests.helpers import oracle_cfg

    code = COMMON_CODES[code_id]
    pc, table, config = make_table_config(code, decode_type)
    F, L = (3, 32) if code.K == 15 else (37, 120)
    S = L + code.K - 1
    _, sym = synth.make_frames_numpy(code, pc, F, L, 3.0 if code.K < 15 else -2.0, seed=code_id)
    want, _, _ = oracle.decode_frames(code.K, code.R, code.G, oracle_cfg(decode_type, code.R), sym, L, threads=4)
    flat = torch.from_numpy(sym.reshape(-1))
    for off in (1, 2, 3):
        big = torch.zeros(flat.numel() + 8, dtype=flat.dtype, device="cuda")
        big[off:off + flat.numel()] = flat.cuda()
        view = big[off:off + flat.numel()].view(F, S, code.R)
        assert view.data_ptr() % 4 == (off * flat.element_size()) % 4
        for plan in plans:
            dec = BatchDecoder(table, config, plan=plan)
            assert np.array_equal(dec.decode(view, L).cpu().numpy(), want), (off, plan)


@pytest.mark.parametrize("code_id,decode_type,F,L,want", [
    (2, "SOFT16", 2048, 1024, (3, 2, 1)),    # register plan, at most one update wave per SIMD: three workspaces, two updates in flight
    (2, "HARD8", 32768, 256, (3, 2, 1)),     # the largest batch of that schedule on an MI355X (4 x 256 CUs x 32 frames)
    (2, "SOFT16", 40000, 128, (2, 1, 1)),    # up to two update waves per SIMD: chainback beside the next update
    (2, "SOFT16", 70000, 64, (2, 1, 1)),     # up to THREE update waves per SIMD (K = 7 only: 3 x 152 + the 32 registers of the LDS-ring chainback): still beside
    (2, "SOFT16", 100000, 64, (2, 1, 0)),    # larger: back to back on one stream
    (7, "SOFT16", 24, 256, (2, 1, 1)),       # K = 15 (PLAN_LDS2, update capped at 120 registers): chainback beside the next update
    ((11, 2, (0o3345, 0o3613)), "SOFT16", 40, 128, (2, 1, 1)),    # K = 11: the same
    ((13, 2, (0o10533, 0o17661)), "SOFT16", 24, 128, (2, 1, 1)),  # K = 13 (144 registers allocated, three waves per SIMD by LDS: 3 x 144 + 24 of 512): overlapped too -- the descriptor rule; 8192 x 4096: 16.4 -> 16.0 ms per batch
    ((10, 2, (0o1167, 0o1545)), "SOFT16", 200, 128, (2, 1, 1)),   # K = 10 (PLAN_LDS2 since round 5, capped at 120 registers): overlapped
    ((6, 3, (0o65, 0o57, 0o75)), "SOFT16", 200, 128, (2, 1, 0)),   # K = 6, R = 3 (PLAN_LDS as long as nobody asks for the run-time compiled register plan): back to back
    (5, "SOFT16", 40000, 64, (2, 1, 1)),     # K = 9, R = 2: two 240-register update waves leave room for the LDS-streaming chainback
    (6, "SOFT16", 40000, 64, (2, 1, 1)),     # K = 9, R = 4: 224 registers with the sub-chunk branch-metric fetch (368 before: sub-batches), 8 KiB of LDS per wave
    (3, "SOFT16", 40000, 64, (2, 1, 1)),     # K = 7, R = 3 (LTE): update capped at 240 registers: 2 x 240 + 32
    (4, "SOFT8", 40000, 64, (2, 1, 1)),      # K = 7, R = 4 (DAB): 2 x 224 + 32, eight 16 KiB update waves + two 12 KiB chainback rings per CU
])
def test_decode_pipeline_matches_serial_decode(code_id, decode_type, F, L, want):
    """vit_hip_pipeline_*: whatever schedule the library picks (two updates in flight, chainback beside the next update, or
    back to back) every batch returns the bytes of the serial vit_hip_decode_batch, including batches smaller than the
    pipeline's maximum and more batches than workspaces; the per-batch timing records line up with the submissions."""
    import ctypes as C
    import torch

    code = COMMON_CODES[code_id] if isinstance(code_id, int) else Code(f"K{code_id[0]}", *code_id)
    pc, table, config = make_table_config(code, decode_type)
    dec = BatchDecoder(table, config)
    lib = _lib.load()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if cus != 256 and code_id == 2:
        want = None                              # the thresholds scale with the CU count: only the results are checked
    pipe = C.c_void_p()
    assert lib.vit_hip_pipeline_create(dec._handle._h, F, L, C.byref(pipe)) == _lib.OK
    sch = _lib.VitHipPipelineSchedule()
    assert lib.vit_hip_pipeline_get_schedule_v2(pipe, C.byref(sch), C.sizeof(sch)) == _lib.OK
    if want is not None:
        assert (sch.workspaces, sch.update_streams, sch.chainback_overlapped) == want
    sub = int(sch.sub_batch_frames)
    if want is not None:
        assert sub == F and sch.chainback_wave_priority == (1 if sch.update_streams == 2 else 0)
    assert sch.workspace_bytes_each == dec.workspace_bytes(min(F, sub), L)
    assert lib.vit_hip_pipeline_set_timing(pipe, 1) == _lib.OK
    batches, outs = [], []
    for k in range(7):
        n = F if k != 3 else max(F // 3, 1)
        tx, sym = dec.synth(n, L, 4.0 if decode_type == "HARD8" else 2.0, seed=50 + k)
        out = torch.zeros((n, L // 8), dtype=torch.uint8, device="cuda")
        batches.append((n, sym, tx))
        outs.append(out)
    torch.cuda.synchronize()
    for (n, sym, _), out in zip(batches, outs):
        assert lib.vit_hip_pipeline_submit(pipe, C.c_void_p(sym.data_ptr()), n, C.c_void_p(out.data_ptr()), None, None) == _lib.OK
    assert lib.vit_hip_pipeline_submit(pipe, C.c_void_p(batches[0][1].data_ptr()), F + 1, C.c_void_p(outs[0].data_ptr()), None, None) == _lib.ERR_INVALID_ARG
    assert lib.vit_hip_pipeline_sync(pipe) == _lib.OK
    for (n, sym, tx), out in zip(batches, outs):
        assert torch.equal(out, dec.decode(sym, L))
    # the last batch's decision rows are readable from the pipeline's own workspace
    ws, f0, nf = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
    assert lib.vit_hip_pipeline_last_workspace(pipe, C.byref(ws), C.byref(f0), C.byref(nf)) == _lib.OK
    n_all = batches[-1][0]
    assert f0.value + nf.value == n_all and nf.value == n_all - (n_all - 1) // sub * sub     # the last sub-batch of the last batch
    n_last = min(nf.value, 64)
    got = torch.empty((n_last, L + code.K - 1, dec.W), dtype=torch.int64, device="cuda")
    assert lib.vit_hip_export_decisions(dec._handle._h, ws, n_last, L + code.K - 1, L, C.c_void_p(got.data_ptr()), None) == _lib.OK
    dec.update(batches[-1][1][f0.value:].contiguous(), L)
    assert torch.equal(got, dec.export_decisions(n_last, L))
    # timing: one record per (sub-)batch, in order, completion times non-decreasing
    n_rec_want = sum(-(-n // sub) for n, _, _ in batches)
    nrec = C.c_size_t(0)
    u, c, d = (np.zeros(32, dtype=np.float32) for _ in range(3))
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert lib.vit_hip_pipeline_get_timing(pipe, 32, p(u), p(c), p(d), C.byref(nrec)) == _lib.OK
    assert nrec.value == n_rec_want
    k = n_rec_want
    assert (u[:k] > 0).all() and (c[:k] > 0).all() and (np.diff(d[:k]) >= 0).all() and d[0] >= u[0]
    assert lib.vit_hip_pipeline_set_timing(pipe, 0) == _lib.OK
    assert lib.vit_hip_pipeline_get_timing(pipe, 0, None, None, None, C.byref(nrec)) == _lib.OK and nrec.value == 0
    assert lib.vit_hip_pipeline_destroy(pipe) == _lib.OK


@pytest.mark.parametrize("K,R,G,F", [(7, 2, (0o171, 0o133), 40000), (9, 2, (0o557, 0o663), 40000), (7, 2, (0o171, 0o133), 70000)])
def test_decode_pipeline_with_a_run_time_compiled_code(K, R, G, F):
    """the pipeline's residency rules and its small-footprint chainback kernel for a code that exists only as a run-time compiled
    module: descriptors come from the .hsaco (an offload bundle), the K = 7 chainback beside the update is the module's LDS-ring
    kernel with its dynamic LDS, and the schedule is the stock codes' (two / three update waves per SIMD + chainback: overlapped)."""
    import torch
    from viterbidecodercpp_amd import DecodePipeline

    code = Code(f"custom K{K}", K, R, tuple(G))
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config, plan=_lib.PLAN_REG)
    L = 64
    pipe = DecodePipeline(dec, F, L)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    if cus == 256:
        assert (pipe.schedule.workspaces, pipe.schedule.update_streams, pipe.schedule.chainback_overlapped) == (2, 1, 1)
        assert pipe.schedule.chainback_small_kernel == (1 if K == 7 else 0) and pipe.schedule.sub_batch_frames == F
    outs = []
    for k in range(4):
        tx, sym = dec.synth(F, L, 3.0, seed=300 + k)
        out = torch.zeros((F, L // 8), dtype=torch.uint8, device="cuda")
        pipe.submit(sym, out)
        outs.append((sym, out))
    pipe.sync()
    for sym, out in outs:
        assert torch.equal(out, dec.decode(sym, L))
    pipe.close()


def test_decode_pipeline_orders_itself_against_torch_streams():
    """DecodePipeline runs on private non-blocking streams (round-3 advisor: a data race for `sym = synth(...); submit(sym)`).
    submit() now orders each batch behind torch's current stream (vit_hip_pipeline_wait_event) and wait_done() orders the current
    stream behind the pipeline: producer -> pipeline -> consumer with NO host synchronisation in between, on a side stream, with a
    deliberately slow producer in front (a big matmul), must give the serial decode's bytes."""
    import torch
    from viterbidecodercpp_amd import DecodePipeline

    code = COMMON_CODES[2]
    pc, table, config = make_table_config(code, "SOFT16")
    dec = BatchDecoder(table, config)
    F, L = 40000, 256
    pipe = DecodePipeline(dec, F, L)
    assert pipe.schedule.chainback_overlapped == 1 and pipe.schedule.chainback_small_kernel == 1
    outs, xors = [], []
    a = torch.randn((6144, 6144), device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for k in range(4):
            b = a @ a                                           # tens of milliseconds in front of the producer
            tx, sym = dec.synth(F, L, 2.0, seed=900 + k)        # asynchronous, on `side`
            out = torch.full((F, L // 8), 0xA5, dtype=torch.uint8, device="cuda")
            pipe.submit(sym, out)                               # no synchronisation: ordered behind `side` by an event
            pipe.wait_done()                                    # `side` now waits for this batch ...
            xors.append(torch.bitwise_xor(out, tx).sum())       # ... so this consumer reads finished bytes
            outs.append((sym, out, tx))
            del b
    torch.cuda.synchronize()
    for (sym, out, tx), x in zip(outs, xors):
        ref = dec.decode(sym, L)
        assert torch.equal(out, ref)                                                       # the pipeline saw the finished symbols
        assert int(x.item()) == int(torch.bitwise_xor(ref, tx).sum().item())              # the consumer saw the finished bytes
    pipe.close()


def test_shader_clock_measurement():
    import ctypes as C

    mhz, cyc = C.c_double(0), C.c_double(0)
    assert _lib.load().vit_hip_shader_clock_mhz(0, C.byref(mhz), C.byref(cyc)) == _lib.OK
    assert 500.0 < mhz.value < 3000.0, mhz.value          # MI355X: 2400 MHz peak engine clock
    assert 1.0 < cyc.value < 16.0, cyc.value              # shader clocks per wave64 v_pk_add_u16 and SIMD, four waves resident
    light = C.c_double(0)                                  # clock only: the one-wave-per-CU probe
    assert _lib.load().vit_hip_shader_clock_mhz(0, C.byref(light), None) == _lib.OK
    assert 500.0 < light.value < 3000.0 and abs(light.value - mhz.value) < 0.25 * mhz.value, (light.value, mhz.value)


@pytest.mark.parametrize("K,R,G,plan,F,L", [
    (10, 2, (0o1167, 0o1545), _lib.PLAN_LDS, 7, 104),     # dense [F][S][W]: slab stride 113 * 8 * 8 = 7232 B, not a multiple of 256
    (6, 2, (0o65, 0o57), _lib.PLAN_LDS, 9, 40),           # W = 1: 360-byte slabs
    (7, 2, (109, 79), _lib.PLAN_REG, 70, 104),            # 32-frame tiles
    (11, 2, (0o3345, 0o3613), _lib.PLAN_LDS2, 7, 40),     # frame pairs
])
def test_slab_sub_range_export_and_chainback(oracle, K, R, G, plan, F, L):
    """vit_hip_info.workspace_tile_frames / vit_hip_workspace_slab_bytes: the rows of a slab-aligned sub-range of a batch
    are exported and chained back from the slab's address alone -- on every plan, including PLAN_LDS whose slab stride is the
    dense S*W*8 bytes of the reference layout and not the 256-byte-rounded vit_hip_workspace_bytes(h, 1, L)."""
    import ctypes as C
    import torch
    from oracle import pyoracle

    code = Code(f"K{K}", K, R, G)
    pc, table, config = make_table_config(code, "SOFT16")
    ocfg = pyoracle.stock_config(pyoracle.SOFT16, R)
    _, sym = synth.make_frames_numpy(code, pc, F, L, 1.0, seed=K)
    dec = BatchDecoder(table, config, plan=plan)
    assert dec.plan == plan
    dec.update(torch.from_numpy(sym).cuda(), L)
    lib = _lib.load()
    dec._handle.refresh()
    tile = dec._handle.info.workspace_tile_frames
    slab = lib.vit_hip_workspace_slab_bytes(dec._handle._h, L)
    assert tile >= 1 and slab > 0
    if plan == _lib.PLAN_LDS:
        assert slab == (L + K - 1) * dec.W * 8 and (K != 10 or slab % 256 != 0)
    whole = dec.export_decisions(F, L).cpu().numpy().view(np.uint64)
    for f0 in range(0, F, tile):
        n = min(tile, F - f0)
        part = dec.export_decisions(n, L, first_frame=f0).cpu().numpy().view(np.uint64)
        assert np.array_equal(part, whole[f0:f0 + n]), f0
        out = torch.zeros((n, L // 8 + (1 if L % 8 else 0)), dtype=torch.uint8, device="cuda")
        rc = lib.vit_hip_chainback_batch(dec._handle._h, C.c_void_p(dec._ws.data_ptr() + (f0 // tile) * slab), n, L,
                                         C.c_void_p(out.data_ptr()), None, None)
        assert rc == _lib.OK
        for f in range(f0, f0 + n, max(1, n // 3)):
            want = oracle.decode(K, R, G, ocfg, sym[f], L)
            assert np.array_equal(part[f - f0], want["decisions"]), f
            assert np.array_equal(out[f - f0].cpu().numpy(), want["bytes"]), f
