"""The single-frame latency route (csrc/kernels_one.hpp) behind vit_hip_update_host / vit_hip_chainback_host, i.e. behind the
header-level drop-in ViterbiDecoder_HIP.update() + ViterbiDecoder_Core.chainback() of ONE decoder object (the reference's
examples/run_simple.cpp:67-80; BASELINE configs[0]): one wavefront, lane == state, for every K <= 7.  Everything the reference
leaves behind is compared with the oracle: decision rows, metrics, the renormalisation sum per call, chainback bytes."""
import time

import numpy as np
import pytest

from tests.helpers import make_table_config, oracle_cfg
from viterbidecodercpp_amd import (COMMON_CODES, Code, ViterbiBranchTable, ViterbiDecoder_Config, ViterbiDecoder_Core, ViterbiDecoder_HIP,
                                   synth)

pytestmark = pytest.mark.gpu

CODES = [COMMON_CODES[0], COMMON_CODES[1], COMMON_CODES[2], COMMON_CODES[3], COMMON_CODES[4],
         Code("K2", 2, 2, (0b11, 0b11)), Code("K4 R3", 4, 3, (0o15, 0o17, 0o13)), Code("K6 R1", 6, 1, (0o65,)),
         Code("K7 R1", 7, 1, (0o171,)),          # (K = 7, R <= 4: the in-place kernel with its helper wavefronts; R = 6, 8 and K < 7: the lane == state one)
         Code("K7 R6", 7, 6, (0o171, 0o133, 0o165, 0o117, 0o135, 0o157)), Code("K7 R8", 7, 8, (0o171, 0o133, 0o165, 0o117, 0o135, 0o157, 0o145, 0o173))]


@pytest.mark.parametrize("decode_type", ["SOFT16", "SOFT8", "HARD8"])
@pytest.mark.parametrize("code", CODES, ids=lambda c: c.name.replace(" ", "_"))
def test_single_frame_route_matches_the_oracle(oracle, code, decode_type):
    """whole-frame update, then the same frame in ragged pieces (1, 63, 64, 65, 200 ... steps: the kernel works in groups of 64
    steps), start and end states, a trace length that is no multiple of 8, frames longer than the chainback's 8192-step LDS chunk"""
    pc, table, config = make_table_config(code, decode_type)
    cfg = oracle_cfg(decode_type, code.R)
    rng = np.random.default_rng(code.K * 10 + code.R)
    for L, pieces in ((1300, None), (777, (1, 63, 64, 65, 200, 7, 128)), (20011, (5000, 9000)), (5, None)):
        # (the generator wants whole bytes: the symbol stream of a frame of ceil8(L) bits, cut after L + K-1 steps)
        S = L + code.K - 1
        _, sym = synth.make_frames_numpy(code, pc, 1, (L + 7) // 8 * 8, 2.0, seed=L)
        sym = np.ascontiguousarray(sym[:, :S])
        flat = sym[0].reshape(-1)
        ss, es = int(rng.integers(0, code.num_states)), int(rng.integers(0, code.num_states))
        want = oracle.decode(code.K, code.R, code.G, cfg, sym[0], L, start_state=ss, end_state=es)
        vitdec = ViterbiDecoder_Core(table, config)
        vitdec.set_traceback_length(L)
        vitdec.reset(ss)
        acc, t = 0, 0
        if pieces is None:
            acc += ViterbiDecoder_HIP.update(vitdec, flat)
        else:
            k = 0
            while t < S:
                n = min(pieces[k % len(pieces)], S - t)
                acc += ViterbiDecoder_HIP.update(vitdec, flat[t * code.R:(t + n) * code.R])
                t += n
                k += 1
        acc += vitdec.take_unreported_renormalisation()
        assert vitdec.get_error(es) == want["error"], (L, pieces)
        acc += vitdec.take_unreported_renormalisation()
        assert acc == want["renorm_sum"], (L, pieces)
        assert np.array_equal(vitdec.m_metrics.astype(np.uint32), want["metrics"])
        assert np.array_equal(np.asarray(vitdec.m_decisions).reshape(S, -1), want["decisions"].reshape(S, -1)), (L, pieces)
        assert np.array_equal(vitdec.chainback(L, es), want["bytes"]), (L, pieces)


def test_single_frame_latency_of_the_drop_in_route(oracle):
    """BASELINE configs[0]'s call pattern on one 8192-bit Voyager frame: reset -> update -> chainback through the header-level
    drop-in in ONE launch (round 6: vit_hip_update_host_lazy; round 5: 0.76 ms with two launches and four staging copies; round 4:
    3.3 ms through the general LDS kernel; the reference's scalar decoder needs 0.68 ms for the same frame on one core of the box, its
    AVX2 strategy 0.05 ms).  8198 dependent steps of ~72 ns are 0.59 ms: the bound leaves a slow box a margin."""
    code = COMMON_CODES[2]
    pc, table, config = make_table_config(code, "SOFT16")
    L = 8192
    tx, sym = synth.make_frames_numpy(code, pc, 1, L, 3.0, seed=3)
    flat = sym[0].reshape(-1)
    vitdec = ViterbiDecoder_Core(table, config)
    vitdec.set_traceback_length(L)
    times = []
    for rep in range(8):
        vitdec.reset()
        t0 = time.perf_counter()
        ViterbiDecoder_HIP.update(vitdec, flat)
        out = vitdec.chainback(L)
        times.append(time.perf_counter() - t0)
    assert np.array_equal(out, tx[0])
    best = sorted(times[2:])[len(times[2:]) // 2]
    print(f"single 8192-bit K7 frame, update + chainback through the drop-in: median {best * 1e3:.3f} ms")
    # (round 6, second session: the in-place kernel: 8198 steps of ~30 ns + ~35 us of launch, chainback and polling = 0.28 - 0.30 ms)
    assert best < 0.45e-3, times


@pytest.mark.parametrize("R", [1, 2, 3, 4])
@pytest.mark.parametrize("width", [2, 1])
def test_in_place_single_frame_kernel_fuzzed(oracle, R, width):
    """the K = 7 single-frame kernel (csrc/kernels_one.hpp: one_update7_body) under what its shortcuts must survive: thresholds of 0
    ("always"), of a few steps' error (a renormalisation every two to six steps: the late test, and blocks that are never "safe"),
    of the type's maximum; full-range garbage symbols (the block's largest branch metric then says nothing can be skipped, and the
    16-bit sums wrap); calls cut into pieces of every length around the 32-step blocks and the 96-step unrolled period; start and
    end states.  Decision rows, metrics, renormalisation sums, error and bytes against the oracle."""
    from tests.test_gpu_fuzz import random_config
    from viterbidecodercpp_amd import ViterbiBranchTable, ViterbiDecoder_Config

    rng = np.random.default_rng(7000 + 10 * R + width)
    G = tuple(int(g) | 1 | 64 for g in rng.integers(0, 128, R))
    sdt, edt = (np.int16, np.uint16) if width == 2 else (np.int8, np.uint8)
    lim = 1 << (8 * width - 1)
    for trial in range(12):
        cfg = random_config(rng, width, trial % 4)
        if trial >= 8:          # a threshold a few steps' error above the start: state 0 crosses it every two to six steps
            M = (1 << (8 * width)) - 1
            cfg = type(cfg)(width, width, cfg.high, cfg.low, (cfg.high - cfg.low) * R & M, 0, min(M, 3 * ((cfg.high - cfg.low) * R & M)),
                            min(M, 5 * ((cfg.high - cfg.low) * R & M) + 1))
        table = ViterbiBranchTable(7, R, G, cfg.high, cfg.low, sdt)
        config = ViterbiDecoder_Config(cfg.max_error, cfg.initial_start_error, cfg.initial_non_start_error, cfg.renormalisation_threshold, edt)
        L = int(rng.integers(1, 700))
        S = L + 6
        if trial % 2:
            sym = rng.integers(cfg.low, cfg.high + 1, size=(S, R)).astype(sdt)
        else:
            sym = rng.integers(-lim, lim, size=(S, R)).astype(sdt)
        ss, es = int(rng.integers(0, 64)), int(rng.integers(0, 64))
        want = oracle.decode(7, R, G, cfg, sym, L, start_state=ss, end_state=es)
        vitdec = ViterbiDecoder_Core(table, config)
        vitdec.set_traceback_length(L)
        vitdec.reset(ss)
        flat = np.ascontiguousarray(sym.reshape(-1))
        acc, t = 0, 0
        pieces = (S,) if trial % 3 == 0 else (31, 32, 33, 1, 95, 96, 97, 64, 5)
        k = 0
        while t < S:
            n = min(pieces[k % len(pieces)], S - t)
            acc += ViterbiDecoder_HIP.update(vitdec, flat[t * R:(t + n) * R])
            t += n
            k += 1
        acc += vitdec.take_unreported_renormalisation()
        tag = (R, width, trial, L, cfg)
        assert vitdec.get_error(es) == want["error"], tag
        acc += vitdec.take_unreported_renormalisation()
        assert acc == want["renorm_sum"], tag
        assert np.array_equal(vitdec.m_metrics.astype(np.uint32), want["metrics"]), tag
        assert np.array_equal(np.asarray(vitdec.m_decisions).reshape(S, -1), want["decisions"].reshape(S, -1)), tag
        out = vitdec.chainback(L, es)
        assert np.array_equal(out, want["bytes"]), tag


@pytest.mark.parametrize("K", [7, 5, 3, 2])
def test_parallel_chainback_is_exact_when_its_guesses_are_wrong(oracle, K):
    """The single-frame chainback chases 64 segments of the frame at once from GUESSED top states and repairs the wrong guesses.
    Real decision rows make every guess right (survivor paths merge within a few constraint lengths); rows of random bits keep
    paths apart, so here the repair pass does the work -- and rows that decide 'predecessor 0' everywhere or repeat with a short
    period stress the boundaries.  Lengths: ragged top byte, several 8192-step chunks, frames shorter than 64 segments."""
    import ctypes as C
    from viterbidecodercpp_amd import _lib

    code = Code(f"K{K}", K, 2, ((1 << K) - 1, (1 << (K - 1)) | 1))
    pc, table, config = make_table_config(code, "SOFT16")
    vitdec = ViterbiDecoder_Core(table, config)           # only its handle is used: the rows below are not a decoder's
    lib = _lib.load()
    rng = np.random.default_rng(K)
    N = code.num_states
    for L in (8192, 8191, 20000, 16389, 517, 64, 9, 1):
        S = L + K - 1
        for kind in ("random", "zeros", "period3"):
            if kind == "random":
                rows = rng.integers(0, 1 << 63, size=S, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=S).astype(np.uint64)
            elif kind == "zeros":
                rows = np.zeros(S, dtype=np.uint64)
            else:
                rows = np.tile(rng.integers(0, 1 << 62, size=3, dtype=np.uint64), S // 3 + 1)[:S].copy()
            if N < 64:
                rows &= np.uint64((1 << N) - 1)
            for es in (0, N - 1, int(rng.integers(0, N))):
                want = oracle.chainback(K, rows.reshape(S, 1), L, es)
                out = np.zeros((L + 7) // 8, dtype=np.uint8)
                assert lib.vit_hip_chainback_host(vitdec._handle._h, rows.ctypes.data_as(C.c_void_p), L, es, out.ctypes.data_as(C.c_void_p)) == _lib.OK
                assert np.array_equal(out, want), (L, kind, es)


@pytest.mark.parametrize("code,decode_type", [(COMMON_CODES[2], "SOFT16"), (COMMON_CODES[1], "SOFT8"), (COMMON_CODES[4], "HARD8"), (COMMON_CODES[5], "SOFT16")],
                         ids=lambda x: x.name.replace(" ", "_") if hasattr(x, "name") else x)
def test_frame_route_keeps_rows_on_the_device(oracle, code, decode_type):
    """update() leaves its decision rows in the handle's device row store (vit_hip_update_host_lazy) and the call that completes a
    frame also chains it back: chainback(traceback length, end state 0) right behind it is a memcpy; any other (bits, end state)
    runs one kernel over the device rows; reading m_decisions brings the rows home, bit for bit the reference's, and hands authority
    back to the host -- rows CHANGED there are what the next chainback follows.  K = 9: the same bookkeeping over the LDS plan."""
    pc, table, config = make_table_config(code, decode_type)
    cfg = oracle_cfg(decode_type, code.R)
    L = 1544
    S = L + code.K - 1
    N = code.num_states
    vitdec = ViterbiDecoder_Core(table, config)
    vitdec.set_traceback_length(L)
    for frame, pieces in enumerate((None, (700, 65, 200), None)):
        _, sym = synth.make_frames_numpy(code, pc, 1, L, 2.0, seed=50 + frame)
        flat = sym[0].reshape(-1)
        want0 = oracle.decode(code.K, code.R, code.G, cfg, sym[0], L)
        vitdec.reset()
        acc, t, k = 0, 0, 0
        while t < S:
            n = S - t if pieces is None else min(pieces[k % len(pieces)], S - t)
            acc += ViterbiDecoder_HIP.update(vitdec, flat[t * code.R:(t + n) * code.R])
            t, k = t + n, k + 1
        assert acc == want0["renorm_sum"] and vitdec.rows_on_device == S
        assert np.array_equal(vitdec.chainback(L), want0["bytes"])                       # decoded by the completing update
        assert vitdec.rows_on_device == S                                                # nothing came home for that
        es = N - 1
        assert np.array_equal(vitdec.chainback(L, es), oracle.chainback(code.K, want0["decisions"], L, es))          # another end state: one kernel
        assert np.array_equal(vitdec.chainback(L - 11, 3 % N), oracle.chainback(code.K, want0["decisions"], L - 11, 3 % N))   # fewer bits, ragged
        assert vitdec.get_error(0) == want0["error"] and vitdec.rows_on_device == S
    rows = vitdec.m_decisions                                                            # rows come home; the array is the caller's to change
    assert vitdec.rows_on_device == 0
    assert np.array_equal(np.asarray(rows).reshape(S, -1), want0["decisions"].reshape(S, -1))
    rng = np.random.default_rng(7)
    rows[100:900] ^= rng.integers(0, 1 << 62, size=rows[100:900].shape, dtype=np.uint64) & np.uint64((1 << min(N, 63)) - 1)
    assert np.array_equal(vitdec.chainback(L, 1), oracle.chainback(code.K, np.asarray(rows), L, 1))
    # and the decoder goes on: the next frame is lazy again
    vitdec.reset()
    ViterbiDecoder_HIP.update(vitdec, flat)
    assert vitdec.rows_on_device == S and np.array_equal(vitdec.chainback(L), want0["bytes"])
