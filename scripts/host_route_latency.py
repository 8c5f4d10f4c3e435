#!/usr/bin/env python3
"""Latency of the header-level drop-in route (ViterbiDecoder_HIP.update / ViterbiDecoder_Core.chainback over
vit_hip_update_host / vit_hip_chainback_host): one decoder object, host-resident state, one kernel launch per call.
Prints: one-shot update() of a whole frame, streaming update() with N = R symbols per call (queued on the host and run in one
launch per 2048 steps: the Python figure is interpreter overhead, the C++ header's own cost is printed by
tests/cpp/run_stream_hip, run at the end), chainback()."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from viterbidecodercpp_amd import (COMMON_CODES, ViterbiBranchTable, ViterbiDecoder_Config, ViterbiDecoder_Core, ViterbiDecoder_HIP,
                                   get_decoding_config, synth)

for code_id, L in ((2, 8192), (5, 8192), (7, 1024)):
    code = COMMON_CODES[code_id]
    pc = get_decoding_config("SOFT16", code.R)
    table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    vitdec = ViterbiDecoder_Core(table, ViterbiDecoder_Config.from_decoder_config(pc))
    tx, sym = synth.make_frames_numpy(code, pc, 1, L, 3.0, seed=3)
    sym = sym[0].reshape(-1)
    vitdec.set_traceback_length(L)
    times = []
    for rep in range(4):
        vitdec.reset()
        t0 = time.perf_counter(); ViterbiDecoder_HIP.update(vitdec, sym); t1 = time.perf_counter()
        out = vitdec.chainback(L); t2 = time.perf_counter()
        times.append((t1 - t0, t2 - t1))
    assert np.array_equal(out, tx[0]) or code.K == 15
    one, cb = np.median([t[0] for t in times[1:]]), np.median([t[1] for t in times[1:]])
    n_stream = L + code.K - 1                         # the whole frame, one trellis step per call: every queue flush is inside
    vitdec.reset()
    t0 = time.perf_counter()
    for t in range(n_stream):
        ViterbiDecoder_HIP.update(vitdec, sym[t * code.R:(t + 1) * code.R])
    vitdec.flush_pending()
    per_call = (time.perf_counter() - t0) / n_stream
    print(f"{code.name} K={code.K} L={L}: one-shot update {one*1e3:.2f} ms ({L/one/1e6:.1f} Mbit/s), chainback {cb*1e3:.2f} ms, "
          f"streaming update(N=R) {per_call*1e6:.2f} us per call = {1/per_call/1e3:.1f} kbit/s")

import subprocess
exe = os.path.join(ROOT, "tests", "cpp", "run_stream_hip")
if os.access(exe, os.X_OK):
    print("C++ header drop-in:", subprocess.run([exe], capture_output=True, text=True).stdout.strip().splitlines()[0])
