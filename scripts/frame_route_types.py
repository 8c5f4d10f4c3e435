#!/usr/bin/env python3
"""ns per trellis step of the single-frame route for the stock K = 7 codes and decode types (update + chainback of one 8192-bit frame, median of 40)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viterbidecodercpp_amd import COMMON_CODES, ViterbiDecoder_Core, ViterbiDecoder_HIP, synth
from tests.helpers import make_table_config
L = 8192
for ci in (2, 3, 4):
    code = COMMON_CODES[ci]
    for dt in ("SOFT16", "SOFT8", "HARD8"):
        pc, table, config = make_table_config(code, dt)
        _, sym = synth.make_frames_numpy(code, pc, 1, L, 5.0 if dt == "HARD8" else 3.0, seed=L)
        flat = np.ascontiguousarray(sym[0].reshape(-1))
        vitdec = ViterbiDecoder_Core(table, config)
        vitdec.set_traceback_length(L)
        ts = []
        for it in range(50):
            vitdec.reset(0)
            t0 = time.perf_counter()
            ViterbiDecoder_HIP.update(vitdec, flat)
            vitdec.chainback(L)
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts[10:]))
        print(f"{code.name:12s} {dt:6s}: {med * 1e6:7.1f} us per 8192-bit frame, {med / (L + 6) * 1e9:5.1f} ns per step all-in")
