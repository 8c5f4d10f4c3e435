#!/usr/bin/env python3
"""How often does the reference renormalise?  One noisy frame per (code, decode type) stepped through the CPU checker ONE trellis step
at a time (oracle/, test infrastructure: this is an analysis script, not product code): the share of steps in which new_metric[0]
reached the threshold, what that makes for a 32-frame wavefront of the register plan (some frame trips: the out-of-line body runs) and
for the four-step blocks of the large-K plan (a trip after one of the first three steps: the careful routine).  -> profiles/r6_trip_rates.txt"""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from oracle import pyoracle
from viterbidecodercpp_amd import COMMON_CODES, get_decoding_config, synth
pyoracle.ensure_built()
o = pyoracle.Oracle()
for cid, dt, L in ((7,"SOFT8",1024),(7,"SOFT16",1024),(7,"HARD8",1024),(2,"SOFT8",4096),(4,"SOFT8",4096),(3,"SOFT8",4096),(5,"SOFT8",2048)):
    code = COMMON_CODES[cid]
    pc = get_decoding_config(dt, code.R)
    cfg = pyoracle.stock_config({"SOFT16":pyoracle.SOFT16,"SOFT8":pyoracle.SOFT8,"HARD8":pyoracle.HARD8}[dt], code.R)
    _, sym = synth.make_frames_numpy(code, pc, 1, L, 3.0 if dt!="HARD8" else 5.0, seed=1)
    table = o.branch_table(code.K, code.R, code.G, cfg.high, cfg.low)
    m = o.reset(code.K, code.R, cfg, 0)
    flat = np.ascontiguousarray(sym[0], dtype=cfg.soft_dtype).reshape(-1)
    S = L + code.K - 1
    trips = []
    m0 = []
    for t in range(S):
        d, r = o.update(code.K, code.R, cfg, table, m, flat[t*code.R:(t+1)*code.R])
        trips.append(1 if r else 0)   # r == 0 also when min == 0; approximate
        m0.append(int(m[0]))
    trips = np.array(trips)
    mid = sum(1 for b in range(0, S-3, 4) if trips[b:b+3].any())
    end = sum(1 for b in range(0, S-3, 4) if trips[b+3])
    print(f"{code.name} {dt}: trips/step {trips.mean():.4f} (every {1/max(trips.mean(),1e-9):.1f} steps); blocks with a mid-block trip {mid/(S/4):.3f}, end-of-block trip {end/(S/4):.3f}; P(any of 32 frames)/step ~ {1-(1-trips.mean())**32:.3f}; max m0 {max(m0)}")
