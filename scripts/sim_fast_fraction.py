#!/usr/bin/env python3
"""How often could an update WAVE (32 frames, frame A of each pair in the low halves) replace v_pk_add_u16 by v_add_u32?
v_add_u32 == v_pk_add_u16 iff no add carries out of the LOW half.  Metrics are kept biased (m ^ 0x8000), so the low half
carries exactly when a frame-A candidate sum crosses 2^15 (true value) from below.  A rigorous per-decision test has to bound
the candidates of the next U steps from what is known at the decision point: max over the lane's metrics + growth bound.
This script runs the reference recursion (numpy, K = 7 Voyager, stock SOFT16 / HARD8 configs) on the bench's synthetic frames and
reports, for waves of 16 frame pairs, the fraction of U-step windows in which EVERY frame A passes
    max_state(m_A) + (U - 1) * G + max_error < 2^15      or      min_state(m_A) >= 2^15          (G = per-step growth bound)
usage: sim_fast_fraction.py [SOFT16|HARD8] [ebn0] [frames] [L]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from viterbidecodercpp_amd import COMMON_CODES, get_decoding_config, synth

dt = sys.argv[1] if len(sys.argv) > 1 else "SOFT16"
ebn0 = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
F = int(sys.argv[3]) if len(sys.argv) > 3 else 256
L = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
code = COMMON_CODES[2]
pc = get_decoding_config(dt, code.R)
K, R, G = code.K, code.R, code.G
N = 1 << (K - 1)
tx, sym = synth.make_frames_numpy(code, pc, F, L, ebn0, seed=5)
S = sym.shape[1]
bits = 16 if dt == "SOFT16" else 8
MASK = (1 << bits) - 1
HALF = 1 << (bits - 1)
high, low = pc.soft_decision_high, pc.soft_decision_low
ME, thr = pc.soft_decision_max_error, pc.renormalisation_threshold
print(f"{dt} Eb/N0 {ebn0} dB, {F} frames x {L} bits: max_error {ME}, threshold {thr}, init non-start {pc.initial_non_start_error}")
# branch table bit of polynomial i for state pair index s (viterbi_branch_table.h): parity((s << 1) & G[i])
par = lambda v: bin(v).count("1") & 1
bt = np.array([[high if par((s << 1) & G[i]) else low for s in range(N // 2)] for i in range(R)], dtype=np.int64)   # [R][N/2]
m = np.full((F, N), pc.initial_non_start_error, dtype=np.int64)
m[:, 0] = pc.initial_start_error
mx = np.zeros((S, F), dtype=np.int64); mn = np.zeros((S, F), dtype=np.int64)
nren = 0
for t in range(S):
    mx[t] = m.max(axis=1); mn[t] = m.min(axis=1)
    y = sym[:, t, :].astype(np.int64)                                   # [F][R]
    err = np.zeros((F, N // 2), dtype=np.int64)
    for i in range(R):
        d = (bt[i][None, :] - y[:, i:i + 1])
        d = ((d + HALF) & MASK) - HALF                                   # soft_t wrap
        err += np.abs(d) & MASK
    err &= MASK
    erb = (ME - err) & MASK
    a, b = m[:, :N // 2], m[:, N // 2:]
    x0, y0 = (a + err) & MASK, (b + erb) & MASK
    x1, y1 = (a + erb) & MASK, (b + err) & MASK
    new = np.empty_like(m)
    new[:, 0::2] = np.minimum(x0, y0)
    new[:, 1::2] = np.minimum(x1, y1)
    m = new
    ren = m[:, 0] >= thr
    if ren.any():
        nren += int(ren.sum())
        m[ren] -= m[ren].min(axis=1, keepdims=True)
print(f"renormalisations per frame: {nren / F:.1f} (every {S * F / max(nren, 1):.0f} steps)")
for growth_name, Gb in (("max_error", ME), ("max_error/2", ME // 2)):
    for U in (4, 12, 24):
        nwin = S // U
        t0 = np.arange(nwin) * U
        safe = (mx[t0] + (U - 1) * Gb + ME < HALF) | (mn[t0] >= HALF)   # [nwin][F], decided from the window's first step
        # ... and the high zone must really stay a no-carry zone: min >= 2^15 at the start, metrics only fall at a renormalisation (back to zone 1 with margin)
        for pairs in (1, 16):
            w = safe[:, : (F // pairs) * pairs].reshape(nwin, -1, pairs).all(axis=2)
            print(f"  growth bound {growth_name:12s} U = {U:2d}: {pairs:2d} frame-A per wave -> fast windows {w.mean() * 100:5.1f} %")
