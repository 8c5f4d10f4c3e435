import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from viterbidecodercpp_amd import COMMON_CODES, ViterbiDecoder_Core, ViterbiDecoder_HIP, synth
from tests.helpers import make_table_config
code = COMMON_CODES[2]
pc, table, config = make_table_config(code, "SOFT16")
res = []
for L in (1024, 4096, 16384, 65536):
    S = L + code.K - 1
    _, sym = synth.make_frames_numpy(code, pc, 1, L, 3.0, seed=L)
    flat = np.ascontiguousarray(sym[0].reshape(-1))
    vitdec = ViterbiDecoder_Core(table, config)
    vitdec.set_traceback_length(L)
    ts = []
    for it in range(60):
        vitdec.reset(0)
        t0 = time.perf_counter()
        ViterbiDecoder_HIP.update(vitdec, flat)
        out = vitdec.chainback(L)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts[10:]))
    res.append((S, med))
    print(f"L={L}: median {med*1e6:.1f} us, {med/S*1e9:.1f} ns per step all-in")
for (s0, t0), (s1, t1) in zip(res, res[1:]):
    print(f"slope {s0}->{s1}: {(t1-t0)/(s1-s0)*1e9:.1f} ns per step")
