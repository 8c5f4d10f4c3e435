#!/usr/bin/env python3
"""Shader clock WHILE the decode kernels run: vit_hip_shader_clock_mhz (s_memtime over s_memrealtime inside its own one-wave-
per-CU kernel on the null stream) is called with batches in flight on the pipeline's streams; prints (MHz, probe ms).  usage: clock_under_load.py <code> <type> <frames> <bits>"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from viterbidecodercpp_amd import _lib, COMMON_CODES, BatchDecoder, ViterbiBranchTable, ViterbiDecoder_Config, get_decoding_config
from viterbidecodercpp_amd.decoder import DecodePipeline

code = COMMON_CODES[int(sys.argv[1])]; dt = sys.argv[2]; F = int(sys.argv[3]); L = int(sys.argv[4])
pc = get_decoding_config(dt, code.R)
table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
dec = BatchDecoder(table, ViterbiDecoder_Config.from_decoder_config(pc))
lib = _lib.load()
tx, sym = dec.synth(F, L, 3.0, seed=1)
out = torch.zeros((F, L // 8), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()

def clock():
    mhz = C.c_double(0)
    t0 = time.perf_counter()
    assert lib.vit_hip_shader_clock_mhz(0, C.byref(mhz), None) == _lib.OK       # the light probe: one wave per CU
    return round(mhz.value), round((time.perf_counter() - t0) * 1e3, 2)

for _ in range(3): clock()
print(code.name, dt, F, L)
print("idle            ", [clock() for _ in range(3)])
pipe = DecodePipeline(dec, F, L)
for rep in range(2):
    for _ in range(12): pipe.submit(sym, out)
    time.sleep(0.02)
    c = [clock() for _ in range(3)]
    pipe.sync()
    print("pipeline busy   ", c)
s = torch.cuda.Stream()
for rep in range(2):
    with torch.cuda.stream(s):
        for _ in range(12): dec.update(sym, L, want_metrics=False)
    time.sleep(0.02)
    c = [clock() for _ in range(3)]
    torch.cuda.synchronize()
    print("update only busy", c)
