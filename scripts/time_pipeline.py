#!/usr/bin/env python3
"""time the decode pipeline (vit_hip_pipeline_*) for ANY code; schedule experiments through vit_hip_pipeline_create_ex.
usage: time_pipeline.py K R G0,G1[,..] decode_type frames L [steps] [option=value ...]   (polynomials in decimal or 0o.. octal)
options (vit_hip_pipeline_options fields): chainback_overlap= update_streams= sub_batches= workspaces= chainback_small_kernel=
chainback_wave_priority=      e.g.  time_pipeline.py 7 3 91,117,121 SOFT16 65536 8192 12 sub_batches=1"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from viterbidecodercpp_amd import BatchDecoder, DecodePipeline, ViterbiBranchTable, ViterbiDecoder_Config, _lib, get_decoding_config
from viterbidecodercpp_amd.codes import Code

K, R = int(sys.argv[1]), int(sys.argv[2])
G = tuple(int(x, 0) for x in sys.argv[3].split(","))
dt, F, L = sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
steps = int(sys.argv[7]) if len(sys.argv) > 7 and "=" not in sys.argv[7] else 12
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[7:] if "=" in a}
code = Code(f"K{K}", K, R, G)
pc = get_decoding_config(dt, R)
table = ViterbiBranchTable(K, R, G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
dec = BatchDecoder(table, ViterbiDecoder_Config.from_decoder_config(pc))
if dec.plan == _lib.PLAN_LDS and "PLAN_REG" in dec.plan_note.split(";", 1)[-1]:
    dec.set_plan(_lib.PLAN_REG)          # the run-time compiled register plan (what vit_hip_plan_note points to)
tx, sym = dec.synth(F, L, 3.0, seed=1)
out = torch.empty((F, L // 8), dtype=torch.uint8, device="cuda")
pipe = DecodePipeline(dec, F, L, options=_lib.VitHipPipelineOptions(**kw) if kw else None)
s = pipe.schedule
for _ in range(3):
    pipe.submit(sym, out)
pipe.sync()
pipe.set_timing(True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    pipe.submit(sym, out)
pipe.sync()
dtm = (time.perf_counter() - t0) / steps * 1e3
u, c, d = pipe.timing()
ber = int(dec.count_bit_errors(out, tx).item()) / float(F * L)
print(f"K{K} R{R} {dt} {F}x{L} [{_lib.PLAN_NAMES[dec.plan]}]: {dtm:.3f} ms per batch = {F * L / dtm / 1e6:.2f} Gbit/s; update {np.median(u):.3f} chainback {np.median(c):.3f} ms; "
      f"schedule ws={s.workspaces} upd={s.update_streams} overlapped={s.chainback_overlapped} sub={s.sub_batch_frames}; BER {ber:.2e}")
