#!/usr/bin/env python3
"""time the update / chainback kernels of one configuration alone (HIP events, median of N): experiments on kernel variants.
usage: time_update.py code decode_type frames L [reps]      (VIT_HIP_LIB_PATH selects another build of the library)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from viterbidecodercpp_amd import COMMON_CODES, BatchDecoder, ViterbiBranchTable, ViterbiDecoder_Config, get_decoding_config

code = COMMON_CODES[int(sys.argv[1])]; dt = sys.argv[2]; F = int(sys.argv[3]); L = int(sys.argv[4]); reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
pc = get_decoding_config(dt, code.R)
table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
dec = BatchDecoder(table, ViterbiDecoder_Config.from_decoder_config(pc))
tx, sym = dec.synth(F, L, 3.0, seed=1)
out = torch.empty((F, L // 8), dtype=torch.uint8, device="cuda")
upd, cb = [], []
for r in range(reps + 1):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); dec.update(sym, L, want_metrics=False); e[1].record(); dec.chainback(F, L, out=out); e[2].record()
    torch.cuda.synchronize()
    if r: upd.append(e[0].elapsed_time(e[1])); cb.append(e[1].elapsed_time(e[2]))
ber = int(dec.count_bit_errors(out, tx).item()) / float(F * L)
print(f"{os.environ.get('VIT_HIP_LIB_PATH', 'product')}: {code.name} {dt} {F}x{L}: update {np.median(upd):.3f} ms  chainback {np.median(cb):.3f} ms  BER {ber:.2e}")
