#!/usr/bin/env python3
"""One update + one chainback launch (alone, back to back) of each listed stock code: the program scripts/pmc_codes.sh profiles.
usage: pmc_codes.py frames L code[:decode_type] ...        e.g. pmc_codes.py 65536 2048 2 3 4 5 6"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from viterbidecodercpp_amd import COMMON_CODES, BatchDecoder, ViterbiBranchTable, ViterbiDecoder_Config, get_decoding_config

F, L = int(sys.argv[1]), int(sys.argv[2])
for spec in sys.argv[3:]:
    ci, _, dt = spec.partition(":")
    dt = dt or "SOFT16"
    code = COMMON_CODES[int(ci)]
    pc = get_decoding_config(dt, code.R)
    table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
    dec = BatchDecoder(table, ViterbiDecoder_Config.from_decoder_config(pc))
    f = F if code.K < 11 else max(2, F // 16)
    tx, sym = dec.synth(f, L, 3.0, seed=1)
    out = torch.empty((f, L // 8), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        dec.update(sym, L, want_metrics=False)
        dec.chainback(f, L, out=out)
    torch.cuda.synchronize()
    print(code.name, dt, f, L, "BER", int(dec.count_bit_errors(out, tx).item()) / float(f * L), flush=True)
    del dec, sym, out, tx
    torch.cuda.empty_cache()
