#!/usr/bin/env python3
"""experiment: TWO update kernels in flight (three decision workspaces, two update streams, one high-priority chainback
stream) against the bench's one-update pipeline.  usage: exp_pipeline3.py [frames] [steps] [n_update_streams] [decode_type] [ebn0]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from viterbidecodercpp_amd import COMMON_CODES, BatchDecoder, ViterbiBranchTable, ViterbiDecoder_Config, get_decoding_config

F = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
NU = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L = 8192
code = COMMON_CODES[2]
DT = sys.argv[4] if len(sys.argv) > 4 else "SOFT16"
EBN0 = float(sys.argv[5]) if len(sys.argv) > 5 else 3.0
pc = get_decoding_config(DT, code.R)
table = ViterbiBranchTable(code.K, code.R, code.G, pc.soft_decision_high, pc.soft_decision_low, pc.soft_dtype)
dec = BatchDecoder(table, ViterbiDecoder_Config.from_decoder_config(pc))
tx, sym = dec.synth(F, L, EBN0, seed=1)
NW = NU + 1
ws = [dec.new_workspace(F, L) for _ in range(NW)]
outs = [torch.empty((F, L // 8), dtype=torch.uint8, device="cuda") for _ in range(NW)]
lo, hi = torch.cuda.Stream.priority_range()
s_u = [torch.cuda.Stream(priority=lo) for _ in range(NU)]
s_cb = torch.cuda.Stream(priority=hi)
cb_done = [torch.cuda.Event() for _ in range(NW)]
upd_done = [torch.cuda.Event() for _ in range(NW)]


def run(n):
    for i in range(n):
        w = i % NW
        su = s_u[i % NU]
        su.wait_event(cb_done[w])
        with torch.cuda.stream(su):
            dec.update(sym, L, want_metrics=False, workspace=ws[w])
            upd_done[w].record(su)
        s_cb.wait_event(upd_done[w])
        with torch.cuda.stream(s_cb):
            dec.chainback(F, L, out=outs[w], workspace=ws[w])
            cb_done[w].record(s_cb)


for e in cb_done:
    e.record()
run(6)
torch.cuda.synchronize()
t = time.perf_counter()
run(steps)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / steps
ok = all(torch.equal(o, tx) or int(dec.count_bit_errors(o, tx).item()) < F * L * 1e-2 for o in outs)
print(f"{DT}: {NU} update stream(s), {NW} workspaces, {F} frames: {dt*1e3:.3f} ms per batch = {F*L/dt/1e9:.1f} Gbit/s  outputs sane: {ok}")
