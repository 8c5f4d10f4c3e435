#!/usr/bin/env python3
"""per-kernel sums of the counter passes of scripts/pmc_codes.sh: usage pmc_codes_summary.py gpurun_out/pmc_<tag>"""
import csv, glob, os, re, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int)); dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "update_kernel" not in k and "chainback" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k in acc: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
def short(k):
    m = re.search(r"(reg_\w+|lds2_\w+)<.*?RegSpec<(\d+), (\d+)", k)
    return f"{m.group(1)} K{m.group(2)}R{m.group(3)}" if m else k[:60]
names = sorted(acc, key=short)
cols = ["SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_LDS",
        "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VMEM", "SQ_INSTS_SALU"]
print("per LAUNCH (sum over the launches / number of launches); ratios are per wave cycle")
for k in names:
    a = {c: acc[k][c] / max(n[k][c], 1) for c in acc[k]}
    wc = a.get("SQ_WAVE_CYCLES", 0) or 1
    d = sorted(dur[k])[len(dur[k]) // 2] if dur[k] else float("nan")
    print(f"\n{short(k)}: median duration {d:.3f} ms over {len(dur[k])} launches; waves {a.get('SQ_WAVES', 0):.0f}")
    print(f"  VALU insts {a.get('SQ_INSTS_VALU', 0):.4g} ({a.get('SQ_INSTS_VALU', 0) / max(a.get('SQ_WAVES', 1), 1):.1f} per wave)  LDS insts {a.get('SQ_INSTS_LDS', 0):.4g}  SALU {a.get('SQ_INSTS_SALU', 0):.4g}  VMEM rd/wr {a.get('SQ_INSTS_VMEM_RD', 0):.4g}/{a.get('SQ_INSTS_VMEM_WR', 0):.4g}")
    print(f"  of wave cycles: VALU active {a.get('SQ_ACTIVE_INST_VALU', 0) / wc:.3f}  wait_any {a.get('SQ_WAIT_ANY', 0) / wc:.3f}  wait_inst_any {a.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}"
          f"  LDS active {a.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f}  wait_inst_lds {a.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}  VMEM active {a.get('SQ_ACTIVE_INST_VMEM', 0) / wc:.3f}")
    if a.get("SQ_LDS_IDX_ACTIVE"): print(f"  LDS bank conflict cycles / LDS active cycles: {a['SQ_LDS_BANK_CONFLICT'] / a['SQ_LDS_IDX_ACTIVE']:.3f}  (LDS_IDX_ACTIVE {a['SQ_LDS_IDX_ACTIVE']:.4g}, BUSY_CYCLES {a.get('SQ_BUSY_CYCLES', 0):.4g})")

# ---- machine-readable: VALU wave-instructions per wave and trellis step of every update kernel (scripts/matrix.py prices the
# full-size launches of the benchmark matrix with it: instructions scale with waves x steps) ----
import json
meta = {}
for line in open(os.path.join(root, "sq1.log")) if os.path.exists(os.path.join(root, "sq1.log")) else []:
    t = line.split()
    if len(t) >= 6 and t[-2] == "BER":
        meta[(t[0] if len(t) == 6 else " ".join(t[:-5]), t[-5])] = (int(t[-4]), int(t[-3]))       # (code name, decode type) -> (frames, L)
out = {}
for k in names:
    if "update_kernel" not in k: continue
    a = {c: acc[k][c] / max(n[k][c], 1) for c in acc[k]}
    m = re.search(r"RegSpec<(\d+), (\d+),.*?>, (\d+)>", k) or re.search(r"lds2_update_kernel\w*<(\d+), (\d+), (\d+)>", k)
    if not m: continue
    if "RegSpec" in k: K, R, shift = int(m.group(1)), int(m.group(2)), int(m.group(3))
    else: K, shift, R = int(m.group(1)), int(m.group(2)), int(m.group(3))
    out[f"K{K}|R{R}|shift{shift}|{'reg' if 'RegSpec' in k else 'lds2'}"] = {"kernel": k, "valu_insts": a.get("SQ_INSTS_VALU"), "waves": a.get("SQ_WAVES"),
                                                                            "lds_insts": a.get("SQ_INSTS_LDS"), "wave_cycles": a.get("SQ_WAVE_CYCLES"),
                                                                            "valu_active": a.get("SQ_ACTIVE_INST_VALU"), "wait_any": a.get("SQ_WAIT_ANY")}
json.dump({"frames_L_by_code": {f"{c}|{t}": v for (c, t), v in meta.items()}, "update_kernels": out}, open(os.path.join(root, "valu.json"), "w"), indent=1)
