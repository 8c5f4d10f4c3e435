#!/usr/bin/env python3
"""Condense gpurun_out/prof_<tag>/ (scripts/profile.sh) into profiles/<tag>_summary.md + profiles/<tag>_kernel_stats.csv and
name the entry of profiles/traffic.json it feeds (scripts/make_traffic.py writes that file from the summaries).

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are in KiB (x1024); on gfx950
FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced streaming reads, so the read side is doubled
("corrected"); WRITE_SIZE is exact for 16-byte-per-lane stores.  Both the raw and corrected numbers are printed.
usage: summarize_profile.py <tag> [<traffic-key>]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else None
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

KERNELS = ("reg_update_kernel", "reg_chainback_kernel", "reg_chainback_alt_kernel", "reg_export_kernel", "lds_update_kernel", "lds_chainback_kernel")


def short(name):
    name = name.replace("void ", "")
    for k in KERNELS:
        if k in name:
            return "vit::" + k + name.split(k, 1)[1].split("(")[0]
    return name.split("(")[0][:60]


lines = [f"# rocprofv3 summary `{tag}`", "",
         "Command per pass: `rocprofv3 <pass flags> --output-format csv -- python3 bench.py --steps " + os.environ.get("PROFILE_STEPS", "20") + " --warmup " + os.environ.get("PROFILE_WARMUP", "10") + " --no-cpu-baseline " + " ".join(sys.argv[3:]) + "`",
         "(scripts/profile.sh; rocprofv3's own averages include the warm-up launches).", ""]

# ---- pass 1: kernel stats ----
def newest(pattern):
    """gpurun merges every run into the same directory: keep only the most recent file of a pass"""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
        w.writerow(cols)
        for r in rows:
            w.writerow([r[c] for c in cols])
    lines += ["## --kernel-trace --stats (vit:: kernels)", "", "| kernel | calls | avg ms | min ms | max ms | % of GPU time |",
              "|---|---|---|---|---|---|"]
    for r in rows:
        if "vit::" in r["Name"]:
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['MinNs'])/1e6:.3f} | "
                         f"{float(r['MaxNs'])/1e6:.3f} | {r['Percentage']} |")
    lines.append("")
    # the same pass launch by launch: rocprofv3's average spans the warm-up launches, during which the card is still ramping
    # its clocks (the first update launch takes about twice as long); the timed region of bench.py is the LAST `steps` launches
    tr = newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))
    if tr:
        per = defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            if "vit::" in r["Kernel_Name"]:
                per[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        steps = int(os.environ.get("PROFILE_STEPS", "20"))
        warm = int(os.environ.get("PROFILE_WARMUP", "10"))
        lines += [f"Launch by launch (same pass; of each kernel's launches bench.py times {steps} after {warm} warm-up ones; those behind them are the untimed batches of its clock-under-load probe):", "",
                  "| kernel | launches | avg ms, timed launches only | median ms | first launch ms |", "|---|---|---|---|---|"]
        for k, v in per.items():
            if len(v) >= warm + steps:
                t = sorted(v[warm:warm + steps])
                lines.append(f"| `{k}` | {len(v)} | {sum(t)/len(t):.3f} | {t[len(t)//2]:.3f} | {v[0]:.3f} |")
        lines.append("")

# ---- PMC passes ----
counters = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> [values per dispatch]
meta = {}
pmc_files = []
for pass_dir in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if os.path.isdir(pass_dir):
        pmc_files += newest(os.path.join(pass_dir, "*", "*_counter_collection.csv"))
for path in pmc_files:
    for r in csv.DictReader(open(path)):
        if "vit::" not in r["Kernel_Name"]:
            continue
        k = short(r["Kernel_Name"])
        counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = dict(vgpr=r["VGPR_Count"], agpr=r["Accum_VGPR_Count"], sgpr=r["SGPR_Count"], lds=r["LDS_Block_Size"],
                       scratch=r["Scratch_Size"], grid=r["Grid_Size"], wg=r["Workgroup_Size"])

traffic = {}
for k in sorted(counters):
    c = {n: sum(v) / len(v) for n, v in counters[k].items()}
    m = meta[k]
    lines += [f"## PMC, per launch: `{k}`", "",
              # rocprofv3's VGPR_Count / Accum_VGPR_Count are HALVES of the unified 512-entry register file a wave of this kernel
              # allocates (K9 update: 120 = 240 / 2).  Printed doubled, which is the number that decides co-residency -- round 3
              # printed the raw 132 of a chainback kernel thought to hold 22 registers (it allocated 264)
              f"grid {m['grid']} threads, workgroup {m['wg']}, registers ALLOCATED per wave {2 * (int(m['vgpr']) + int(m['agpr']))} "
              f"(rocprofv3 VGPR_Count {m['vgpr']} + Accum {m['agpr']}, in halves of the unified file), SGPR {m['sgpr']}, "
              f"LDS {m['lds']} B, scratch {m['scratch']} B", "", "| counter | value per launch |", "|---|---|"]
    for n in sorted(c):
        lines.append(f"| {n} | {c[n]:.6g} |")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd_raw, wr = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
        rd = 2 * rd_raw
        lines += ["", f"HBM read  : FETCH_SIZE x 1024 = {rd_raw/1e9:.3f} GB raw, x2 gfx950 correction = **{rd/1e9:.3f} GB**",
                  f"HBM write : WRITE_SIZE x 1024 = **{wr/1e9:.3f} GB**",
                  f"HBM total (corrected) = **{(rd+wr)/1e9:.3f} GB** per launch"]
        traffic[k] = dict(read_raw=rd_raw, read_corrected=rd, write=wr, total=rd + wr)
    if "SQ_INSTS_VALU" in c and c.get("SQ_WAVES"):
        lines.append(f"VALU instructions per wave = {c['SQ_INSTS_VALU']/c['SQ_WAVES']:.0f}; "
                     f"SALU per wave = {c.get('SQ_INSTS_SALU', 0)/c['SQ_WAVES']:.0f}")
    if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
        lines.append(f"SQ_WAIT_ANY / SQ_WAVE_CYCLES = {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.1%}; "
                     f"SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = {c.get('SQ_ACTIVE_INST_VALU', 0)/c['SQ_WAVE_CYCLES']:.1%}; "
                     f"SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {c.get('SQ_WAIT_INST_ANY', 0)/c['SQ_WAVE_CYCLES']:.1%}")
    if "SQ_LDS_BANK_CONFLICT" in c:
        act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        lines.append(f"LDS bank-conflict cycles / LDS active cycles = {c['SQ_LDS_BANK_CONFLICT']:.6g} / {act:.6g}"
                     + (f" = {c['SQ_LDS_BANK_CONFLICT']/act:.3%}" if act else ""))
    if "TCC_HIT_sum" in c:
        tot = c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0)
        if tot:
            lines.append(f"L2 hit rate = {c['TCC_HIT_sum']/tot:.3%}")
    lines.append("")

open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

if key and traffic:
    # profiles/traffic.json has ONE writer, scripts/make_traffic.py, which derives every entry from the summary it names: register the
    # pair there (ENTRIES) and regenerate
    print(f"\nnext: add (\"{key}\", \"{tag}_summary.md\") to ENTRIES in scripts/make_traffic.py and run it")
