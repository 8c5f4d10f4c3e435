#!/usr/bin/env python3
"""The reference's whole benchmark matrix -- 8 stock codes x {SOFT16, SOFT8, HARD8} (examples/run_benchmark.cpp:168-179,
examples/helpers/common_codes.h:20-30) -- end to end through vit_hip_pipeline_* at full batch size: one `bench.py` child per cell
(the same timed loop, roofline and in-run parity check as the headline line), collected into one JSON file.
usage: matrix.py out.json [valu.json from scripts/pmc_codes.sh] [--frames N] [--steps K] [--only code:type,...]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viterbidecodercpp_amd import COMMON_CODES

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
out_path = args[0]
valu = json.load(open(args[1])) if len(args) > 1 and os.path.exists(args[1]) else None
rates = json.load(open(os.path.join(ROOT, "profiles", "issue_rates.json")))
only = set(opt.get("--only", "").split(",")) - {""}
records = []
for ci, code in enumerate(COMMON_CODES):
    for dt in ("SOFT16", "SOFT8", "HARD8"):
        if only and f"{ci}:{dt}" not in only: continue
        F = int(opt.get("--frames", 65536)) if code.K < 11 else 4096
        steps = int(opt.get("--steps", 12)) if code.K < 9 else 8 if code.K < 11 else 4
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--code", str(ci), "--decode-type", dt, "--frames", str(F), "--bits", "8192",
               "--steps", str(steps), "--warmup", str(max(2, steps // 3)), "--no-cpu-baseline", "--parity", "--sustain-seconds", "0",
               "--ebn0", "5.0" if dt == "HARD8" else "3.0"]
        t0 = time.time()
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode != 0 or not line:
            records.append({"code": code.name, "decode_type": dt, "error": (p.stderr or p.stdout)[-800:]})
            print(f"{code.name:16s} {dt:6s} FAILED rc={p.returncode}", flush=True)
            continue
        r = json.loads(line[0])
        rec = {"code": code.name, "K": code.K, "R": code.R, "decode_type": dt, "frames": F, "bits": 8192,
               "value": r["value"], "value_steady": r["value_steady"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
               "ms_per_step_median": r["ms_per_step_median"], "update_ms": r["update_ms"], "chainback_ms": r["chainback_ms"],
               "schedule": r["config"]["pipeline"], "plan": r["config"]["plan"], "roofline": {k: r["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "frames_per_launch")},
               "roofline_end_to_end": r["roofline_end_to_end"], "state_updates_per_s": r["state_updates_per_s"], "parity": r.get("parity"), "ber": r["ber"],
               "clock_mhz_under_load": r["clock_mhz"]["under_load"], "bench_seconds": round(time.time() - t0, 1)}
        if valu:
            key = f"K{code.K}|R{code.R if code.K < 11 else (6 if code.R == 6 and code.K == 15 else 0)}|shift{0 if dt == 'SOFT16' else 8}|{rec['plan']}"
            v = valu["update_kernels"].get(key)
            fl = valu["frames_L_by_code"].get(f"{code.name}|{dt}") or valu["frames_L_by_code"].get(f"{code.name}|SOFT16")
            if v and fl and v["valu_insts"]:
                S_pmc, S = fl[1] + code.K - 1, 8192 + code.K - 1
                F_launch = r["roofline"]["frames_per_launch"]
                insts = v["valu_insts"] * (F_launch / fl[0]) * (S / S_pmc)          # instructions scale with waves x steps
                nupd = r["update_launches_in_flight"]
                tile = 32 if code.K in (7, 9) else 128
                waves = (F_launch / tile / 1024.0 * nupd) if rec["plan"] == "reg" else 4
                wkey = str(int(min(4, max(1, -(-waves // 1)))))
                peak = 1024 / rates["packed16_ns_per_wave_instr_per_simd"][wkey]
                ach = insts / (r["update_ms"] * 1e-3) / 1e9 * nupd
                rec["roofline_valu"] = {"bound": "valu", "achieved": ach, "peak": peak, "unit": "G wave-instr/s", "frac": ach / peak,
                                        "peak_spec": 1024 * 2.4 / 2, "frac_of_spec": ach / (1024 * 2.4 / 2),
                                        "valu_insts_per_launch": insts, "valu_per_wave_step": v["valu_insts"] / v["waves"] / S_pmc,
                                        "valu_per_state_update_pair": insts * 64.0 / (F_launch * S * (1 << (code.K - 1)) / 2.0),
                                        "update_waves_per_simd": waves, "insts_source": f"rocprofv3 --pmc SQ_INSTS_VALU at {fl[0]} x {fl[1]}, scaled by waves x steps"}
        records.append(rec)
        print(f"{code.name:16s} {dt:6s} {F:6d}x8192 {r['value'] / 1e3:8.2f} Gbit/s steady {r['value_steady'] / 1e3:8.2f} step {r['ms_per_step_median']:7.3f} ms upd {r['update_ms']:7.3f} cb {r['chainback_ms']:6.3f} "
              f"hbm {rec['roofline']['frac']:.3f} valu {rec.get('roofline_valu', {}).get('frac', float('nan')):.3f} parity {r.get('parity', {}).get('bit_exact')} | {rec['schedule'][18:]}", flush=True)
        json.dump({"records": records}, open(out_path, "w"), indent=1)
json.dump({"what": "8 stock codes x 3 decode types through vit_hip_pipeline_* (bench.py per cell), one MI355X", "records": records}, open(out_path, "w"), indent=1)
