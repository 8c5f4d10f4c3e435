#!/usr/bin/env python3
"""bench.py with the CPU baseline, digest of the interesting fields."""
import json, subprocess, sys
p = subprocess.run([sys.executable, "bench.py"] + sys.argv[1:], capture_output=True, text=True)
try:
    d = json.loads(p.stdout.strip().splitlines()[-1])
    cb = d.get("cpu_baseline", {})
    print(" ".join(sys.argv[1:]), {"value": round(d["value"], 1), "upd": round(d["update_ms"], 2), "cb": round(d["chainback_ms"], 2),
          "plan": d["config"]["plan"], "cpu": round(cb.get("value", 0), 1), "cpu_kind": cb.get("strategy"), "cores": cb.get("cores"),
          "scalar1": round(cb.get("scalar_1thread_Mbit_s", 0), 2), "speedup": round(d.get("speedup_vs_cpu_baseline", 0), 1),
          "parity": d.get("parity"), "ber": d["ber"]})
except Exception as e:
    print("FAILED", e, p.stdout[-500:], p.stderr[-1500:])
