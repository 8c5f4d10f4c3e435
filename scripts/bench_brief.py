#!/usr/bin/env python3
"""run bench.py with the given args and print a one-line digest (value, update_ms, chainback_ms)."""
import json, subprocess, sys
p = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline"] + sys.argv[1:], capture_output=True, text=True)
try:
    d = json.loads(p.stdout.strip().splitlines()[-1])
    print(" ".join(sys.argv[1:]), {k: round(d[k], 3) for k in ("value", "update_ms", "chainback_ms")}, d["config"]["plan"])
except Exception as e:
    print("FAILED", e, p.stdout[-500:], p.stderr[-1500:])
