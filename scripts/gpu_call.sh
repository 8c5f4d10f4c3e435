#!/bin/bash
# scripts/gpu_call.sh <tag> <step> [<step> ...] -- ONE gpurun call as a list of steps (replaces the one-off scripts/gpu_r5_[a-z].sh
# of round 5).  Every step runs under its own `timeout -k 10`, logs to gpurun_out/<tag>_<n>_<kind>.log, prints a one-line verdict,
# and a failed or killed step ENDS the call (no GPU step is ever started behind a kill).
#
#   pytest[:SECONDS]=<pytest arguments>      python -m pytest <arguments> -q -m gpu            (default 900 s)
#   time[:SECONDS]=<time_pipeline.py args>   scripts/time_pipeline.py K R G,G.. TYPE F L STEPS [option=value ..]   (default 300 s)
#   bench[:SECONDS]=<bench.py arguments>     python bench.py <arguments>; the JSON line goes to gpurun_out/<tag>_<n>_bench.json
#   profile=<ptag> <bench.py arguments>      scripts/profile.sh (rocprofv3 --kernel-trace --stats + separate --pmc passes)
#   matrix[:SECONDS]=<matrix.py arguments>   scripts/matrix.py (8 codes x 3 decode types)
#   sh[:SECONDS]=<command>                   anything else
#
# e.g.  gpurun --timeout 1100 -- 'bash scripts/gpu_call.sh r6a "pytest=tests/test_gpu_api.py -x" "bench=--steps 20 --warmup 5" \
#                                   "time=7 2 121,91 SOFT16 65536 8192 20"'
TAG=${1:?tag}; shift
mkdir -p gpurun_out
n=0
for step in "$@"; do
    n=$((n + 1))
    head=${step%%=*}; body=${step#*=}
    kind=${head%%:*}; secs=${head#*:}; [ "$secs" = "$head" ] && secs=""
    log=gpurun_out/${TAG}_${n}_${kind}.log
    case $kind in
        pytest)  timeout -k 10 ${secs:-900} python -m pytest $body -q -m gpu > "$log" 2>&1; rc=$?; tail -4 "$log" ;;
        time)    timeout -k 10 ${secs:-300} python scripts/time_pipeline.py $body > "$log" 2>&1; rc=$?; grep -v amdgpu.ids "$log" | tail -6 ;;
        bench)   timeout -k 10 ${secs:-600} python bench.py $body > "$log" 2>&1; rc=$?
                 grep '^{' "$log" > gpurun_out/${TAG}_${n}_bench.json
                 python - "$log" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        def one(r, name):
            rv = r.get("roofline_valu", {})
            print(f"  {name}: value {r['value'] / 1e3:.2f} steady {r.get('value_steady', 0) / 1e3:.2f} sustained {(r.get('value_sustained') or 0) / 1e3:.2f} Gbit/s, "
                  f"step {r['ms_per_step']:.3f} ms, update {r.get('update_ms', 0):.3f} chainback {r.get('chainback_ms', 0):.3f} ms, hbm frac {r['roofline']['frac']:.3f}, "
                  f"valu frac {rv.get('frac', float('nan')):.3f}, parity {r.get('parity', {}).get('bit_exact')}")
        one(r, r["config"]["workload"][:40])
        for c in r.get("configs", []):
            if "error" in c:
                print("  config", c["baseline_config"], "ERROR", c["error"])
            else:
                one(c, c["config"]["workload"][:40] + f" [{c['seconds_spent']:.0f} s]")
PY
                 ;;
        profile) bash scripts/profile.sh $body > "$log" 2>&1; rc=$?; tail -8 "$log" ;;
        matrix)  timeout -k 10 ${secs:-1000} python scripts/matrix.py $body > "$log" 2>&1; rc=$?; tail -30 "$log" ;;
        sh)      timeout -k 10 ${secs:-600} bash -c "$body" > "$log" 2>&1; rc=$?; tail -15 "$log" ;;
        *)       echo "gpu_call.sh: unknown step kind '$kind'"; exit 2 ;;
    esac
    echo "[$TAG step $n $kind] rc=$rc"
    [ $rc -ne 0 ] && exit $rc
done
exit 0
