#!/bin/bash
# scripts/gpu_evidence.sh <tag> [lines|profiles|all] -- the measured evidence of a round, on the GPU box:
#   lines    : bench.py's JSON line as the driver runs it (the default line carries BASELINE configs[2..4] as `configs`), configs[0],
#              the Python-route, the 98304-frame and the two-ranks-one-card lines  -> gpurun_out/<tag>_bench_*.json  (copy to profiles/)
#   profiles : rocprofv3 passes of the four BASELINE configs (scripts/profile.sh: one --kernel-trace --stats pass + six --pmc passes
#              each) -> gpurun_out/prof_<tag>_*/   (scripts/summarize_profile.py -> profiles/<tag>_*_summary.md; scripts/make_traffic.py)
TAG=${1:-r6}; WHAT=${2:-all}
mkdir -p gpurun_out
if [ "$WHAT" = lines ] || [ "$WHAT" = all ]; then
  E=gpurun_out/${TAG}_bench.err; : > $E
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_default.json 2>> $E; echo "default rc=$?"
  python bench.py --config 0 > gpurun_out/${TAG}_bench_config0.json 2>> $E; echo "config0 rc=$?"
  python bench.py --frames 98304 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_k7_98304.json 2>> $E; echo "98304 rc=$?"
  python bench.py --config 1 --steps 20 --warmup 5 --via python --no-cpu-baseline > gpurun_out/${TAG}_bench_default_via_python.json 2>> $E; echo "python-route rc=$?"
  python bench.py --gpus 2 --share-gpu --backend gloo --frames 32768 --steps 10 --warmup 3 2>> $E | grep '^{' > gpurun_out/${TAG}_bench_2rank_share_gpu.json; echo "2rank rc=$?"
  python - "$TAG" <<'PY'
import json, glob, sys
def one(name, r):
    print(name, round(r["value"]), "steady", round(r.get("value_steady") or 0), "sustained", round(r.get("value_sustained") or 0), round(r["ms_per_step"], 3), round(r.get("update_ms", 0), 3),
          round(r.get("chainback_ms", 0), 3), round(r["roofline"]["frac"], 4), r.get("roofline_valu", {}).get("frac"), r.get("cpu_baseline", {}).get("value"), r.get("parity", {}).get("bit_exact"))
for f in sorted(glob.glob(f"gpurun_out/{sys.argv[1]}_bench_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][0])
        one(f, r)
        for c in r.get("configs", []):
            one(f"   configs[{c['baseline_config']}]", c)
    except Exception as e:
        print(f, "ERR", e)
PY
fi
if [ "$WHAT" = profiles ] || [ "$WHAT" = all ]; then
  PROFILE_STEPS=20 PROFILE_WARMUP=10 bash scripts/profile.sh ${TAG}_k7_default --config 1
  PROFILE_STEPS=20 PROFILE_WARMUP=10 bash scripts/profile.sh ${TAG}_hard8 --config 3
  PROFILE_STEPS=10 PROFILE_WARMUP=4 bash scripts/profile.sh ${TAG}_k9 --config 2
  PROFILE_STEPS=5 PROFILE_WARMUP=2 bash scripts/profile.sh ${TAG}_k15 --config 4
fi
