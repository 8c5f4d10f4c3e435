# round-3 bench lines of the final code, one per BASELINE configuration, in bench.py's schema (driver form: --steps 20 --warmup 5)
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench.err; echo "default rc=$?"
python bench.py --config 3 --steps 20 --warmup 5 > gpurun_out/r3_bench_hard8.json 2>> gpurun_out/r3_bench.err; echo "hard8 rc=$?"
python bench.py --config 2 --steps 10 --warmup 3 > gpurun_out/r3_bench_k9.json 2>> gpurun_out/r3_bench.err; echo "k9 rc=$?"
python bench.py --config 4 --steps 5 --warmup 2 > gpurun_out/r3_bench_k15.json 2>> gpurun_out/r3_bench.err; echo "k15 rc=$?"
python bench.py --config 1 --steps 20 --warmup 5 --via python --no-cpu-baseline > gpurun_out/r3_bench_default_via_python.json 2>> gpurun_out/r3_bench.err; echo "python-route rc=$?"
python bench.py --gpus 2 --share-gpu --backend gloo --frames 32768 --steps 10 --warmup 3 2>> gpurun_out/r3_bench.err | grep '^{' > gpurun_out/r3_bench_2rank_share_gpu.json; echo "2rank rc=$?"
[ -n "${BENCH_LINES_PROFILE:-}" ] && PROFILE_STEPS=${PROFILE_STEPS:-10} PROFILE_WARMUP=${PROFILE_WARMUP:-4} bash scripts/profile.sh ${BENCH_LINES_PROFILE} > /dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3_bench_*.json")):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, round(r["value"]), round(r["ms_per_step"], 3), r["update_ms"], r["chainback_ms"], r["roofline"]["frac"], r.get("roofline_valu", {}).get("frac"),
              r.get("cpu_baseline", {}).get("value"), r.get("parity", {}).get("bit_exact"))
    except Exception as e:
        print(f, "ERR", e)
PY
