set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_api.py tests/test_gpu_bench.py tests/test_gpu_resume.py -x -q -m gpu -k "slab or bench or pipeline or clock" > gpurun_out/r3_t1.log 2>&1; echo "pytest rc=$?" 
tail -5 gpurun_out/r3_t1.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err; echo rc=$?
python bench.py --steps 20 --warmup 5 --via python --no-cpu-baseline > gpurun_out/r3_bench_default_py.json 2>> gpurun_out/r3_bench_default.err; echo rc=$?
python bench.py --config 3 --steps 20 --warmup 5 > gpurun_out/r3_bench_hard8.json 2> gpurun_out/r3_bench_hard8.err; echo rc=$?
VIT_HIP_PIPELINE_UPDATES=1 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench_hard8_1upd.json 2>> gpurun_out/r3_bench_hard8.err; echo rc=$?
python bench.py --config 2 --steps 10 --warmup 3 > gpurun_out/r3_bench_k9.json 2> gpurun_out/r3_bench_k9.err; echo rc=$?
python - <<'PY'
import json
for f in ["r3_bench_default","r3_bench_default_py","r3_bench_hard8","r3_bench_hard8_1upd","r3_bench_k9"]:
    try:
        r=json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][0])
        print(f, round(r["value"]), r["ms_per_step"], r["ms_per_step_min"], r["ms_per_step_median"], r["ms_per_step_max"], r["update_ms"], r["chainback_ms"], r["clock_mhz"], r["config"]["pipeline"], r.get("parity",{}).get("bit_exact"))
    except Exception as e:
        print(f, "ERR", e)
PY
