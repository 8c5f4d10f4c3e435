mkdir -p gpurun_out
python -m pytest tests/test_gpu_api.py tests/test_gpu_bench.py -x -q -m gpu -k "pipeline or bench or presets or headline or single_rank" > gpurun_out/r3_pipe_tests.log 2>&1; echo rc=$?
tail -5 gpurun_out/r3_pipe_tests.log
for c in 2 3 1; do
python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_pipe_c$c.json 2> gpurun_out/r3_pipe_c$c.err; echo rc=$?
python - <<PY
import json
r=json.loads([l for l in open("gpurun_out/r3_pipe_c$c.json") if l.startswith("{")][0])
print("config $c", round(r["value"]), r["ms_per_step"], r["ms_per_step_median"], r["update_ms"], r["chainback_ms"], r["config"]["pipeline"], r["roofline"]["frac"], r.get("roofline_valu",{}).get("frac"))
PY
done
