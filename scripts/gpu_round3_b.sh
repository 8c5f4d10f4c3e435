mkdir -p gpurun_out
for i in 1 2; do
VIT_BENCH_SETTLE_MS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_b_nosettle_$i.json 2> gpurun_out/r3_b.err; echo rc=$?
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_b_settle_$i.json 2>> gpurun_out/r3_b.err; echo rc=$?
done
VIT_BENCH_SETTLE_MS=0 python bench.py --steps 20 --warmup 40 --no-cpu-baseline > gpurun_out/r3_b_w40.json 2>> gpurun_out/r3_b.err; echo rc=$?
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3_b_*.json")):
    r=json.loads([l for l in open(f) if l.startswith("{")][0])
    print(f, round(r["value"]), round(r["ms_per_step"],3), r["update_ms"], r["chainback_ms"], r["clock_mhz"], r["ms_per_step_series"])
PY
