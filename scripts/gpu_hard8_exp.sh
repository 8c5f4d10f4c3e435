mkdir -p gpurun_out
for f in 32768 65536 32768 65536; do
python bench.py --code 2 --decode-type SOFT16 --frames $f --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/r3_k7_f$f.json 2>/dev/null
python - <<PY
import json
r=json.loads([l for l in open("gpurun_out/r3_k7_f$f.json") if l.startswith("{")][0])
print("K7 soft16 frames=$f", round(r["value"]), r["ms_per_step_median"], r["update_ms"], r["chainback_ms"], r["config"]["pipeline"])
PY
done
for f in 16384 32768; do
python bench.py --code 5 --decode-type SOFT16 --frames $f --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/r3_k9_f$f.json 2>/dev/null
python - <<PY
import json
r=json.loads([l for l in open("gpurun_out/r3_k9_f$f.json") if l.startswith("{")][0])
print("K9 soft16 frames=$f", round(r["value"]), r["ms_per_step_median"], r["update_ms"], r["chainback_ms"], r["config"]["pipeline"])
PY
done
