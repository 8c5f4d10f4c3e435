#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5e
for rep in 1 2; do
for lib in libvit_hip.so libvit_hip_a2.so; do
VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/$lib python scripts/time_update.py 6 SOFT16 65536 8192 4 2>&1 | grep -v amdgpu.ids
VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/$lib timeout -k 10 200 python scripts/time_pipeline.py 9 4 501,441,331,315 SOFT16 65536 8192 10 2>&1 | grep -v amdgpu.ids
done
done
bash scripts/pmc_codes.sh r5e 16384 1024 0 1 2 3 4 5 6 7 0:SOFT8 1:SOFT8 2:SOFT8 3:SOFT8 4:SOFT8 5:SOFT8 6:SOFT8 7:SOFT8 > ${O}_pmc.log 2>&1; tail -3 ${O}_pmc.log
timeout -k 10 900 python scripts/matrix.py gpurun_out/r5_matrix.json gpurun_out/pmc_r5e/valu.json 2>&1 | grep -v amdgpu.ids | tee ${O}_matrix.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_default.json 2> ${O}_bench.err; echo "bench rc=$?"
VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/libvit_hip_stamps.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_bench_default_stamps.json 2>> ${O}_bench.err; echo "stamps rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r5_bench_default.json", "gpurun_out/r5_bench_default_stamps.json"):
    try:
        r = json.loads([l for l in open(f) if l.startswith("{")][0])
        print(f, round(r["value"]), "steady", round(r["value_steady"]), "sustained", r.get("value_sustained"), {k: v for k, v in (r.get("sustained") or {}).items() if "series" not in k})
        print("  clock", {k: v for k, v in r["clock_mhz"].items() if k != "under_load_probe"})
        print("  parity", r.get("parity"), "cpu", (r.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
