#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5j
timeout -k 10 600 python -m pytest tests/test_gpu_latency.py -x -q -m gpu > ${O}_pytest.log 2>&1; echo "latency pytest rc=$?"; tail -3 ${O}_pytest.log
B="timeout -k 10 300 python bench.py --no-cpu-baseline --parity --sustain-seconds 0 --steps 24 --warmup 6"
line() { python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(r['value']), 'steady', round(r['value_steady']), 'upd', round(r['update_ms'],3), 'cb', round(r['chainback_ms'],3), 'parity', r['parity']['bit_exact'])"; }
for rep in 1 2; do
for lib in libvit_hip.so libvit_hip_stag4.so; do
VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/$lib $B --config 1 2>/dev/null | line "$lib k7"
VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/$lib $B --config 2 --steps 10 --warmup 3 2>/dev/null | line "$lib k9"
done
done
VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/libvit_hip_stag4.so timeout -k 10 300 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "chainback_bodies" > ${O}_stag_pytest.log 2>&1; echo "stag4 chainback tests rc=$?"; tail -2 ${O}_stag_pytest.log
timeout -k 10 900 python -m viterbidecodercpp_amd.tools.run_snr_ber --codes 2 5 7 --decode-types SOFT16 HARD8 --bits-scale 4 > gpurun_out/r5_snr_ber_hip.json 2> ${O}_snr.log; echo "snr rc=$?"; tail -2 ${O}_snr.log
timeout -k 10 600 python -m viterbidecodercpp_amd.tools.run_benchmark -T 0.3 > gpurun_out/r5_run_benchmark_hip.json 2> ${O}_rb.log; echo "run_benchmark rc=$?"; tail -2 ${O}_rb.log
