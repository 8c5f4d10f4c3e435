#!/bin/bash
# round 4 experiment 5: tests on the new schedule rules + residency probe; K15 / K9 gather masks in VGPRs, same-box A/B against the last commit
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_residency.py tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_punctured.py tests/test_gpu_cpp.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r4_exp5_tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -5 gpurun_out/r4_exp5_tests.log
line() { python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(r['value']), 'Mbit/s step', round(r['ms_per_step'],3), 'median', round(r['ms_per_step_median'],3), 'upd', round(r['update_ms'],3), 'cb', round(r['chainback_ms'],3), 'clk', round(r['clock_mhz']['under_load']))"; }
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
for rep in 1 2; do
for lib in build_ab/libvit_hip_prev.so viterbidecodercpp_amd/libvit_hip.so; do
export VIT_HIP_LIB_PATH=$PWD/$lib
python scripts/time_update.py 7 SOFT16 4096 8192 3 2>&1 | grep -v amdgpu.ids
$B --config 4 --steps 6 --warmup 2 2>/dev/null | line "$lib k15" || exit 1
$B --config 1 --steps 24 --warmup 6 2>/dev/null | line "$lib k7 " || exit 1
$B --frames 98304 --steps 24 --warmup 6 2>/dev/null | line "$lib k7 98304" || exit 1
$B --config 3 --steps 24 --warmup 6 2>/dev/null | line "$lib hard8" || exit 1
$B --code 4 --steps 10 --warmup 3 2>/dev/null | line "$lib DAB k7r4" || exit 1
done
done
exit $rc
