#!/bin/bash
# round 4 experiment 4: K7 LDS-ring chainback (32 registers, 24 KiB ring) as the pipeline's kernel; three update waves per SIMD again
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_punctured.py tests/test_gpu_cpp.py -x -q -m gpu > gpurun_out/r4_exp4_tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -3 gpurun_out/r4_exp4_tests.log
[ $rc -eq 0 ] || exit $rc
line() { python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(r['value']), 'Mbit/s step', round(r['ms_per_step'],3), 'median', round(r['ms_per_step_median'],3), 'upd', round(r['update_ms'],3), 'cb', round(r['chainback_ms'],3), 'clk', round(r['clock_mhz']['under_load']))"; }
B="timeout -k 10 200 python bench.py --steps 24 --warmup 6 --no-cpu-baseline"
for rep in 1 2; do
VIT_HIP_PIPELINE_CB_SMALL=0 $B 2>/dev/null | line "k7 65536 reg-ring cb       " || exit 1
$B 2>/dev/null | line "k7 65536 lds-ring cb D=4   " || exit 1
VIT_HIP_LIB_PATH=$PWD/build_ab/libvit_hip_d8.so $B 2>/dev/null | line "k7 65536 lds-ring cb D=8   " || exit 1
VIT_HIP_PIPELINE_SPLIT=1 $B 2>/dev/null | line "k7 65536 as 2x32768, 2 upd " || exit 1
VIT_HIP_PIPELINE_SPLIT=1 VIT_HIP_PIPELINE_UPDATES=3 $B 2>/dev/null | line "k7 65536 as 2x32768, 3 upd " || exit 1
VIT_HIP_PIPELINE_SPLIT=1 VIT_HIP_PIPELINE_UPDATES=3 VIT_HIP_PIPELINE_CB_SMALL=0 $B 2>/dev/null | line "k7 65536 as 2x32768, 3 upd reg cb" || exit 1
VIT_HIP_PIPELINE_OVERLAP=1 $B --frames 98304 2>/dev/null | line "k7 98304 overlap lds-ring  " || exit 1
$B --frames 98304 2>/dev/null | line "k7 98304 back to back      " || exit 1
VIT_HIP_PIPELINE_CB_SMALL=0 $B --config 3 2>/dev/null | line "hard8 32768 reg-ring cb    " || exit 1
$B --config 3 2>/dev/null | line "hard8 32768 lds-ring cb    " || exit 1
VIT_HIP_PIPELINE_UPDATES=3 $B --config 3 2>/dev/null | line "hard8 32768 3 upd lds-ring " || exit 1
done
