#!/bin/bash
# round 5, call n: DAB producer without the unreachable patterns -- parity on the PLAN_REG codes, DAB / Voyager / LTE timing
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_resume.py tests/test_gpu_punctured.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/n_tests.log; rc=$?
cat gpurun_out/n_tests.log
[ $rc -eq 0 ] || exit $rc
T="timeout -k 10 300 python scripts/time_pipeline.py"
for rep in 1 2 3; do
for dt in SOFT16 SOFT8 HARD8; do
$T 7 4 109,79,83,109 $dt 65536 8192 12 2>&1 | grep -v amdgpu.ids
done
$T 7 2 109,79 SOFT16 65536 8192 12 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/n_dab_time.log
