#!/bin/bash
# round 5, call x: table passes at K = 13 (two groups per thread) -- parity and rate
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_resume.py -x -q -m gpu 2>&1 | tee gpurun_out/x_tests.log | tail -4 || exit 1
T="timeout -k 10 300 python scripts/time_pipeline.py"
{
$T 13 2 4443,8113 SOFT16 8192 4096 6
$T 13 3 4443,4541,8113 SOFT16 8192 4096 6
$T 12 2 2787,3645 SOFT16 16384 8192 6
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x_k13.log
