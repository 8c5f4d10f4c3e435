#!/bin/bash
# round 5, call v: K = 10, 11 with address-free metric loads -- parity (LDS2 tests) and rates
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_resume.py tests/test_gpu_api.py -x -q -m gpu -k "not runtime_instantiation and not rate_5" 2>&1 | tee gpurun_out/v_tests.log | tail -4 || exit 1
T="timeout -k 10 300 python scripts/time_pipeline.py"
{
$T 10 2 1005,755 SOFT16 32768 8192 6
$T 11 2 1845,1995 SOFT16 32768 8192 6
$T 10 2 1005,755 HARD8 32768 8192 6
$T 11 2 1845,1995 SOFT8 32768 8192 6
$T 12 2 2787,3645 SOFT16 4096 8192 6
$T 12 2 2787,3645 SOFT16 16384 8192 6
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/v_k10_k11.log
