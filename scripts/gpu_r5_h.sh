#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5h
export VIT_HIP_CACHE_DIR=$PWD/gpurun_out/jit_cache2; mkdir -p $VIT_HIP_CACHE_DIR; chmod 700 $VIT_HIP_CACHE_DIR
timeout -k 10 1100 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_resume.py tests/test_gpu_punctured.py tests/test_gpu_cpp.py tests/test_gpu_residency.py -x -q -m gpu > ${O}_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 ${O}_pytest.log
T="timeout -k 10 300 python scripts/time_pipeline.py"
{
$T 7 5 0o171,0o133,0o165,0o117,0o135 SOFT16 65536 8192 12
$T 9 5 0o557,0o663,0o711,0o561,0o753 SOFT16 65536 8192 8
$T 10 2 0o1167,0o1545 SOFT16 32768 8192 6 
} > ${O}_time.log 2>&1
grep -v amdgpu.ids ${O}_time.log
timeout -k 10 200 python -u tests/soak_fuzz.py 60 930000 > ${O}_soak.log 2>&1; echo "soak rc=$?"; tail -1 ${O}_soak.log
