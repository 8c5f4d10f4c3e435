#!/bin/bash
# round 5, GPU call C: parity after the lane-basis ring addressing + A/B against the previous build (libvit_hip_base.so)
mkdir -p gpurun_out
O=gpurun_out/r5c
timeout -k 10 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_residency.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_resume.py tests/test_gpu_fuzz.py tests/test_gpu_punctured.py -x -q -m gpu > ${O}_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 ${O}_pytest.log
timeout -k 10 200 python -u tests/soak_fuzz.py 60 910000 > ${O}_soak.log 2>&1; echo "soak rc=$?"; tail -1 ${O}_soak.log
T="timeout -k 10 200 python scripts/time_pipeline.py"
for rep in 1 2; do
for lib in libvit_hip_base.so libvit_hip.so; do
export VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/$lib
echo "== $lib"
$T 7 2 109,79 SOFT16 65536 8192 20
$T 7 2 109,79 HARD8 32768 8192 20
$T 7 3 91,117,121 SOFT16 65536 8192 20
$T 7 4 109,79,83,109 SOFT16 65536 8192 20
$T 9 2 491,369 SOFT16 65536 8192 10
$T 9 4 501,441,331,315 SOFT16 65536 8192 10
done
done > ${O}_time.log 2>&1
grep -v amdgpu.ids ${O}_time.log
unset VIT_HIP_LIB_PATH
$T 9 4 501,441,331,315 SOFT16 65536 8192 10 sub_batches=1 2>&1 | grep -v amdgpu.ids
$T 7 3 91,117,121 SOFT16 98304 8192 12 2>&1 | grep -v amdgpu.ids
$T 7 3 91,117,121 SOFT16 98304 8192 12 chainback_overlap=1 2>&1 | grep -v amdgpu.ids
