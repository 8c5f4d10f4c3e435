#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5d
timeout -k 10 300 python -m pytest tests/test_gpu_residency.py tests/test_gpu_fuzz.py -x -q -m gpu > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 ${O}_pytest.log
T="timeout -k 10 200 python scripts/time_pipeline.py"
for rep in 1 2 3; do
for lib in libvit_hip_base.so libvit_hip.so; do
export VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/$lib
echo "== $lib"
$T 7 2 109,79 SOFT16 65536 8192 20
$T 7 3 91,117,121 SOFT16 65536 8192 20
$T 7 4 109,79,83,109 SOFT16 65536 8192 20
done
done > ${O}_time.log 2>&1
grep -v amdgpu.ids ${O}_time.log
unset VIT_HIP_LIB_PATH
python scripts/time_update.py 6 SOFT16 65536 8192 4 2>&1 | grep -v amdgpu.ids
python scripts/time_update.py 5 SOFT16 65536 8192 4 2>&1 | grep -v amdgpu.ids
bash scripts/pmc_codes.sh r5d 65536 2048 4 5 6
