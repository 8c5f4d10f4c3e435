#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5f
./scripts/ubench/clock_rates 2>&1 | tee ${O}_clock_rates.txt
for rep in 1 2; do
for lib in libvit_hip_endprod.so libvit_hip.so; do
VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/$lib python scripts/time_update.py 6 SOFT16 65536 8192 4 2>&1 | grep -v amdgpu.ids
VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/$lib timeout -k 10 200 python scripts/time_pipeline.py 9 4 501,441,331,315 SOFT16 65536 8192 10 2>&1 | grep -v amdgpu.ids
done
done
timeout -k 10 200 python scripts/time_pipeline.py 9 2 491,369 SOFT16 65536 8192 10 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python -m pytest tests/test_gpu_resume.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -x -q -m gpu > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 ${O}_pytest.log
timeout -k 10 200 python -u tests/soak_fuzz.py 45 920000 > ${O}_soak.log 2>&1; echo "soak rc=$?"; tail -1 ${O}_soak.log
