#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5b
timeout -k 10 300 python -m pytest tests/test_gpu_residency.py -x -q -m gpu > ${O}_pytest.log 2>&1; echo "residency rc=$?"; tail -3 ${O}_pytest.log
for rep in 1 2; do
python scripts/time_update.py 6 SOFT16 65536 8192 4 2>&1 | grep -v amdgpu.ids
VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/libvit_hip_noxor.so python scripts/time_update.py 6 SOFT16 65536 8192 4 2>&1 | grep -v amdgpu.ids
python scripts/time_update.py 5 SOFT16 65536 8192 4 2>&1 | grep -v amdgpu.ids
done
bash scripts/pmc_codes.sh r5b 65536 2048 2 3 4 5 6
