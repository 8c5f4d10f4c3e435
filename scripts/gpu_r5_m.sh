#!/bin/bash
# round 5, call m: K15 fast block with spelled-out LDS accesses; the scalar-unit table build (libvit_hip_sb.so) against it
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_resume.py tests/test_gpu_api.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/m_tests.log; rc=$?
cat gpurun_out/m_tests.log
[ $rc -eq 0 ] || exit $rc
VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/libvit_hip_sb.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_resume.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/m_tests_sb.log; rc=$?
cat gpurun_out/m_tests_sb.log
[ $rc -eq 0 ] || exit $rc
bash scripts/gpu_ab_l2.sh viterbidecodercpp_amd/libvit_hip.so viterbidecodercpp_amd/libvit_hip_sb.so 3 > gpurun_out/ab_sb.log 2>&1
grep -v "SOFT8" gpurun_out/ab_sb.log
