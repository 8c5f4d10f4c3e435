#!/bin/bash
# round 4 experiment 6: v_cndmask cost; K15 table build with mask selects (A/B against the last commit's kernels)
mkdir -p gpurun_out
timeout -k 10 400 scripts/ubench/op_rates > gpurun_out/r4_op_rates.txt 2>&1; echo op_rates rc=$?
grep -E "cnd|bfe_i32|pk_ashr|^waves|^opcode" gpurun_out/r4_op_rates.txt
for rep in 1 2 3; do
for lib in build_ab/libvit_hip_prev.so viterbidecodercpp_amd/libvit_hip.so; do
VIT_HIP_LIB_PATH=$PWD/$lib python scripts/time_update.py 7 SOFT16 4096 8192 3 2>&1 | grep -v amdgpu.ids
done
done
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_parity.py -x -q -m gpu -k "pipeline or K15 or Cassini or 15 or lds2 or custom" > gpurun_out/r4_exp6_tests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r4_exp6_tests.log
