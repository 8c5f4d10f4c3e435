#!/bin/bash
# round 5, call o: evidence on the final kernels, part 1 -- per-code VALU counts (PMC), the 8 x 3 matrix, the bench lines
mkdir -p gpurun_out
O=gpurun_out/r5o
bash scripts/pmc_codes.sh r5o 16384 1024 0 1 2 3 4 5 6 7 0:SOFT8 1:SOFT8 2:SOFT8 3:SOFT8 4:SOFT8 5:SOFT8 6:SOFT8 7:SOFT8 > ${O}_pmc.log 2>&1; tail -3 ${O}_pmc.log
timeout -k 10 900 python scripts/matrix.py gpurun_out/r5_matrix.json gpurun_out/pmc_r5o/valu.json 2>&1 | grep -v amdgpu.ids | tee ${O}_matrix.log
bash scripts/gpu_evidence.sh r5 lines
