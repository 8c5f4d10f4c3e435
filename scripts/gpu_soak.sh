#!/bin/bash
# scripts/gpu_soak.sh [seconds per leg] [first seed] -- the randomised soak (tests/soak_fuzz.py) on the GPU box, three legs: the
# kernels as shipped, then each of the two chainback kernels of K7 / K9 forced in turn (vit_hip_chainback_batch_ex).  Logs go straight
# to files under gpurun_out/ (a pipe would hold the progress lines back and the run would look hung).
mkdir -p gpurun_out
SECS=${1:-300}; SEED=${2:-700000}
python -u tests/soak_fuzz.py $SECS $SEED > gpurun_out/soak_shipped.log 2>&1 && tail -1 gpurun_out/soak_shipped.log &&
python -u tests/soak_fuzz.py $SECS $((SEED + 100000)) 1 > gpurun_out/soak_alt0.log 2>&1 && tail -1 gpurun_out/soak_alt0.log &&
python -u tests/soak_fuzz.py $SECS $((SEED + 200000)) 2 > gpurun_out/soak_alt1.log 2>&1 && tail -1 gpurun_out/soak_alt1.log
