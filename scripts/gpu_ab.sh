#!/bin/bash
# scripts/gpu_ab.sh <lib A> <lib B> [reps] -- same-box A/B of two builds of libvit_hip.so (VIT_HIP_LIB_PATH): update / chainback
# kernels alone (scripts/time_update.py) and the bench lines of BASELINE configs 1, 2, 3, alternating A B A B ...
A=$1; B=$2; REPS=${3:-2}
line() { python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(r['value']), 'Mbit/s step', round(r['ms_per_step'],3), 'median', round(r['ms_per_step_median'],3), 'upd', round(r['update_ms'],3), 'cb', round(r['chainback_ms'],3), 'clk', round(r['clock_mhz']['under_load']))"; }
BENCH="timeout -k 10 300 python bench.py --no-cpu-baseline --steps 24 --warmup 6"
for rep in $(seq $REPS); do
for lib in $A $B; do
export VIT_HIP_LIB_PATH=$PWD/$lib
python scripts/time_update.py 2 SOFT16 65536 8192 5 2>&1 | grep -v amdgpu.ids
python scripts/time_update.py 2 HARD8 32768 8192 5 2>&1 | grep -v amdgpu.ids
$BENCH --config 1 2>/dev/null | line "$lib k7   " || exit 1
$BENCH --config 3 2>/dev/null | line "$lib hard8" || exit 1
done
done
