#!/bin/bash
# round 4 experiment 2: opcode rates (operand kinds, realistic mixes); bitop3 gather A/B against the round-3 library
mkdir -p gpurun_out
timeout -k 10 400 scripts/ubench/op_rates > gpurun_out/r4_op_rates.txt 2>&1; echo op_rates rc=$?
tail -18 gpurun_out/r4_op_rates.txt
line() { python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(r['value']), 'Mbit/s step', round(r['ms_per_step'],3), 'median', round(r['ms_per_step_median'],3), 'upd', round(r['update_ms'],3), 'cb', round(r['chainback_ms'],3), 'clk', round(r['clock_mhz']['under_load']))"; }
B="timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2; do
for lib in build_ab/libvit_hip_prev.so viterbidecodercpp_amd/libvit_hip.so; do
export VIT_HIP_LIB_PATH=$PWD/$lib
python scripts/time_update.py 2 SOFT16 65536 8192 5 2>&1 | grep -v amdgpu.ids
python scripts/time_update.py 5 SOFT16 65536 8192 3 2>&1 | grep -v amdgpu.ids
$B 2>/dev/null | line "$lib k7 default" || exit 1
VIT_HIP_CHAINBACK_ALT=1 $B 2>/dev/null | line "$lib k7 alt-cb " || exit 1
$B --config 3 2>/dev/null | line "$lib hard8      " || exit 1
VIT_HIP_CHAINBACK_ALT=1 $B --config 3 2>/dev/null | line "$lib hard8 alt  " || exit 1
$B --config 2 2>/dev/null | line "$lib k9         " || exit 1
done
done
