#!/bin/bash
# round 4 experiment 1: opcode issue rates; K7 three-update-waves schedules with the LDS-ring chainback at its real 32 registers
mkdir -p gpurun_out
timeout -k 10 300 scripts/ubench/op_rates > gpurun_out/r4_op_rates.txt 2>&1; echo op_rates rc=$?
line() { python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$1', round(r['value']), 'Mbit/s step', round(r['ms_per_step'],3), 'median', round(r['ms_per_step_median'],3), 'upd', round(r['update_ms'],3), 'cb', round(r['chainback_ms'],3), 'clk', round(r['clock_mhz']['under_load']))"; }
B="timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2; do
$B 2>/dev/null | line "65536 default      " || exit 1
VIT_HIP_CHAINBACK_ALT=1 $B 2>/dev/null | line "65536 alt-cb       " || exit 1
$B --frames 98304 2>/dev/null | line "98304 back-to-back " || exit 1
VIT_HIP_PIPELINE_OVERLAP=1 $B --frames 98304 2>/dev/null | line "98304 overlap reg  " || exit 1
VIT_HIP_PIPELINE_OVERLAP=1 VIT_HIP_CHAINBACK_ALT=1 $B --frames 98304 2>/dev/null | line "98304 overlap alt  " || exit 1
done
VIT_HIP_CHAINBACK_ALT=1 timeout -k 10 120 python scripts/exp_pipeline3.py 32768 36 3 || exit 1
timeout -k 10 120 python scripts/exp_pipeline3.py 32768 36 3 || exit 1
timeout -k 10 120 python scripts/exp_pipeline3.py 32768 36 2 || exit 1
