#!/bin/bash
# scripts/profile_config0.sh -- rocprofv3 record of BASELINE configs[0] (one 4096-bit frame through the drop-in): per-kernel durations, then one
# counter pass (separate runs).  Output: gpurun_out/prof_config0/ ; the stats CSV and the counter rows are copied to profiles/ by hand.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_config0
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
ARGS="--config 0 --steps 200 --warmup 20 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" --output-format csv -- python3 "$REPO/bench.py" $ARGS > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$OUT/pmc_sq" --output-format csv -- python3 "$REPO/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1; echo "pmc rc=$?"
find "$OUT" -name "*kernel_stats.csv" | head -2
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cat "$f" | head -8
