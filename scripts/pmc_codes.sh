#!/bin/bash
# scripts/pmc_codes.sh <tag> <frames> <L> code... -- rocprofv3 counter passes over scripts/pmc_codes.py (update and chainback of each
# code ALONE); scripts/pmc_codes_summary.py turns gpurun_out/pmc_<tag>/ into a table
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() { name=$1; shift; timeout -k 10 400 rocprofv3 "$@" -d "$OUT/$name" --output-format csv -- python3 "$REPO/scripts/pmc_codes.py" $ARGS > "$OUT/$name.log" 2>&1; echo "$TAG $name rc=$?"; }
ARGS="$*"
run trace --kernel-trace --stats
run sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
cd "$REPO" && python3 scripts/pmc_codes_summary.py "$OUT" > "$OUT/summary.txt" 2>&1; cat "$OUT/summary.txt"
