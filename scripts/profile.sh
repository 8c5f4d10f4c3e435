#!/bin/bash
# scripts/profile.sh <tag> [bench args...] -- rocprofv3 evidence for bench.py on the GPU box.
#   pass 1: --kernel-trace --stats        (per-kernel durations)
#   pass 2..: --pmc <counters>            (separate passes: TCC slots do not fit FETCH_SIZE and WRITE_SIZE together)
# Output under gpurun_out/prof_<tag>/ ; scripts/summarize_profile.py turns it into profiles/<tag>_*.{md,json}.
# The profiled program is `python3 bench.py` itself (no env/sh hop: the profiler's preload has the GPU initialised).
set -u
TAG=${1:-r2}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
STEPS=${PROFILE_STEPS:-20}
WARM=${PROFILE_WARMUP:-10}      # the card ramps its clocks over the first ~10 launches of a fresh process
ARGS="--steps $STEPS --warmup $WARM --no-cpu-baseline --no-extra-configs --sustain-seconds 0 $*"
run() { name=$1; shift; timeout -k 10 600 rocprofv3 "$@" -d "$OUT/$name" --output-format csv -- python3 "$REPO/bench.py" $ARGS > "$OUT/$name.log" 2>&1; echo "$TAG $name rc=$?"; }
run trace --kernel-trace --stats
run pmc_fetch --kernel-trace --pmc FETCH_SIZE
run pmc_write --kernel-trace --pmc WRITE_SIZE
run pmc_sq1 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run pmc_sq2 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run pmc_grbm --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT
run pmc_tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
du -sh "$OUT"
