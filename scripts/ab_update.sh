#!/bin/bash
# scripts/ab_update.sh <frames> <L> <reps> <lib>[,<lib>..] <code:type> ... -- same-box A/B of the update / chainback kernels alone
# (scripts/time_update.py, HIP events, median), the libraries alternating per configuration.  Libraries relative to the repo root.
F=$1; L=$2; REPS=$3; LIBS=${4//,/ }; shift 4
for ct in "$@"; do
    for lib in $LIBS; do
        VIT_HIP_LIB_PATH=$PWD/viterbidecodercpp_amd/$lib python scripts/time_update.py ${ct%%:*} ${ct##*:} $F $L $REPS 2>&1 | grep -v amdgpu.ids | sed 's/.*viterbidecodercpp_amd.//'
    done
done
