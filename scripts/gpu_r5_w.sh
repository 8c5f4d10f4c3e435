#!/bin/bash
# round 5, call w: the whole GPU suite and a soak run on the final library
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 1000 python -m pytest tests -q -m gpu 2>&1 | tee gpurun_out/w_tests.log || exit 1
timeout -k 10 400 python -u tests/soak_fuzz.py 180 960000 2>&1 | tee gpurun_out/w_soak.log | tail -3
