#!/bin/bash
# round 5, call q: C++ drop-in programs (copyable decoder), whole GPU suite
mkdir -p gpurun_out
set -o pipefail
./tests/cpp/run_simple_hip | tee gpurun_out/q_simple.log || exit 1
timeout -k 10 1000 python -m pytest tests -q -m gpu 2>&1 | tee gpurun_out/q_tests.log
