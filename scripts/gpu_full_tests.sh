#!/bin/bash
# the driver's round-end GPU tier, with the log kept under gpurun_out/
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/gpu_full.log 2>&1; rc=$?
tail -30 gpurun_out/gpu_full.log
exit $rc
