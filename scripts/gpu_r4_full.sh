#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_full_tests.log 2>&1; rc=$?; echo tests rc=$rc; tail -5 gpurun_out/r4_full_tests.log
exit $rc
