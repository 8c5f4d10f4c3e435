#!/bin/bash
# round 5, call z: rates of the small-K run-time instantiations at R = 5, 6 and K = 6 at an odd rate
mkdir -p gpurun_out
T="timeout -k 10 400 python scripts/time_pipeline.py"
{
$T 5 5 23,25,27,31,29 SOFT16 65536 8192 8
$T 6 6 53,47,61,43,57,39 SOFT16 65536 8192 8
$T 6 3 53,47,61 SOFT16 65536 8192 8
$T 6 2 53,47 SOFT16 65536 8192 8
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/z_smallk.log
