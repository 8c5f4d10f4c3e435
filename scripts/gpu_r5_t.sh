#!/bin/bash
# round 5, call t: rates of the run-time compiled K = 8 instantiations beside their neighbours (K7 R4 DAB, K9 R4 CDMA 2000)
mkdir -p gpurun_out
T="timeout -k 10 400 python scripts/time_pipeline.py"
{
$T 8 2 249,167 SOFT16 65536 8192 8
$T 8 3 247,217,149 SOFT16 65536 8192 8
$T 8 4 249,167,247,217 SOFT16 65536 8192 8
$T 8 6 249,167,247,217,149,203 SOFT16 65536 8192 8
$T 7 4 109,79,83,109 SOFT16 65536 8192 8
$T 9 4 501,441,331,315 SOFT16 65536 8192 8
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t_k8.log
