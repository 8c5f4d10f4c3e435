#!/bin/bash
# round 4: K9 chainback with dynamic LDS (24 allocated registers) -- parity tests, then same-box A/B against the round-3 library
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_api.py -x -q -m gpu -k "CDMA or K9 or k9 or 9- or chainback_bodies or custom or runtime" > gpurun_out/r4_k9_tests.log 2>&1; rc=$?; echo tests rc=$rc
tail -3 gpurun_out/r4_k9_tests.log
[ $rc -eq 0 ] || exit $rc
for lib in build_ab/libvit_hip_prev.so viterbidecodercpp_amd/libvit_hip.so build_ab/libvit_hip_prev.so viterbidecodercpp_amd/libvit_hip.so; do
VIT_HIP_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$lib bench', round(r['value']), r['ms_per_step'], r['ms_per_step_median'], r['update_ms'], r['chainback_ms'], r.get('clock_mhz'))" || exit 1
done
