#!/bin/bash
# round 5, call p: evidence on the final kernels, part 2 -- K15 profile, all 4096 K15 frames against the oracle, soak, SNR/BER, run_benchmark
mkdir -p gpurun_out
O=gpurun_out/r5p
PROFILE_STEPS=5 PROFILE_WARMUP=2 bash scripts/profile.sh r5_k15 --config 4 > ${O}_profile.log 2>&1; tail -2 ${O}_profile.log
VIT_TEST_K15_ALL=1 timeout -k 10 900 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "cassini_k15" > ${O}_k15_all.log 2>&1; echo "K15_ALL rc=$?"; tail -2 ${O}_k15_all.log
timeout -k 10 200 python -u tests/soak_fuzz.py 60 950000 > ${O}_soak.log 2>&1; echo "soak rc=$?"; tail -1 ${O}_soak.log
timeout -k 10 900 python -m viterbidecodercpp_amd.tools.run_snr_ber --codes 2 5 7 --decode-types SOFT16 HARD8 --bits-scale 4 > gpurun_out/r5_snr_ber_hip.json 2> ${O}_snr.log; echo "snr rc=$?"; tail -2 ${O}_snr.log
timeout -k 10 600 python -m viterbidecodercpp_amd.tools.run_benchmark -T 0.3 > gpurun_out/r5_run_benchmark_hip.json 2> ${O}_rb.log; echo "run_benchmark rc=$?"; tail -2 ${O}_rb.log
