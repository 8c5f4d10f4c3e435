# randomised soak on the final kernels, both chainback kernels of K7 / K9 forced in turn (VIT_HIP_CHAINBACK_ALT).  The logs go
# straight to files under gpurun_out/ (a pipe would hold the progress lines back and the run would look hung)
mkdir -p gpurun_out
SECS=${SOAK_SECONDS:-360}
VIT_HIP_CHAINBACK_ALT=0 python -u tests/soak_fuzz.py $SECS 500000 > gpurun_out/r3_soak_alt0.log 2>&1 && tail -1 gpurun_out/r3_soak_alt0.log &&
VIT_HIP_CHAINBACK_ALT=1 python -u tests/soak_fuzz.py $SECS 600000 > gpurun_out/r3_soak_alt1.log 2>&1 && tail -1 gpurun_out/r3_soak_alt1.log
