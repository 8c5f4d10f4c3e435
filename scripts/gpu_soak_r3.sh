# randomised soak on the final kernels, both chainback kernels of K7 / K9 forced in turn (VIT_HIP_CHAINBACK_ALT)
mkdir -p gpurun_out
SECS=${SOAK_SECONDS:-360}
VIT_HIP_CHAINBACK_ALT=0 python tests/soak_fuzz.py $SECS 500000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_soak_alt0.log | tail -3 &&
VIT_HIP_CHAINBACK_ALT=1 python tests/soak_fuzz.py $SECS 600000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_soak_alt1.log | tail -3
