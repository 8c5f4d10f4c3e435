mkdir -p gpurun_out
: > gpurun_out/r3_k15_ab.log
for rep in 1 2; do
for lib in build_ab/libvit_hip_base.so build_ab/libvit_hip_prev.so viterbidecodercpp_amd/libvit_hip.so; do
VIT_HIP_LIB_PATH=$PWD/$lib python scripts/time_update.py 7 SOFT16 4096 8192 3 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3_k15_ab.log
done
done
cat gpurun_out/r3_k15_ab.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_resume.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r3_k15_tests.log 2>&1; echo rc=$?
tail -3 gpurun_out/r3_k15_tests.log
