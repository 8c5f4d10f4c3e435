mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_resume.py tests/test_gpu_fuzz.py tests/test_gpu_api.py -x -q -m gpu -k "CDMA or K9 or k9 or 9- or runtime or 5- or long_frames or ragged or custom" > gpurun_out/r3_k9_tests.log 2>&1; echo rc=$?
tail -3 gpurun_out/r3_k9_tests.log
for lib in build_ab/libvit_hip_prev.so viterbidecodercpp_amd/libvit_hip.so build_ab/libvit_hip_prev.so viterbidecodercpp_amd/libvit_hip.so; do
VIT_HIP_LIB_PATH=$PWD/$lib python scripts/time_update.py 5 SOFT16 65536 8192 3 2>&1 | grep -v amdgpu.ids
VIT_HIP_LIB_PATH=$PWD/$lib python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('   bench', round(r['value']), r['ms_per_step'], r['ms_per_step_median'], r['update_ms'], r['chainback_ms'])"
done
