mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_resume.py tests/test_gpu_fuzz.py tests/test_gpu_api.py tests/test_gpu_punctured.py tests/test_gpu_soak.py -x -q -m gpu > gpurun_out/r3_k7_tests.log 2>&1; rc=$?
tail -8 gpurun_out/r3_k7_tests.log
exit $rc
