mkdir -p gpurun_out
python -m pytest tests/test_gpu_golden.py tests/test_gpu_punctured.py tests/test_gpu_cpp.py -x -q -m gpu > gpurun_out/r3_stream_tests.log 2>&1; echo rc=$?
tail -5 gpurun_out/r3_stream_tests.log
python scripts/host_route_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_host_route_latency.txt
