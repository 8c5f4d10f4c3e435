#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out/r5i
timeout -k 10 900 python -m pytest tests/test_gpu_latency.py tests/test_gpu_golden.py tests/test_gpu_punctured.py tests/test_gpu_cpp.py -x -q -m gpu -s > ${O}_pytest.log 2>&1
echo "pytest rc=$?"; grep "single 8192\|passed\|failed\|Error" ${O}_pytest.log | tail -8
python scripts/host_route_latency.py 2>&1 | grep -v amdgpu.ids | tee ${O}_latency.txt
python bench.py --config 0 --steps 200 --warmup 20 > gpurun_out/r5_bench_config0.json 2> ${O}_bench0.err; echo "config0 rc=$?"; python -c "
import json; r=json.loads([l for l in open('gpurun_out/r5_bench_config0.json') if l.startswith('{')][0]); print({k: r[k] for k in ('value','ms_per_step','update_ms','chainback_ms','ns_per_trellis_step_update','parity','cpu_baseline')})"
./tests/cpp/run_simple_hip 2>&1 | tail -3
