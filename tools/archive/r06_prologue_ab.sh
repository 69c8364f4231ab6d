#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_inference.py tests/test_gpu_model.py -m gpu -q -x > gpurun_out/r06_l_tests.log 2>&1; tail -3 gpurun_out/r06_l_tests.log
timeout 600 python tools/decode_latency.py 2>&1 | grep -v amdgpu | head -8
bash tools/ab_bench.sh "INET_X=1" 2>&1 | tail -2
