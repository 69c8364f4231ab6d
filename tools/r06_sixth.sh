#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_inference.py tests/test_gpu_arnn.py tests/test_gpu_chain_stress.py -m gpu -q -x > gpurun_out/r06_f_tests.log 2>&1; tail -5 gpurun_out/r06_f_tests.log
timeout 600 python tools/decode_latency.py > gpurun_out/r06_f_decode_latency.txt 2>&1; grep -v amdgpu.ids gpurun_out/r06_f_decode_latency.txt
INET_DECODE_B1_STAMPS=1 timeout 600 python tools/decode_latency.py 2>&1 | grep -A12 "mean us per phase" > gpurun_out/r06_f_decode_stamps.txt; cat gpurun_out/r06_f_decode_stamps.txt
