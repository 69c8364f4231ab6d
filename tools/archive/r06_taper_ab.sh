#!/bin/bash
# the tapered last chunk of the two-layer LSTM pipelines against the previous build (build/lib_notaper.so), alternating on one box
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_arnn.py -m gpu -q -x > gpurun_out/r06_k_tests.log 2>&1; tail -3 gpurun_out/r06_k_tests.log
for i in 1 2 3; do
  timeout 300 python tools/arnn_time.py 2>&1 | grep -o "'ms_per_step': [0-9.]*, 'ms_per_step_free_running': [0-9.]*" | head -1 | sed "s/^/tapered last chunk (16, 8, 8): /"
  INET_LIB_PATH=build/lib_notaper.so timeout 300 python tools/arnn_time.py 2>&1 | grep -o "'ms_per_step': [0-9.]*, 'ms_per_step_free_running': [0-9.]*" | head -1 | sed "s/^/twelve chunks of 32:           /"
done | tee gpurun_out/r06_k_taper_ab.txt
