#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/r06_b_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_b_gpu_tests.log
tail -3 gpurun_out/r06_b_gpu_tests.log
# what the recorder costs the chain kernels: the same sources compiled with -DINET_RECORDER=0, alternating
for i in 1 2 3; do
  bash tools/ab_bench.sh "INET_X=0" "INET_LIB_PATH=build/lib_norec.so"
done 2>&1 | tee gpurun_out/r06_b_recorder_ab.txt
