mkdir -p gpurun_out/r03b
python -m pytest tests -m gpu -q -x > gpurun_out/r03b/gpu_tests.log 2>&1; tail -5 gpurun_out/r03b/gpu_tests.log
tools/ab_bench.sh "X=0" "INET_CHAIN_H0PACK=1" "X=1" "INET_CHAIN_H0PACK=1" "INET_GEMM_GROUP=0" "INET_EVENT_FENCE=1" > gpurun_out/r03b/ab.txt 2>&1; cat gpurun_out/r03b/ab.txt
