mkdir -p gpurun_out/r03e
python -m pytest tests -m gpu -q -x > gpurun_out/r03e/gpu_tests.log 2>&1; tail -12 gpurun_out/r03e/gpu_tests.log
