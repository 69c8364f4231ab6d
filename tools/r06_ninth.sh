#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_inference.py -m gpu -q -x > gpurun_out/r06_i_tests.log 2>&1; tail -3 gpurun_out/r06_i_tests.log
timeout 600 python tools/big_vocab_time.py 2> gpurun_out/r06_i_big_vocab.txt; grep "V =" gpurun_out/r06_i_big_vocab.txt
for v in 48 100; do timeout 300 python tools/arnn_token_pass.py $v 2>&1 | grep "mode" ; done | tee -a gpurun_out/r06_i_big_vocab.txt
