#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_chain_stress.py tests/test_gpu_dp.py tests/test_gpu_entry.py -m gpu -q -x > gpurun_out/r06_c_tests.log 2>&1; tail -3 gpurun_out/r06_c_tests.log
python tools/arnn_anomaly_rep.py 2>/dev/null | tail -1 >> gpurun_out/r06_anomaly_d_reps.jsonl
for br in tf fr; do python tools/vae4096_table.py 4096 $br 2>/dev/null; done > gpurun_out/r06_c_vae4096_table.txt
cat gpurun_out/r06_c_vae4096_table.txt
INET_CHAIN_CHUNK_MAX=4096 python tools/vae4096_table.py 4096 tf 2>/dev/null > gpurun_out/r06_c_vae4096_table_chunked.txt
head -30 gpurun_out/r06_c_vae4096_table_chunked.txt
