#!/bin/bash
mkdir -p gpurun_out
python tools/arnn_anomaly_rep.py 2>/dev/null | tail -1 | cut -c1-330
for i in 1 2; do
  timeout 300 python tools/arnn_time.py 2>&1 | grep -o "'ms_per_step': [0-9.]*, 'ms_per_step_free_running': [0-9.]*" | head -1 | sed "s/^/rs hand-off:      /"
  INET_LIB_PATH=build/lib_norsbwd.so timeout 300 python tools/arnn_time.py 2>&1 | grep -o "'ms_per_step': [0-9.]*, 'ms_per_step_free_running': [0-9.]*" | head -1 | sed "s/^/counter protocol: /"
done | tee gpurun_out/r06_j_arnn_ab.txt
