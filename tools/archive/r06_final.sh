#!/bin/bash
mkdir -p gpurun_out
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_z_bench_line.json 2> gpurun_out/r06_z_bench.err; echo "bench rc $? $(wc -c < gpurun_out/r06_z_bench_line.json) bytes"
cat gpurun_out/r06_z_bench_line.json; cp bench_detail.json gpurun_out/r06_z_bench_detail.json
python bench.py --gpus 1 --steps 400 --warmup 20 --no-extras --no-cpu-baseline > gpurun_out/r06_z_bench_400.json 2>/dev/null; cat gpurun_out/r06_z_bench_400.json | cut -c1-400
timeout 600 python tools/decode_latency.py > gpurun_out/r06_z_decode_latency.txt 2>&1
INET_DECODE_B1_STAMPS=1 timeout 600 python tools/decode_latency.py 2>&1 | grep -A12 "mean us per phase" > gpurun_out/r06_z_decode_stamps.txt
grep -v amdgpu gpurun_out/r06_z_decode_latency.txt | head -20
python -c "import __graft_entry__ as g; g.smoke()"
