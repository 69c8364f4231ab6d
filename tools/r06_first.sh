#!/bin/bash
# round 6, first GPU call: the suite, the driver's command, ten fresh-process short runs (no priming any more)
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r06_a_gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_a_gpu_tests.log
tail -3 gpurun_out/r06_a_gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_a_bench_line.json 2> gpurun_out/r06_a_bench.err; echo "bench rc $?"
wc -c gpurun_out/r06_a_bench_line.json; cat gpurun_out/r06_a_bench_line.json
cp bench_detail.json gpurun_out/r06_a_bench_detail.json
for i in 1 2 3 4 5 6 7 8 9 10; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-roofline --no-parity 2>/dev/null >> gpurun_out/r06_a_short_x10.txt
done
python - <<'PY'
import json
for l in open("gpurun_out/r06_a_short_x10.txt"):
    d = json.loads(l); print(d["value"], d["ms_per_step"], d.get("slow_waits"), max(d["first_steps_ms"]))
PY
