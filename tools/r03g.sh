mkdir -p gpurun_out/r03g
python -m pytest tests/test_gpu_inference.py tests/test_gpu_chain_stress.py -m gpu -q -x > gpurun_out/r03g/gpu_tests.log 2>&1; tail -6 gpurun_out/r03g/gpu_tests.log
cat > /tmp/d.py <<'PY'
import os, sys, json
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
sys.path.insert(0, os.getcwd())
import bench
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
print(os.environ.get("INET_DECODE_FULLV"), json.dumps(bench.decode_latency_extra(wl.model, iters=50)))
PY
python /tmp/d.py 2>&1 | tail -1 > gpurun_out/r03g/decode.txt; INET_DECODE_FULLV=0 python /tmp/d.py 2>&1 | tail -1 >> gpurun_out/r03g/decode.txt; cat gpurun_out/r03g/decode.txt
