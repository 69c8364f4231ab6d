"""per-kernel table of the B=256 step (bench.kernel_table: HIP events per launch, side streams off) for the chain kernels"""
import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
for _ in range(5): wl.step()
torch.cuda.synchronize()
t = bench.kernel_table(wl.step, nprof=6)
for r in t:
    if "chain" in r["kernel"] or "bf3" in r["kernel"] or r["ms_per_step"] > 0.1:
        print(f'{r["kernel"]:<52} n/step {r["launches_per_step"]:<5} avg {r["avg_us"]:8.1f} us  {r["tflops"]:7.1f} TF/s')
