"""per-kernel table of the LatentRNN step (bench.kernel_table: HIP events per launch, side streams off)"""
import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
sys.stdout = sys.stderr
wl = bench.LatentWorkload(torch.device("cuda", 0), 0)
for _ in range(3): wl.step()
torch.cuda.synchronize()
t = bench.kernel_table(wl.step, nprof=3)
tot = sum(r["ms_per_step"] for r in t)
print(f"total kernel time per step {tot:.3f} ms, {sum(r['launches_per_step'] for r in t):.0f} launches")
for r in t[:40]:
    print(f'{r["kernel"]:<52} n/step {r["launches_per_step"]:<5} avg {r["avg_us"]:8.1f} us  ms/step {r["ms_per_step"]:.3f}  {r["tflops"]:7.1f} TF/s  frac {r["frac_mfma"]:.2f} ({r["mfma_pipe"]})')
