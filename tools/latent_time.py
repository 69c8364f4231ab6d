"""LatentRNN training step time (bench.LatentWorkload, 128 sequences x 16 measures): python tools/latent_time.py"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
sys.stdout = sys.stderr
wl = bench.LatentWorkload(torch.device("cuda", 0), 0)
for _ in range(5): wl.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30): wl.step()
torch.cuda.synchronize()
print(f"latent step {1e3 * (time.perf_counter() - t0) / 30:.3f} ms")
