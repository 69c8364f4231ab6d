"""Where a LatentRNN step with auto_reg=True goes (the script default, train_inpaintnet.py:53): python tools/latent_ar_profile.py [fr|tf|nar]
Per step: wall time with the queue kept full, HOST time to queue one step (no synchronisation inside), and the host's top
functions by cumulative time (cProfile over 20 steps).  fr = free-running side of the coin (decode -> re-encode per generated
measure), tf = teacher-forced, nar = auto_reg=False."""
import cProfile
import os
import pstats
import random
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fr"
wl = bench.LatentWorkload(torch.device("cuda", 0), 0, auto_reg=(mode != "nar"))
random.random = (lambda: 0.99) if mode == "fr" else (lambda: 0.0)
for _ in range(5):
    wl.step()
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for _ in range(n):
    wl.step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n
host = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.step()
    host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print(f"mode {mode}: {1e3 * wall:.3f} ms per step; host time to queue one step {1e3 * min(host):.3f} .. {1e3 * max(host):.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    wl.step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(12)
table = bench.kernel_table(wl.step, nprof=3)
print(f"kernels sum {sum(r['ms_per_step'] for r in table):.3f} ms per step, {sum(r['launches_per_step'] for r in table):.0f} launches")
print(f"{'kernel':<58} {'n':>5} {'avg us':>9} {'ms/step':>8} {'TFLOP/s':>8} {'pipe':>8} {'frac':>6}")
for r in table[:36]:
    print(f"{r['kernel']:<58} {r['launches_per_step']:>5.0f} {r['avg_us']:>9.1f} {r['ms_per_step']:>8.3f} {r['tflops']:>8.1f} {r['mfma_pipe']:>8} {r['frac_mfma']:>6.3f}")
wl.trainer.finish()
