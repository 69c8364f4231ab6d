"""Go / no-go arithmetic for splitting the B = 256 step into two software-pipelined 128-row micro-batches.

A micro-batch pipeline can only pay if a 128-row step costs clearly less than a 256-row one -- otherwise two of them back to back
(whatever overlaps with whatever) are slower than the step they replace.  The chain kernels are hand-off-latency-bound: a step of
the recurrence costs the same at 128 rows as at 256 (the same number of dependent exchanges, half-full MFMA tiles), so halving the
batch halves only the GEMM part.  This measures it: the full MeasureVAE training step at B = 64 / 128 / 256 / 512, and at B = 128
with the chain launches restricted to half the chip (INET_CHAIN_CUS=128 -- what a pipelined micro-batch would get).

    python3 tools/batch_scaling.py
"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
sys.stdout = sys.stderr
from inpaintnet_amd import dp, synthetic

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dp.seed_rank(1234, 0)
dp.seed_shared(4321)
wl = bench.VaeWorkload(dev, 0)
res = {}
for B in (64, 128, 256, 512):
    wl.tokens = torch.from_numpy(synthetic.det_tokens(f"scal/{B}", (B, 24), bench.NUM_NOTES)).to(dev)
    dt, _ = bench.timed(wl.step, 200, 20, torch.cuda.synchronize)
    res[B] = 1e3 * dt / 200
    print(f"B = {B:4d}: {res[B]:.3f} ms per step = {B / res[B]:.1f} k measures/s")
wl.trainer.check_steps(wait_all=True)
print(f"two 128-row steps back to back: {2 * res[128]:.3f} ms against one 256-row step {res[256]:.3f} ms "
      f"(ratio {2 * res[128] / res[256]:.2f}; a pipeline would have to hide {2 * res[128] - res[256]:.3f} ms to break even, "
      f"and its target was <= 3.40 ms)")
