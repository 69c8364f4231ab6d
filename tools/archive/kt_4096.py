"""per-kernel table of the 4096-measure MeasureVAE step (the reference's default batch): python tools/kt_4096.py [batch]"""
import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from inpaintnet_amd import synthetic
sys.stdout = sys.stderr
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
tok = torch.from_numpy(synthetic.det_tokens("bench/4096", (B, 24), bench.NUM_NOTES)).cuda()
t = wl.trainer
import random
random.seed(3)
def step():
    t.zero_grad()
    loss, acc = t.loss_and_acc_for_batch(tok, 0, train=True)
    loss.backward()
    t.step()
for _ in range(2): step()
torch.cuda.synchronize()
tab = bench.kernel_table(step, nprof=4)
tot = sum(r["ms_per_step"] for r in tab)
print(f"B={B}: total kernel time per step {tot:.3f} ms, {sum(r['launches_per_step'] for r in tab):.0f} launches")
for r in tab[:24]:
    print(f'{r["kernel"]:<52} n/step {r["launches_per_step"]:<6} avg {r["avg_us"]:8.1f} us  ms/step {r["ms_per_step"]:7.3f}  {r["tflops"]:7.1f} TF/s  frac {r["frac_mfma"]:.2f} ({r["mfma_pipe"]})')
