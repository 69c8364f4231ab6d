"""eval-mode decode latency at small batch (bench.decode_latency_extra): python tools/decode_latency.py"""
import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
sys.stdout = sys.stderr
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
r = bench.decode_latency_extra(wl.model, iters=200)["decoder_eval"]
print({k: v["ms_per_call"] for k, v in r.items() if isinstance(v, dict)})
# b = 1 under the decode kernels (inet_set_option key 15: 0 = decode_chain.hip's exchange kernel, 1 / 2 = decode_b1.hip behind the beat
# path's launches, 3 = decode_b1.hip with the beat path folded in)
from inpaintnet_amd import ops
import time
vae = wl.model
vae.eval()
z = torch.randn(1, vae.latent_space_dim, device="cuda")
dummy = torch.zeros(1, 24, device="cuda")
for mode in (0, 1, 2, 3):
    ops.set_option(15, mode)
    with torch.no_grad():
        for _ in range(5):
            vae.decoder(z, dummy, train=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            vae.decoder(z, dummy, train=False)
        torch.cuda.synchronize()
    print(f"b = 1, decode kernel mode {mode}: {1e3 * (time.perf_counter() - t0) / 300:.4f} ms per call, chain status {ops.chain_status()}")
ops.set_option(15, 3)
