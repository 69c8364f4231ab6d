#!/usr/bin/env python3
"""Which workload touches the library first: python tools/arnn_order.py [vae|latent|vae+latent|none]"""
import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
what = sys.argv[1] if len(sys.argv) > 1 else "none"
sys.stdout = sys.stderr
dev = torch.device("cuda", 0)
wl = None
if "vae" in what:
    wl = bench.VaeWorkload(dev, 0)
    for _ in range(60): wl.step()
    torch.cuda.synchronize()
if "latent" in what:
    lw = bench.LatentWorkload(dev, 0, vae=wl.model, ds=wl.ds) if wl is not None else bench.LatentWorkload(dev, 0)
    for _ in range(24): lw.step()
    torch.cuda.synchronize()
    if "del" in what:
        del lw
r = bench.arnn_extra(steps=30, warmup=4, tables=False)["anticipation_rnn_train"]["ms_per_step"]
print(f"first: {what:<12} arnn {r} ms")
