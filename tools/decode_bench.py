"""Latency of HierarchicalDecoder.forward in eval mode (free-running) at small batch: fused decode kernel vs per-tick."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from inpaintnet_amd import ops, synthetic
from inpaintnet_amd.measure_vae import MeasureVAE
ds = synthetic.SyntheticFolkDataset(num_notes=48)
vae = MeasureVAE(ds)
for chain in (1, 0):
    ops.set_option(4, chain)
    r = bench.decode_latency_extra(vae, iters=50)["decoder_eval"]
    print("chain", chain, {k: (v["ms_per_call"], v["weights_once_GBps"]) for k, v in r.items() if k.startswith("b")}, "status", ops.chain_status())
import csv, tempfile
ops.set_option(4, 1)
vae.eval()
for b in (1, 16):
    z = torch.randn(b, 256, device="cuda"); dummy = torch.zeros(b, 24, device="cuda")
    with torch.no_grad():
        vae.decoder(z, dummy, train=False); torch.cuda.synchronize()
        ops.prof_enable(True)
        vae.decoder(z, dummy, train=False); torch.cuda.synchronize()
    with tempfile.TemporaryDirectory() as td:
        ops.prof_dump(os.path.join(td, "l.csv")); rows = list(csv.DictReader(open(os.path.join(td, "l.csv"))))
    ops.prof_enable(False)
    agg = {}
    for r in rows:
        k = r["label"].split(" ")[0] if not r["label"].startswith("M") else "gemm"
        a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["us"])
    print("b", b, {k: (n, round(us, 1)) for k, (n, us) in agg.items()})
