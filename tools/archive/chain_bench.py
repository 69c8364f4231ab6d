"""Micro-benchmark of the recurrent layers at the bench shape (B=256, H=512, T=24): encoder forward + backward through the
C-ABI with per-launch event timing (chain kernels vs per-step launches).   python tools/chain_bench.py"""
import csv, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inpaintnet_amd import ops, synthetic
from tests.golden_util import vae_params
from tests.test_gpu_kernels import pack

cfg = ops.vae_config(48)
table, total = ops.vae_param_table(cfg)
params = pack(table, total, vae_params("full"))
grads = torch.zeros_like(params)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tok = torch.from_numpy(synthetic.det_tokens("cb", (B, 24), 48)).cuda()
mask = ops.dropout_mask((24, B, 1024), 0.5, 1, 0, "cuda")
dmu = torch.randn(B, 256, device="cuda"); dls = torch.randn(B, 256, device="cuda")
ops.set_option(0, 0)
for chain in (1, 0):
    ops.set_option(4, chain)
    for it in range(3):
        if it == 2:
            torch.cuda.synchronize(); ops.prof_enable(True)
        mu, ls, ws = ops.encoder_fwd(cfg, tok, params, mask=mask, save=True)
        ops.encoder_bwd(cfg, tok, params, grads, mask, dmu, dls, ws)
    torch.cuda.synchronize()
    with tempfile.TemporaryDirectory() as td:
        ops.prof_dump(os.path.join(td, "l.csv")); rows = list(csv.DictReader(open(os.path.join(td, "l.csv"))))
    ops.prof_enable(False)
    agg = {}
    for r in rows:
        if r["label"].startswith("gru"):
            a = agg.setdefault(r["label"], []); a.append(float(r["us"]))
    for k, v in agg.items():
        steps = 24 if "chain" in k else 1
        each = " ".join(f"{x / steps:.2f}" for x in v) if "chain" in k else ""
        print(f"chain={chain} {k:<44} n={len(v):<3} avg {sum(v) / len(v):8.2f} us   per step {sum(v) / len(v) / steps:6.2f} us  {each}")
print("status", ops.chain_status())
