"""per-launch table of one eval-mode HierarchicalDecoder.forward at batch b (HIP events per launch): python tools/decode_kernels.py [b]"""
import os, sys, csv, tempfile
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from inpaintnet_amd import ops
sys.stdout = sys.stderr
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
vae = wl.model
vae.eval()
z = torch.randn(b, vae.latent_space_dim, device="cuda")
dummy = torch.zeros(b, 24, device="cuda")
with torch.no_grad():
    for _ in range(3):
        vae.decoder(z, dummy, train=False)
    torch.cuda.synchronize()
    ops.prof_enable(True)
    vae.decoder(z, dummy, train=False)
    torch.cuda.synchronize()
with tempfile.TemporaryDirectory() as td:
    ops.prof_dump(td + "/l.csv")
    rows = list(csv.DictReader(open(td + "/l.csv")))
ops.prof_enable(False)
tot = 0.0
for r in rows:
    tot += float(r["us"])
    print(f'{r["label"]:<56} {float(r["us"]):8.1f} us')
print(f"sum of profiled launches {tot:.1f} us over {len(rows)} launches")
