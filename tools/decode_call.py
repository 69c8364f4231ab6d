"""One b = 1 eval-mode decoder call, repeated (for a kernel trace):  rocprofv3 --kernel-trace ... -- python3 tools/decode_call.py [b]
With a CSV as the first argument instead: the kernels of the LAST call in that trace (start, end, gap, name), i.e. everything
between the last two launches of the fused decode kernel."""
import csv, os, sys
if len(sys.argv) > 1 and sys.argv[1].endswith(".csv"):
    rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "decode_b1_kernel" in r["Kernel_Name"] or "decode_chain_kernel" in r["Kernel_Name"]]
    call = rows[idx[-2] + 1: idx[-1] + 1]
    t0 = int(call[0]["Start_Timestamp"])
    prev = t0
    print(f"{len(call)} kernels, {(int(call[-1]['End_Timestamp']) - t0) / 1e3:.1f} us from the first start to the last end")
    for r in call:
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:70]
        print(f"{(a - t0) / 1e3:8.1f} {(b - t0) / 1e3:8.1f}  gap {(a - prev) / 1e3:6.1f}  dur {(b - a) / 1e3:6.1f}  {name}")
        prev = b
    sys.exit(0)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
sys.stdout = sys.stderr
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
vae = wl.model
vae.eval()
z = torch.randn(b, vae.latent_space_dim, device="cuda")
dummy = torch.zeros(b, 24, device="cuda")
with torch.no_grad():
    for _ in range(30):
        vae.decoder(z, dummy, train=False)
torch.cuda.synchronize()
