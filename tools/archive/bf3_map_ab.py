"""Kernel-level A/B of the gemm_bf3 tile order (INET_BF3_MAP=0: contiguous tile range per XCD, round 3; 1: XCD blocks):
python tools/bf3_map_ab.py   -- runs itself once per setting in child processes and prints the per-launch times."""
import csv
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(6144, 3072, 1024), (6144, 1024, 3072), (12288, 3072, 1024), (3072, 3072, 2048)]


def child():
    import torch
    from inpaintnet_amd import ops
    for M, N, K in SHAPES:
        A = torch.randn(M, K, device="cuda")
        B = torch.randn(N, K, device="cuda") * 0.05
        C = torch.empty(M, N, device="cuda")
        for _ in range(3):
            ops.gemm_bf3(A, B, M, N, K, out=C, ksplit=1)
        torch.cuda.synchronize()
        ops.prof_enable(True)
        for _ in range(20):
            ops.gemm_bf3(A, B, M, N, K, out=C, ksplit=1)
        torch.cuda.synchronize()
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "l.csv")
            ops.prof_dump(p)
            us = [float(r["us"]) for r in csv.DictReader(open(p)) if r["label"].startswith(f"M{M} N{N} K{K}")]
        ops.prof_enable(False)
        us.sort()
        print(f"  MAP={os.environ.get('INET_BF3_MAP', '1')}  {M}x{N}x{K}: median {us[len(us) // 2]:7.1f} us  min {us[0]:7.1f}  "
              f"({2.0 * M * N * K / us[len(us) // 2] / 1e6:6.1f} TFLOP/s of f32 products)", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for rep in range(2):
            for m in ("0", "1"):
                subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, INET_BF3_MAP=m))
