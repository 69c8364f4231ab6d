#!/usr/bin/env python3
"""Forced tile/split sweep of one GEMM shape (INET_GEMM_FORCE is read once per process, so one child per point).
    python tools/gemm_sweep.py M N K akm bkm"""
import os
import subprocess
import sys

if len(sys.argv) > 6:           # child
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from inpaintnet_amd import ops
    M, N, K, akm, bkm = map(int, sys.argv[1:6])
    A = torch.randn((K, M) if akm else (M, K), device="cuda")
    B = torch.randn((K, N) if bkm else (N, K), device="cuda")
    C = torch.zeros(M, N, device="cuda")
    f = lambda: ops.gemm(A, B, M, N, K, a_kmajor=akm, b_kmajor=bkm, out=C)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        f()
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 30 * 1e3
    print(f"{us:8.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF", end="")
    sys.exit(0)

args = sys.argv[1:6]
tiles = ["64x64", "128x128", "192x64", "192x128", "192x192"]
print("shape", args)
env = dict(os.environ)
env.pop("INET_GEMM_FORCE", None)
print("  model's pick:", subprocess.run([sys.executable, __file__] + args + ["child"], env=env, capture_output=True, text=True).stdout)
for ci, tname in enumerate(tiles):
    row = []
    for sp in (1, 2, 4, 8, 16):
        env["INET_GEMM_FORCE"] = f"{ci},{sp}"
        r = subprocess.run([sys.executable, __file__] + args + ["child"], env=env, capture_output=True, text=True)
        row.append(f"s{sp}:{r.stdout.strip()}")
    print(f"  {tname:<8}", " | ".join(row), flush=True)
