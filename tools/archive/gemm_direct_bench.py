#!/usr/bin/env python3
"""Direct (LDS-free) vs LDS-tiled GEMM on the shapes of one training step: time and check both against an fp64 product.
    python tools/gemm_direct_bench.py [M N K akm bkm ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inpaintnet_amd import ops  # noqa: E402

SHAPES = [(1536, 512, 6144, 1, 1), (1536, 1024, 6144, 1, 1), (6144, 1536, 1024, 0, 0), (6144, 1024, 1536, 0, 1),
          (6144, 512, 1536, 0, 1), (6144, 1536, 512, 0, 0),
          # medium / small shapes of the step (workgroup split-K kernel)
          (1536, 512, 1024, 1, 1), (1024, 512, 1024, 1, 1), (1024, 2048, 256, 1, 1), (512, 512, 1024, 1, 1),
          (256, 1024, 256, 1, 1), (1024, 256, 256, 1, 1),
          (1024, 1536, 512, 0, 0), (256, 1024, 2048, 0, 0), (256, 256, 1024, 0, 0), (1024, 512, 512, 0, 0),
          (1024, 1024, 512, 0, 0), (256, 1024, 256, 0, 0),
          (256, 1024, 256, 0, 1), (256, 2048, 1024, 0, 1), (1024, 512, 1536, 0, 1), (1024, 512, 1024, 0, 1),
          (1024, 512, 512, 0, 1), (256, 256, 1024, 0, 1)]


def timed(f, n=30):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    args = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(args[i:i + 5]) for i in range(0, len(args), 5)] or SHAPES
    for M, N, K, akm, bkm in shapes:
        g = torch.Generator().manual_seed(M + N + K)
        A = torch.randn(M, K, generator=g)
        B = torch.randn(N, K, generator=g)
        ref = (A.double() @ B.double().t()).cuda()
        Ad = (A.t().contiguous() if akm else A).cuda()
        Bd = (B.t().contiguous() if bkm else B).cuda()
        C = torch.zeros(M, N, device="cuda")
        row = [f"M{M} N{N} K{K} {'T' if akm else 'N'}{'N' if bkm else 'T'}"]
        splits = [int(x) for x in os.environ.get("SPLITS", "0").split(",")]
        for mode, sp in [(0, 0)] + [(1, x) for x in splits] + ([(2, 0)] if os.environ.get('FORCE2') else []):
            ops.set_option(5, mode)
            ops.set_option(3, sp)
            f = lambda: ops.gemm(Ad, Bd, M, N, K, a_kmajor=akm, b_kmajor=bkm, out=C)
            us = timed(f)
            err = float((C.double() - ref).abs().max() / ref.abs().max())
            ops.prof_enable(True)
            f()
            torch.cuda.synchronize()
            ops.prof_dump("/tmp/_gd.csv")
            lab = open("/tmp/_gd.csv").read().strip().splitlines()[-1].split(",")[1]
            ops.prof_enable(False)
            row.append(f"{('forced' if mode == 2 else 'new   ') if mode else 'tiled '} {us:7.1f} us {2.0 * M * N * K / us / 1e6:6.1f} TF err {err:.1e} [{lab}]")
        ops.set_option(5, 1)
        ops.set_option(3, 0)
        print("\n    ".join(row), flush=True)


if __name__ == "__main__":
    main()
