#!/usr/bin/env python3
"""Micro-benchmarks of the two MFMA kernel classes at the shapes of the MeasureVAE training step
(B=256, H=512, T=24).  Times each shape with HIP events on the launch stream (torch's current stream)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inpaintnet_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def gemm_case(M, N, K, akm, bkm, tag):
    A = torch.randn((K, M) if akm else (M, K), device=DEV)
    B = torch.randn((K, N) if bkm else (N, K), device=DEV)
    C = torch.empty(M, N, device=DEV)
    us = timeit(lambda: ops.gemm(A, B, M, N, K, a_kmajor=akm, b_kmajor=bkm, out=C))
    print(f"gemm {tag:<28} M={M:<5} N={N:<5} K={K:<5} {'T' if akm else 'N'}{'N' if bkm else 'T'}  {us:9.1f} us  "
          f"{2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)


def gru_case(B, H):
    gi = torch.randn(B, 3 * H, device=DEV)
    h = torch.randn(B, H, device=DEV)
    W = torch.randn(3 * H, H, device=DEV) / H ** 0.5
    b = torch.randn(3 * H, device=DEV)
    us = timeit(lambda: ops.gru_step(gi, h, W, b, save=False))
    print(f"gru_step_fwd B={B:<5} H={H:<5} {us:9.1f} us  {2.0 * B * 3 * H * H / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    gru_case(256, 512)
    gru_case(1024, 512)
    gru_case(128, 1024)
    gemm_case(6144, 1536, 1024, 0, 0, "enc gi1 (NT)")
    gemm_case(6144, 1024, 1536, 0, 1, "enc dx1 (NN)")
    gemm_case(1536, 1024, 6144, 1, 1, "enc dW_ih_l1 (TN)")
    gemm_case(1536, 512, 6144, 1, 1, "dW_hh (TN)")
    gemm_case(6144, 512, 1536, 0, 1, "tick dx1 (NN)")
    gemm_case(1024, 1536, 512, 0, 0, "beat gi / cgi (NT)")
    gemm_case(256, 1024, 2048, 0, 0, "enc head (NT)")
    gemm_case(256, 48, 512, 0, 0, "logits (NT)")
    gemm_case(48, 1536, 6144, 1, 1, "dTable (TN)")
    gemm_case(1024, 512, 1536, 0, 1, "beat dgrad (NN)")
    gemm_case(1024, 2048, 256, 1, 1, "small wgrad (TN)")
    gemm_case(256, 1024, 256, 0, 1, "head dgrad (NN)")
    gemm_case(1536, 512, 1024, 1, 1, "beat dW (TN)")
    gemm_case(4096, 4096, 4096, 0, 0, "square (NT)")
