#!/usr/bin/env python3
"""The V > 64 builds of the register-resident kernels, timed once (VERDICT r05 weak 1 ii: the real vocabulary is data-derived,
MeasureVAE/measure_vae.py:56, and nobody knows on which side of 64 it lands): decode calls of 1 .. 16 measures at V = 48 / 80 / 100 / 128
under decode mode 4 (default) and 0 (the exchange kernel).   python tools/big_vocab_time.py"""
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from inpaintnet_amd import ops, synthetic  # noqa: E402
from inpaintnet_amd.measure_vae import MeasureVAE  # noqa: E402

for V in (48, 80, 100, 128):
    ds = synthetic.SyntheticFolkDataset(num_notes=V)
    vae = MeasureVAE(ds)
    vae.eval()
    for b in (1, 4, 16):
        z = torch.randn(b, vae.latent_space_dim, device="cuda")
        dummy = torch.zeros(b, 24, device="cuda")
        line = []
        for mode in (0, 4):
            ops.set_option(15, mode)
            with torch.no_grad():
                for _ in range(5):
                    vae.decoder(z, dummy, train=False)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    vae.decoder(z, dummy, train=False)
                torch.cuda.synchronize()
            line.append(f"mode {mode}: {1e3 * (time.perf_counter() - t0) / 200:.4f} ms")
        print(f"V = {V:3d}  b = {b:2d}: " + ", ".join(line) + f", chain status {ops.chain_status()}", file=sys.stderr)
ops.set_option(15, 4)
