#!/usr/bin/env python3
"""Per-kernel table of the MeasureVAE training step at the reference's default batch (256 sequences x 16 bars = 4096 measures:
train_measure_vae.py:33, vae_trainer.py:49-52), per coin branch:   python tools/vae4096_table.py [batch] [tf|fr|coin]
Prints ms per step, measures/s and the kernels by summed time (bench.kernel_table: per-launch HIP events, side streams off)."""
import importlib.util
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    branch = sys.argv[2] if len(sys.argv) > 2 else "coin"
    from inpaintnet_amd import synthetic
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench.VaeWorkload(dev, 0)
    tok = torch.from_numpy(synthetic.det_tokens("bench/4096", (batch, 24), bench.NUM_NOTES)).to(dev)
    t = wl.trainer
    if branch in ("tf", "fr"):
        random.random = (lambda: 0.0) if branch == "tf" else (lambda: 0.99)

    def step():
        t.zero_grad()
        loss, acc = t.loss_and_acc_for_batch(tok, 0, train=True)
        loss.backward()
        t.step()
        return loss
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    table = bench.kernel_table(step, nprof=2)
    tot = sum(r["ms_per_step"] for r in table)
    print(f"batch {batch} branch {branch}: {1e3 * dt:.3f} ms per step = {batch / dt:.0f} measures/s; kernels sum {tot:.3f} ms, "
          f"{sum(r['launches_per_step'] for r in table):.0f} launches, {sum(r['gflop_per_launch'] * r['launches_per_step'] for r in table):.0f} GFLOP")
    print(f"{'kernel':<58} {'n':>5} {'avg us':>9} {'ms/step':>8} {'TFLOP/s':>8} {'pipe':>8} {'frac':>6} {'GB/s':>7}")
    for r in table[:40]:
        print(f"{r['kernel']:<58} {r['launches_per_step']:>5.0f} {r['avg_us']:>9.1f} {r['ms_per_step']:>8.3f} {r['tflops']:>8.1f} "
              f"{r['mfma_pipe']:>8} {r['frac_mfma']:>6.3f} {r['gbps']:>7.0f}")
    t.finish()


if __name__ == "__main__":
    main()
