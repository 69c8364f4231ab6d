#!/usr/bin/env python3
"""Per-launch event timing of ONE MeasureVAE training step (teacher-forced and free-running), grouped by GEMM shape."""
import collections
import csv
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inpaintnet_amd import _lib, ops, synthetic  # noqa: E402
from inpaintnet_amd.measure_vae import MeasureVAE  # noqa: E402
from inpaintnet_amd.vae_trainer import VAETrainer  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/step_prof.csv"
ds = synthetic.SyntheticFolkDataset(num_notes=48)
model = MeasureVAE(ds)
trainer = VAETrainer(ds, model)
model.train()
tok = torch.from_numpy(synthetic.det_tokens("prof", (256, 24), 48)).cuda()


def step(tf):
    trainer.zero_grad()
    w, s, zd, pd, z, zp = model(tok, train=True, teacher_forced=tf)
    ce, acc = trainer.mean_crossentropy_loss_and_accuracy(w, tok)
    (ce + trainer.compute_kld_loss(zd, pd)).backward()
    trainer.step()


_lib.lib().inet_set_option(0, 0)          # serial: clean per-kernel times
for _ in range(3):
    step(True); step(False)
torch.cuda.synchronize()
ops.prof_enable(True)
step(False)
torch.cuda.synchronize()
_lib.lib().inet_prof_dump(out.encode())
ops.prof_enable(False)
rows = list(csv.DictReader(open(out)))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    k = (r["class"], r["label"] if r["class"] == "0" else "gflop=%.2f" % float(r["gflop"]))
    agg[k][0] += 1; agg[k][1] += float(r["us"]); agg[k][2] += float(r["gflop"])
tot = sum(v[1] for v in agg.values())
print(f"total {tot:.0f} us in {len(rows)} MFMA-class launches (free-running step)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"cls{k[0]} {k[1]:<44} n={v[0]:<3} {v[1]:8.1f} us  avg {v[1] / v[0]:7.1f}  {v[2] / v[1] * 1e3 if v[1] else 0:6.1f} TF")
