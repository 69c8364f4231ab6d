#!/usr/bin/env python3
"""Is the host ahead of the GPU?  Host-side enqueue time of each phase of a training step (no syncs) vs the GPU time per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inpaintnet_amd import synthetic
from inpaintnet_amd.measure_vae import MeasureVAE
from inpaintnet_amd.vae_trainer import VAETrainer
ds = synthetic.SyntheticFolkDataset(num_notes=48)
model = MeasureVAE(ds); trainer = VAETrainer(ds, model); model.train()
trainer.overlap_backward = True
tok = torch.from_numpy(synthetic.det_tokens("prof", (256, 24), 48)).cuda()
marks = []
def step():
    t = [time.perf_counter()]
    trainer.zero_grad(); t.append(time.perf_counter())
    loss, acc = trainer.loss_and_acc_for_batch(tok, 0, train=True); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    trainer.step(); t.append(time.perf_counter())
    marks.append(t)
for _ in range(5): step()
torch.cuda.synchronize(); marks.clear()
t0 = time.perf_counter()
for _ in range(30): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
import numpy as np
m = np.array(marks)
d = np.diff(m, axis=1) * 1e3
print("host ms per phase (median over 30 steps): zero_grad %.3f  forward+loss %.3f  backward %.3f  step %.3f  | total %.3f" % (*np.median(d, axis=0), np.median(m[:, -1] - m[:, 0]) * 1e3))
print("host loop %.3f ms/step; with final sync %.3f ms/step" % ((t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
print("first 6 steps host totals:", np.round((m[:6, -1] - m[:6, 0]) * 1e3, 2))
