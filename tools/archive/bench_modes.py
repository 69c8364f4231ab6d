#!/usr/bin/env python3
"""ms per MeasureVAE training step with the teacher-forcing coin pinned (B=256)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inpaintnet_amd import synthetic
from inpaintnet_amd.measure_vae import MeasureVAE
from inpaintnet_amd.vae_trainer import VAETrainer
ds = synthetic.SyntheticFolkDataset(num_notes=48)
model = MeasureVAE(ds); trainer = VAETrainer(ds, model); model.train()
trainer.overlap_backward = os.environ.get('OVERLAP', '1') != '0'
tok = torch.from_numpy(synthetic.det_tokens("prof", (256, 24), 48)).cuda()
def step(tf):
    trainer.zero_grad()
    w, s, zd, pd, z, zp = model(tok, train=True, teacher_forced=tf)
    ce, acc = trainer.mean_crossentropy_loss_and_accuracy(w, tok)
    (ce + trainer.compute_kld_loss(zd, pd)).backward()
    trainer.step()
for tf in (True, False):
    for _ in range(5): step(tf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step(tf)
    torch.cuda.synchronize()
    print("teacher_forced=%s: %.3f ms/step" % (tf, (time.perf_counter() - t0) / 30 * 1e3))
