#!/usr/bin/env python3
"""Untraced phase times of the B=256 MeasureVAE training step: torch events on the main stream around every C-ABI module
call (encoder / decoder forward and backward), the loss glue between them and the optimizer.  rocprofv3's kernel trace
inflates every small kernel by a few microseconds; this does not (one event record per boundary, ~10 per step)."""
import os, sys, collections
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inpaintnet_amd import ops, synthetic
from inpaintnet_amd.measure_vae import MeasureVAE
from inpaintnet_amd.vae_trainer import VAETrainer
import random
ds = synthetic.SyntheticFolkDataset(num_notes=48)
model = MeasureVAE(ds); trainer = VAETrainer(ds, model); model.train()
model.load_state_dict({k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape))) for k, v in model.state_dict().items()})
trainer.overlap_backward = True
tok = torch.from_numpy(synthetic.det_tokens("prof", (256, 24), 48)).cuda()
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
def wrap(name):
    f = getattr(ops, name)
    def g(*a, **k):
        mark("<" + name); r = f(*a, **k); mark(">" + name); return r
    setattr(ops, name, g)
for n in ("encoder_fwd", "decoder_fwd", "decoder_bwd", "encoder_bwd", "adam_step"):
    wrap(n)
TF = None if len(sys.argv) < 2 else (sys.argv[1] == "tf")
def step(i):
    mark("step")
    trainer.zero_grad()
    if TF is None:
        loss, acc = trainer.loss_and_acc_for_batch(tok, 0, train=True)
    else:
        import inpaintnet_amd.measure_vae as MV
        w, s, zd, pd, zt, zp = model(tok, train=True, teacher_forced=TF)
        loss = trainer.mean_crossentropy_loss_and_accuracy(w, tok)[0] + trainer.compute_kld_loss(zd, pd)
    loss.backward(); trainer.step()
random.seed(0)
for i in range(10): step(i)
torch.cuda.synchronize(); marks.clear()
N = 40
for i in range(N): step(i)
mark("step")
torch.cuda.synchronize()
seg = collections.defaultdict(list)
for (a, ea), (b, eb) in zip(marks, marks[1:]):
    seg[a + " .. " + b].append(ea.elapsed_time(eb) * 1e3)
tot = marks[0][1].elapsed_time(marks[-1][1]) * 1e3 / N
print(f"step {tot:.1f} us (events on the main stream, {N} steps)")
for k, v in seg.items():
    print(f"  {k:<44} n={len(v):<3d} mean {sum(v)/len(v):8.1f} us   min {min(v):8.1f}")
