#!/usr/bin/env python3
"""How far ahead of the GPU is the host at the forward -> backward boundary?  A busy-wait of D microseconds is inserted
there; the step time grows only by what the host's lead cannot absorb."""
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


def spin(us):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e6 < us:
        pass


def run(where, delay, n=150):
    def step():
        trainer.zero_grad()
        if where == "start": spin(delay)
        loss, acc = trainer.loss_and_acc_for_batch(tok, 0, train=True)
        if where == "mid": spin(delay)
        loss.backward()
        if where == "end": spin(delay)
        trainer.step()
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# a delay in front of one backward entry point: is the host ahead of the GPU INSIDE the backward pass?
from inpaintnet_amd import ops as _ops
for name in ("encoder_bwd", "decoder_bwd", "latent_bwd"):
    orig = getattr(_ops, name)
    res = []
    for d in (0, 100, 300):
        def wrapped(*a, _o=orig, _d=d, **k):
            spin(_d)
            return _o(*a, **k)
        setattr(_ops, name, wrapped)
        res.append(f"D={d}us: {run('none', 0):.3f} ms")
    setattr(_ops, name, orig)
    print("before", name, " ".join(res), flush=True)
for where in ("mid", "start", "end"):
    print(where, " ".join(f"D={d}us: {run(where, d):.3f} ms" for d in (0, 100, 300, 1000)), flush=True)
