#!/usr/bin/env python3
"""Experiment: LatentRNN's optimizer launch (Adam over 160 MB + the zeroing of the gradient store) on a second stream, under the NEXT
step's frozen-encoder forward, which touches neither the LatentRNN's parameters nor its gradients; the main stream waits in front
of the first context GRU.  python tools/latent_late_adam.py [0|1]"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from inpaintnet_amd import ops
from inpaintnet_amd.latent_rnn import LatentRNN
from inpaintnet_amd.trainer import Trainer
mode = sys.argv[1] if len(sys.argv) > 1 else "1"
late = mode != "0"
sys.stdout = sys.stderr
wl = bench.LatentWorkload(torch.device("cuda", 0), 0)
if late:
    lane = ops.twin_stream() if mode == "twin" else torch.cuda.Stream()
    state = {"pending": False}
    orig_launch, orig_zero, orig_ctx = Trainer._launch_optimizer, Trainer.zero_grad, LatentRNN.forward_context

    def launch(self, tag, gscale, flag):
        cur = torch.cuda.current_stream()
        lane.wait_stream(cur)
        with torch.cuda.stream(lane):
            orig_launch(self, tag, gscale, flag)
            self.model._grad_store.zero_()
        state["pending"] = True

    def zero_grad(self):
        if state["pending"]:                       # the store is zeroed behind the late Adam
            from inpaintnet_amd import dp
            dp.reset_buckets(self.model.grad)
            self.model.__dict__.pop("_dp_open", None)
            ops.side_defer(bool(self.overlap_backward))
            return
        orig_zero(self)

    def ctx(self, z, type):
        if state["pending"]:
            torch.cuda.current_stream().wait_stream(lane)
            state["pending"] = False
        return orig_ctx(self, z, type)
    Trainer._launch_optimizer, Trainer.zero_grad, LatentRNN.forward_context = launch, zero_grad, ctx
for _ in range(5): wl.step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(30): loss = wl.step()
    torch.cuda.synchronize()
    print(f"late={mode} latent step {1e3 * (time.perf_counter() - t0) / 30:.3f} ms  loss {float(loss.detach()):.5f}")
