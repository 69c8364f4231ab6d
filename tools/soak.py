"""Soak run of the three training steps: thousands of steps back to back, then the health words (chain timeouts, skipped optimizer
steps, non-finite parameters, bad tokens) and the final losses.  python tools/soak.py [vae_steps latent_steps arnn_steps]"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from inpaintnet_amd import ops
sys.stdout = sys.stderr
nv, nl, na = [int(a) for a in sys.argv[1:4]] if len(sys.argv) > 3 else (3000, 600, 600)
dev = torch.device("cuda", 0)
wl = bench.VaeWorkload(dev, 0)
t0 = time.time()
for i in range(nv):
    loss = wl.step()
wl.trainer.check_steps(wait_all=True)
torch.cuda.synchronize()
print(f"vae: {nv} steps in {time.time() - t0:.1f} s, final loss {float(loss.detach()):.4f}, adam_t {wl.trainer.adam_t}, "
      f"chain_status {ops.chain_status()}, token_status {ops.token_status()}, lost {wl.trainer.lost_steps}", flush=True)
lw = bench.LatentWorkload(dev, 0, vae=wl.model, ds=wl.ds)
t0 = time.time()
for i in range(nl):
    loss = lw.step()
lw.trainer.check_steps(wait_all=True)
torch.cuda.synchronize()
print(f"latent: {nl} steps in {time.time() - t0:.1f} s, final loss {float(loss.detach()):.4f}, adam_t {lw.trainer.adam_t}, "
      f"chain_status {ops.chain_status()}", flush=True)
la = bench.LatentWorkload(dev, 0, vae=wl.model, ds=wl.ds, auto_reg=True)
t0 = time.time()
for i in range(nl // 2):
    loss = la.step()
la.trainer.check_steps(wait_all=True)
torch.cuda.synchronize()
print(f"latent auto_reg: {nl // 2} steps in {time.time() - t0:.1f} s, final loss {float(loss.detach()):.4f}, chain_status {ops.chain_status()}", flush=True)
r = bench.arnn_extra(steps=na, warmup=3, tables=False, free_steps=max(10, na // 4))["anticipation_rnn_train"]
torch.cuda.synchronize()
print(f"arnn: {na} teacher-forced steps at {r['ms_per_step']} ms, {max(10, na // 4)} free-running steps at {r['ms_per_step_free_running']} ms, "
      f"chain_status {ops.chain_status()}", flush=True)
assert ops.chain_status() == 0 and ops.token_status() == 0
print("soak ok")
# round 5: the one-row persistent kernels back to back with training steps in between (their granule areas are re-zeroed per call, their
# 13 / 73 workgroups must all become resident next to whatever the previous step left running on the side streams)
vae = wl.model
lat = bench.LatentWorkload(dev, 0, vae=vae, ds=wl.ds)
z = torch.randn(1, vae.latent_space_dim, device=dev)
dummy = torch.zeros(1, 24, device=dev)
t0 = time.time()
nd = max(200, nv // 4)
for i in range(nd):
    if i % 8 == 0:
        lat.step()                                   # (a training step, then straight into eval-mode decodes)
        vae.eval()
    with torch.no_grad():
        w, s_ = vae.decoder(z, dummy, train=False)
    if i % 50 == 0:
        z = torch.randn(1, vae.latent_space_dim, device=dev)
lat.trainer.finish()
torch.cuda.synchronize()
print(f"b = 1 decodes: {nd} calls interleaved with {nd // 8} LatentRNN steps in {time.time() - t0:.1f} s, chain_status {ops.chain_status()}", flush=True)
assert ops.chain_status() == 0 and ops.token_status() == 0
print("soak of the one-row kernels ok")
