#!/usr/bin/env python3
"""Phase anatomy of the fused GRU step kernels inside ONE real MeasureVAE training step.

Needs a trace build of the library (not the product build):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DINET_STEP_TRACE -o build/libinet_trace.so inpaintnet_amd/csrc/*.hip
    INET_LIB_PATH=build/libinet_trace.so python tools/trace_steps.py
Thread 0 of every workgroup stamps the 100 MHz wall clock at entry (t0), after its wave's contraction (t1), after the
cross-wave reduce (t2) and after the epilogue's stores are issued (t3).  Launches are recovered by clustering on kind/grid.
"""
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inpaintnet_amd import _lib, synthetic  # noqa: E402
from inpaintnet_amd.measure_vae import MeasureVAE  # noqa: E402
from inpaintnet_amd.vae_trainer import VAETrainer  # noqa: E402

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


L = _lib.lib()
L.inet_set_option(0, int(os.environ.get("SIDE", "0")))
for _ in range(3):
    step(True)
torch.cuda.synchronize()
buf = torch.zeros(8 + 8 * 400000, dtype=torch.int64, device="cuda")
fn = L.inet_debug_trace_set
fn.restype = C.c_int
fn.argtypes = [C.c_void_p]
assert fn(C.c_void_p(buf.data_ptr())) == 0
step(True)
torch.cuda.synchronize()
n = int(buf[0].item())
rec = buf[8:8 + 8 * n].cpu().numpy().reshape(n, 8).astype(np.uint64)
t = rec[:, :4].astype(np.int64)
meta = rec[:, 4]
bx = (meta & np.uint64(0xffff)).astype(int)
by = ((meta >> np.uint64(16)) & np.uint64(0xffff)).astype(int)
kind = ((meta >> np.uint64(32)) & np.uint64(0xff)).astype(int)
ms = ((meta >> np.uint64(40)) & np.uint64(0xff)).astype(int)
gy = ((meta >> np.uint64(48)) & np.uint64(0xffff)).astype(int)
# launches: sort by t0 and split where (kind, ms, gy) changes or t0 jumps past the previous launch's end
order = np.argsort(t[:, 0], kind="stable")
launches = []
cur = []
cur_key = None
cur_end = 0
for i in order:
    key = (kind[i], ms[i], gy[i])
    if cur and (key != cur_key or t[i, 0] > cur_end):
        launches.append((cur_key, cur))
        cur = []
    if not cur:
        cur_key = key
        cur_end = 0
    cur.append(i)
    cur_end = max(cur_end, t[i, 3])
if cur:
    launches.append((cur_key, cur))
names = {0: "fwd", 1: "fwd+x", 2: "bwd"}
agg = collections.defaultdict(list)
for key, idx in launches:
    idx = np.array(idx)
    s0 = t[idx, 0].min()
    agg[key].append(dict(
        wgs=len(idx),
        spread=(t[idx, 0].max() - s0) / 100.0,
        span=(t[idx, 3].max() - s0) / 100.0,
        contr=np.mean(t[idx, 1] - t[idx, 0]) / 100.0,
        red=np.mean(t[idx, 2] - t[idx, 1]) / 100.0,
        epi=np.mean(t[idx, 3] - t[idx, 2]) / 100.0,
        wg=np.mean(t[idx, 3] - t[idx, 0]) / 100.0,
        wgmax=np.max(t[idx, 3] - t[idx, 0]) / 100.0,
        mhz=float(np.median(rec[idx, 7].astype(np.float64) / np.maximum(1, (t[idx, 3] - t[idx, 0])) * 100.0)),
        ccyc=float(np.median(rec[idx, 6].astype(np.float64))),
        xa=float(np.median((rec[idx, 5] & np.uint64(0xfffff)).astype(np.float64))),
        xb=float(np.median(((rec[idx, 5] >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.float64))),
        xc=float(np.median(((rec[idx, 5] >> np.uint64(40)) & np.uint64(0xfffff)).astype(np.float64))),
    ))
print(f"{n} workgroup records, {len(launches)} launches (one teacher-forced training step, side stream "
      f"{'on' if int(os.environ.get('SIDE', '0')) else 'off'})")
print(f"{'kernel':<8}{'MS':>3}{'gridY':>6}{'launches':>9}{'WGs':>6} | us: {'start spread':>12}{'contraction':>12}{'reduce':>8}"
      f"{'epilogue':>9}{'per-WG':>8}{'WG max':>8}{'first->last end':>16}{'clock MHz':>10}{'contr cyc':>10}{'  cyc: entry->loads':>20}{'->issued':>9}{'->hook end':>11}")
for key in sorted(agg):
    v = agg[key]
    m = lambda f: float(np.median([x[f] for x in v]))
    print(f"{names[key[0]]:<8}{key[1]:>3}{key[2]:>6}{len(v):>9}{int(m('wgs')):>6} |     {m('spread'):>12.2f}{m('contr'):>12.2f}"
          f"{m('red'):>8.2f}{m('epi'):>9.2f}{m('wg'):>8.2f}{m('wgmax'):>8.2f}{m('span'):>16.2f}{m('mhz'):>10.0f}{m('ccyc'):>10.0f}{m('xa'):>20.0f}{m('xb'):>9.0f}{m('xc'):>11.0f}")
