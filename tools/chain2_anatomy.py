"""Anatomy of a step of the dominant kernel (the second-generation forward chain, csrc/gru_chain2.hip) from in-kernel wall-clock
stamps: needs a build with -DINET_CHAIN2_STAMPS=1 (tools/build_variant.sh stamps -DINET_CHAIN2_STAMPS=1 with
VARIANT_SRCS=gru_chain2; run with INET_LIB_PATH=build/lib_stamps.so).  Wave 0 of the first workgroup (group 0, member 0) stamps six
points per step into the library's diagnostics area (inet_debug_read); the encoder's layer-1 launch is the last to write.
    python3 tools/chain2_anatomy.py"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from inpaintnet_amd import _lib, ops, synthetic
from inpaintnet_amd.measure_vae import MeasureVAE

dev = torch.device("cuda", 0)
ds = synthetic.SyntheticFolkDataset(num_notes=48)
vae = MeasureVAE(ds)
vae.train()
tok = torch.from_numpy(synthetic.det_tokens("anatomy", (256, 24), 48)).to(dev)
for _ in range(5):
    z = vae.encoder(tok)                      # grad mode: the training forward (saves, piece outputs)
torch.cuda.synchronize()
buf = (C.c_uint64 * (32 * 8))()
_lib.check(_lib.lib().inet_debug_read(C.cast(buf, C.c_void_p), C.sizeof(buf)), "inet_debug_read")
st = np.frombuffer(buf, dtype=np.uint64).reshape(32, 8).astype(np.float64) * 0.01      # 100 MHz wall clock -> us
if st[:24].max() == 0:
    print("no stamps: this library was not built with -DINET_CHAIN2_STAMPS=1")
    sys.exit(1)
T = 24
s = st[2:T - 1]                               # steady-state steps
nxt = st[3:T]
names = ["operand requests + WAIT for the row block's counter (the hand-off as the consumer sees it)",
         "contraction: A fragments from the exchange (sc1 loads, ring of 3 k-steps) + 16 k-steps x 27 MFMAs",
         "gates, cell, transpose tiles",
         "publish the new state (3 x 16-byte stores per lane) + drain (vmcnt 0) + counter add",
         "piece outputs, outputs and saves issued",
         "to the next step's first stamp"]
print(f"forward chain T24 B256 H512 (layer 1 of the encoder), wave 0 of workgroup (group 0, member 0): step period "
      f"{float((nxt[:, 0] - s[:, 0]).mean()):.2f} us over steps 2..{T - 2}; the launch {float(st[T - 1, 5] - st[0, 0]):.1f} us first to last stamp")
for i, n in enumerate(names):
    d = (s[:, i + 1] - s[:, i]) if i < 5 else (nxt[:, 0] - s[:, 5])
    print(f"  {float(d.mean()):6.2f} us (min {float(d.min()):5.2f}, max {float(d.max()):5.2f})  {n}")
