#!/usr/bin/env python3
"""ConstraintModelGaussianReg.forward_inpaint latency (B = 1 and 32, 384 ticks, window of 4 measures = 96 ticks):
INET_ARNN_FREE_RUN=batched|loop python tools/arnn_inpaint_latency.py"""
import os, sys, time, types
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from inpaintnet_amd import synthetic
from inpaintnet_amd.arnn import ConstraintModelGaussianReg
sys.stdout = sys.stderr
ds = synthetic.SyntheticFolkDataset(num_notes=bench.NUM_NOTES)
ds.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]
model = ConstraintModelGaussianReg(ds, note_embedding_dim=10, metadata_embedding_dim=2, num_lstm_constraints_units=256,
                                   num_lstm_generation_units=256, linear_hidden_size=256, num_layers=2, dropout_input_prob=0.2,
                                   dropout_prob=0.2, unary_constraint=True, teacher_forcing=True)
model.eval()
for B in (1, 32):
    score = torch.from_numpy(synthetic.folk_score(B, bench.NUM_NOTES, seed=21)).long().cuda()
    md = torch.from_numpy(synthetic.folk_metadata(B)).long().cuda()
    a, b = 7 * 24, 11 * 24
    loc = torch.zeros_like(score); loc[:, :, :a] = 1; loc[:, :, b:] = 1
    with torch.no_grad():
        for _ in range(2): model.forward_inpaint(score, md, loc, a, b)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): model.forward_inpaint(score, md, loc, a, b)
        torch.cuda.synchronize()
    print(f"{os.environ.get('INET_ARNN_FREE_RUN', 'batched')} B={B}: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per forward_inpaint call")
