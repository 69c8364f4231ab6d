#!/usr/bin/env python3
"""The token pass of AnticipationRNN's free-running step alone (ops.arnn_generate, L = 384): python tools/arnn_token_pass.py"""
import os, sys, time, types
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from inpaintnet_amd import ops, synthetic
from inpaintnet_amd.arnn import ConstraintModelGaussianReg
sys.stdout = sys.stderr
ds = synthetic.SyntheticFolkDataset(num_notes=bench.NUM_NOTES)
ds.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]
m = ConstraintModelGaussianReg(ds, note_embedding_dim=10, metadata_embedding_dim=2, num_lstm_constraints_units=256,
                               num_lstm_generation_units=256, linear_hidden_size=256, num_layers=2, dropout_input_prob=0.2,
                               dropout_prob=0.2, unary_constraint=True, teacher_forcing=True)
pr = m.param
oc0 = torch.randn(384, 256, device="cuda") * 0.1
args = (pr("note_embeddings.0.weight"), oc0, pr("lstm_generation.0.weight_ih_l0"), pr("lstm_generation.0.bias_ih_l0"),
        pr("lstm_generation.0.weight_hh_l0"), pr("lstm_generation.0.bias_hh_l0"), pr("lstm_generation.1.weight_ih_l0"),
        pr("lstm_generation.1.bias_ih_l0"), pr("lstm_generation.1.weight_hh_l0"), pr("lstm_generation.1.bias_hh_l0"),
        pr("linear_1.weight"), pr("linear_1.bias"), pr("linear_ouput_notes.0.weight"), pr("linear_ouput_notes.0.bias"))
for _ in range(3): t = ops.arnn_generate(*args)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): t = ops.arnn_generate(*args)
torch.cuda.synchronize()
print(f"token pass, 384 ticks: {1e3 * (time.perf_counter() - t0) / 10:.2f} ms  ({1e3 * (time.perf_counter() - t0) / 10 / 384 * 1e3:.1f} us per tick)  tokens {t[:8].tolist()}")
