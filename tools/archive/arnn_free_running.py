#!/usr/bin/env python3
"""AnticipationRNN training step with the teacher-forcing coin on the other side (the reference draws it per batch, p = 0.5:
anticipation_rnn_gauss_reg_model.py:426-431): the free-running forward is a per-tick loop, here as in the reference.
python tools/arnn_free_running.py [steps]"""
import os, sys, time, types
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from inpaintnet_amd import synthetic
from inpaintnet_amd.arnn import AnticipationRNNGaussianRegTrainer, ConstraintModelGaussianReg, free_positions
sys.stdout = sys.stderr
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ds = synthetic.SyntheticFolkDataset(num_notes=bench.NUM_NOTES)
ds.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]
model = ConstraintModelGaussianReg(ds, note_embedding_dim=10, metadata_embedding_dim=2, num_lstm_constraints_units=256,
                                   num_lstm_generation_units=256, linear_hidden_size=256, num_layers=2, dropout_input_prob=0.2,
                                   dropout_prob=0.2, unary_constraint=True, teacher_forcing=True)
tr = AnticipationRNNGaussianRegTrainer(ds, model, lr=1e-4)
tr.overlap_backward = True
model.train()
data = tr.process_batch_data((torch.from_numpy(synthetic.folk_score(32, bench.NUM_NOTES, seed=21)), torch.from_numpy(synthetic.folk_metadata(32))))
for tf in (True, False):
    def step():
        tr.zero_grad()
        w, _ = model(data[0], data[1], data[2], data[3], data[4], train=True, teacher_forcing=tf)
        free = free_positions(data[2])
        loss, acc = tr.mean_crossentropy_loss_and_accuracy_voices(w, data[0][:, :, free].transpose(0, 1))
        loss.backward()
        tr.step()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    print(f"teacher_forcing={tf}: {1e3 * (time.perf_counter() - t0) / steps:.2f} ms per step")
