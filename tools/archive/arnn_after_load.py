#!/usr/bin/env python3
"""AnticipationRNN step time right after a heavy MFMA load in the same process (the bench's situation) and how long the chip
takes to come back to the latency it has when the workload runs alone: python tools/arnn_after_load.py"""
import os, sys, time, types
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from inpaintnet_amd import synthetic
from inpaintnet_amd.arnn import AnticipationRNNGaussianRegTrainer, ConstraintModelGaussianReg, free_positions
sys.stdout = sys.stderr
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
ds = synthetic.SyntheticFolkDataset(num_notes=bench.NUM_NOTES)
ds.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]
model = ConstraintModelGaussianReg(ds, note_embedding_dim=10, metadata_embedding_dim=2, num_lstm_constraints_units=256,
                                   num_lstm_generation_units=256, linear_hidden_size=256, num_layers=2, dropout_input_prob=0.2,
                                   dropout_prob=0.2, unary_constraint=True, teacher_forcing=True)
tr = AnticipationRNNGaussianRegTrainer(ds, model, lr=1e-4)
tr.overlap_backward = True
model.train()
data = tr.process_batch_data((torch.from_numpy(synthetic.folk_score(32, bench.NUM_NOTES, seed=21)), torch.from_numpy(synthetic.folk_metadata(32))))
def step():
    tr.zero_grad()
    w, _ = model(data[0], data[1], data[2], data[3], data[4], train=True, teacher_forcing=True)
    free = free_positions(data[2])
    loss, acc = tr.mean_crossentropy_loss_and_accuracy_voices(w, data[0][:, :, free].transpose(0, 1))
    loss.backward()
    tr.step()
def arnn(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for _ in range(5): step()
print("cold:", " ".join(f"{arnn(20):.2f}" for _ in range(4)))
for secs in (1.0, 4.0):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(50): wl.step()
        torch.cuda.synchronize()
    print(f"after {secs:.0f} s of VAE steps:", " ".join(f"{arnn(20):.2f}" for _ in range(10)))

# the bench's order: LatentRNN workloads (timed, then their per-kernel table) come first
for what in ("kernel_table(vae)", "latent steps", "latent table", "latent auto_reg"):
    if what == "kernel_table(vae)":
        bench.kernel_table(wl.step, nprof=2)
    elif what == "latent steps":
        lw = bench.LatentWorkload(torch.device("cuda", 0), 0, vae=wl.model, ds=wl.ds)
        for _ in range(24): lw.step()
        torch.cuda.synchronize()
    elif what == "latent table":
        bench.secondary_table(lw.step)
        del lw
    else:
        la = bench.LatentWorkload(torch.device("cuda", 0), 0, vae=wl.model, ds=wl.ds, auto_reg=True)
        for _ in range(24): la.step()
        torch.cuda.synchronize()
        del la
        wl.model.trainable = True
        wl.model.train()
    print(f"after {what}:", " ".join(f"{arnn(20):.2f}" for _ in range(4)))

# a model / trainer / batch created NOW, after everything above (the bench creates its AnticipationRNN this late)
r = bench.arnn_extra(steps=30, warmup=4, tables=False)["anticipation_rnn_train"]["ms_per_step"]
print("bench.arnn_extra() here:", r)
print("the early model again:", " ".join(f"{arnn(20):.2f}" for _ in range(3)))
