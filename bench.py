#!/usr/bin/env python3
"""Headline benchmark: measures/sec of full MeasureVAE training steps on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic FolkDB-shaped
tokens: zero_grad -> encoder fwd -> reparameterise -> hierarchical decoder fwd
(teacher-forcing coin per step, dropout 0.5) -> CE + KL + accuracy -> backward
-> [all-reduce of the flat gradient arena over RCCL when N > 1] -> fused Adam.
Workload at every N: BASELINE.json configs[1], batch = 256 measures PER GPU
(weak scaling), V=48, reference default hyper-parameters, random-init weights.

Prints ONE JSON line on rank 0 (contract in the task brief) carrying
  roofline     : the dominant MFMA kernel class, achieved TFLOP/s = algorithmic
                 FLOPs per launch / average launch duration measured with HIP
                 events on the launch stream (separate, un-timed steps), vs the
                 157.3 TFLOP/s dense fp32 MFMA peak of gfx950
  cpu_baseline : the oracle's port of the same step on the host cores
                 (oracle/torch_ref.CpuVaeTrainStep), bounded sample.
"""
import argparse
import json
import os
import random
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # kernel arguments in device memory (-0.5 ms/step)

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk
BATCH_PER_GPU = 256
NUM_NOTES = 48


def cpu_baseline(batch, max_seconds=25.0):
    """Reported baseline only: the oracle's CPU port of the same training step."""
    from oracle import torch_ref as O
    from inpaintnet_amd import layout, synthetic
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    shapes = layout.vae_param_shapes(NUM_NOTES)
    P = {k: torch.from_numpy(synthetic.det_param(k, s)) for k, s in shapes.items()}
    stepper = O.CpuVaeTrainStep(P, dropout=0.5)
    tok = torch.from_numpy(synthetic.det_tokens("bench/cpu", (batch, 24), NUM_NOTES))
    rng = random.Random(0)
    stepper.step(tok, True)                         # warm-up
    n, t0 = 0, time.time()
    while True:
        stepper.step(tok, rng.random() < 0.5)
        n += 1
        if time.time() - t0 > max_seconds or n >= 8:
            break
    dt = time.time() - t0
    return {"value": round(batch * n / dt, 2), "unit": "measures/s", "cores": threads, "kind": "port",
            "sample": f"{n} training steps of batch {batch} (oracle/torch_ref.CpuVaeTrainStep, fused aten::gru, "
                      f"{threads} threads of {cores} host cores)"}


def latent_rnn_extra(ds, vae, dev, batch=128, steps=10, warmup=3):
    """Secondary number (BASELINE.json configs[2]): LatentRNN training with the frozen MeasureVAE, 16-measure
    sequences, past/target/future = 6/4/6, batch 128 sequences, one GPU.  Not the headline metric."""
    from inpaintnet_amd import synthetic
    from inpaintnet_amd.latent_rnn import LatentRNN
    from inpaintnet_amd.latent_rnn_trainer import LatentRNNTrainer
    model = LatentRNN(ds, vae, num_rnn_layers=2, rnn_hidden_size=512, dropout=0.5, rnn_class=torch.nn.GRU,
                      auto_reg=False, teacher_forcing=True)
    trainer = LatentRNNTrainer(ds, model, lr=1e-4)
    trainer.overlap_backward = True              # as in the epoch loop (Trainer.loss_and_acc_on_epoch)
    model.train()
    score = torch.from_numpy(synthetic.folk_score(batch, NUM_NOTES, seed=9))
    past, future, target = LatentRNNTrainer.split_score(score, 6, 6, 4, 24)

    def step():
        trainer.zero_grad()
        loss, acc = trainer.loss_and_acc_for_batch((past, future, target), 0, train=True)
        loss.backward()
        trainer.step()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"latent_rnn_train": {"sequences_per_s": round(batch * steps / dt, 1),
                                 "measures_per_s": round(16 * batch * steps / dt, 1),
                                 "ms_per_step": round(1e3 * dt / steps, 3),
                                 "workload": "LatentRNN (non-AR) + frozen MeasureVAE, 128 sequences x 16 measures, "
                                             "past/target/future 6/4/6, dropout 0.5"}}


def decode_latency_extra(vae, iters=20):
    """Inference-side number for the fused tick decode (SURVEY 8d 'K7'): full HierarchicalDecoder.forward in eval
    mode (beat GRU + 24 ticks x [layer-0 step, layer-1 step, projection+argmax]) at small batch.  The HBM figure
    counts the 26.44 MB of decoder weights ONCE per call, as SURVEY.md section 8d prescribes."""
    out = {}
    vae.eval()
    for b in (1, 16, 256):
        z = torch.randn(b, vae.latent_space_dim, device=vae.flat.device)
        dummy = torch.zeros(b, 24, device=z.device)
        with torch.no_grad():
            for _ in range(3):
                vae.decoder(z, dummy, train=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                vae.decoder(z, dummy, train=False)
            torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / iters
        out[f"b{b}"] = {"ms_per_call": round(ms, 4), "measures_per_s": round(b / ms * 1e3, 1),
                        "weights_once_GBps": round(26.44e6 / (ms * 1e-3) / 1e9, 2)}
    vae.train()
    return {"decoder_eval": out}


def arnn_extra(batch=32, steps=8, warmup=2):
    """Secondary number (BASELINE.json configs[4]): AnticipationRNN gauss-reg model, teacher-forced training step,
    batch 32 sequences of 384 ticks, script defaults of train_arnn_reg.py.  Not the headline metric."""
    import types
    from inpaintnet_amd import synthetic
    from inpaintnet_amd.arnn import AnticipationRNNGaussianRegTrainer, ConstraintModelGaussianReg
    ds = synthetic.SyntheticFolkDataset(num_notes=NUM_NOTES)
    ds.metadatas = [types.SimpleNamespace(num_values=6), types.SimpleNamespace(num_values=6)]
    model = ConstraintModelGaussianReg(ds, note_embedding_dim=10, metadata_embedding_dim=2,
                                       num_lstm_constraints_units=256, num_lstm_generation_units=256,
                                       linear_hidden_size=256, num_layers=2, dropout_input_prob=0.2, dropout_prob=0.2,
                                       unary_constraint=True, teacher_forcing=True)
    trainer = AnticipationRNNGaussianRegTrainer(ds, model, lr=1e-4)
    trainer.overlap_backward = True
    model.train()
    score = torch.from_numpy(synthetic.folk_score(batch, NUM_NOTES, seed=21))
    md = torch.from_numpy(synthetic.folk_metadata(batch))
    torch.manual_seed(5)
    data = trainer.process_batch_data((score, md))

    def step():
        trainer.zero_grad()
        weights, _ = model(data[0], data[1], data[2], data[3], data[4], train=True, teacher_forcing=True)
        free = (data[2][0, 0, :] == 0).nonzero().squeeze(-1)
        loss, acc = trainer.mean_crossentropy_loss_and_accuracy_voices(weights, data[0][:, :, free].transpose(0, 1))
        loss.backward()
        trainer.step()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"anticipation_rnn_train": {"sequences_per_s": round(batch * steps / dt, 1),
                                       "measures_per_s": round(16 * batch * steps / dt, 1),
                                       "ms_per_step": round(1e3 * dt / steps, 3),
                                       "workload": "AnticipationRNN gauss-reg (LSTM 2x2 layers, H=256), teacher-forced "
                                                   "train step, 32 sequences x 384 ticks"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    from inpaintnet_amd import ops, synthetic
    from inpaintnet_amd.measure_vae import MeasureVAE, set_dropout_seed
    from inpaintnet_amd.vae_trainer import VAETrainer

    ds = synthetic.SyntheticFolkDataset(num_notes=NUM_NOTES)
    model = MeasureVAE(ds)                                     # reference defaults: E=10,H=512,Z=256, dropout 0.5
    sd = {k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape))) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)                                  # identical initial weights on every rank
    trainer = VAETrainer(ds, model, lr=1e-4)
    trainer.overlap_backward = True              # as in the epoch loop (Trainer.loss_and_acc_on_epoch)
    model.train()
    set_dropout_seed(1234, rank)                               # per-rank dropout / eps streams
    torch.manual_seed(1000 + rank)
    random.seed(4321)                                          # the teacher-forcing coin is shared by all ranks
    tokens = torch.from_numpy(synthetic.det_tokens(f"bench/rank{rank}", (BATCH_PER_GPU, 24), NUM_NOTES)).to(dev)

    def one_step():
        trainer.zero_grad()
        loss, acc = trainer.loss_and_acc_for_batch(tokens, 0, train=True)
        loss.backward()
        trainer.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = one_step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    final_loss = float(loss.detach())

    roof = None
    if not args.no_roofline and rank == 0:
        # separate, un-timed steps with every MFMA-kernel launch bracketed by HIP events on its stream; the
        # side-stream overlap is switched off for them so that each duration is the kernel alone
        from inpaintnet_amd import _lib
        _lib.lib().inet_set_option(0, 0)
        one_step()
        torch.cuda.synchronize()
        ops.prof_enable(True)
        nprof = 4
        for _ in range(nprof):
            one_step()
        torch.cuda.synchronize()
        stats = ops.prof_read()
        ops.prof_enable(False)
        _lib.lib().inet_set_option(0, 1)
        name = max(stats, key=lambda k: stats[k][1])
        n, ms, fl = stats[name]
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                "launches_per_step": n // nprof, "avg_launch_us": round(1e3 * ms / max(n, 1), 3),
                "classes": {k: {"launches_per_step": v[0] // nprof, "ms_per_step": round(v[1] / nprof, 4),
                                "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 3) if v[1] > 0 else 0.0}
                            for k, v in stats.items()}}
    if world > 1:
        torch.distributed.barrier()

    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = latent_rnn_extra(ds, model, dev)
        extras.update(arnn_extra())
        extras.update(decode_latency_extra(model))
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(BATCH_PER_GPU)
        out = {
            "metric": "measures/sec training (MeasureVAE: fwd + CE/KL + bwd + Adam)",
            "value": round(world * BATCH_PER_GPU * args.steps / dt, 2),
            "unit": "measures/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "MeasureVAE training, synthetic FolkDB-shaped tokens, 256 measures per GPU "
                                   "(BASELINE.json configs[1]); V=48,E=10,H=512,Z=256, dropout 0.5, "
                                   "teacher-forcing coin per step, Adam lr=1e-4",
                       "batch_per_gpu": BATCH_PER_GPU, "global_batch": world * BATCH_PER_GPU,
                       "parallelism": f"dp{world}", "final_loss": round(final_loss, 5)},
            "roofline": roof,
            "cpu_baseline": cpu,
            "extras": extras,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
