#!/usr/bin/env python3
"""Headline benchmark: measures/sec of full MeasureVAE training steps on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload vae|latent]

One "step" = one pass of the hot path over one batch of synthetic FolkDB-shaped tokens:
  vae    (headline, BASELINE.json configs[1]): zero_grad -> encoder fwd -> reparameterise -> hierarchical decoder
         fwd (teacher-forcing coin per step, dropout 0.5) -> CE + KL + accuracy -> backward -> [all-reduce of the flat
         gradient arena over RCCL when N > 1] -> fused Adam.  256 measures PER GPU (weak scaling).
  latent (BASELINE.json configs[2] on one GPU, configs[3] on N): frozen MeasureVAE encode of 16 measures -> context
         bi-GRUs -> generator bi-GRU -> Linear -> frozen decode -> CE -> backward -> all-reduce of the 159.6 MB
         LatentRNN arena -> Adam.  128 sequences PER GPU (1024 global at N = 8).

Launching: with --gpus N > 1 and no WORLD_SIZE in the environment this process only SPAWNS
`python -m torch.distributed.run --nproc-per-node N bench.py ...` (before anything touches the GPU) and exits with its
code; under torch.distributed.run it is one rank.  --gpus must equal the world size RCCL reports, and N may not exceed
the visible devices: a mismatch is an error, never a silent 1-GPU number.

Prints ONE JSON line (< 4 KB) on rank 0 (contract in the task brief) carrying
  roofline       the kernel with the largest summed time in the step (per-launch HIP-event timing on the launch
                 stream during separate un-timed steps), its algorithmic FLOPs and bytes per launch against the MFMA roof of
                 the pipe it runs on (157.3 TFLOP/s f32-input; kernels on the bf16 pipe through exact three-piece splits:
                 2500 / 9 = 277.8 TFLOP/s of f32 products) and the 8 TB/s HBM peak of gfx950, `traffic` = HBM bytes per launch from the
                 committed rocprofv3 PMC pass (profiles/), and the same table for the top kernels;
  cpu_baseline   the oracle's CPU port of the same step on the host cores (bounded sample, best thread count);
  parity_checked one un-timed step at the bench's own batch compared with the oracle (loss, logits, every gradient).
plus `chain_timeouts`, `slow_waits` (the library's recorder of waits > ~50 us inside persistent kernels, timed region only), the
timed steps one by one (`first_steps_ms`) and one scalar per secondary workload.  The line is HARD-LIMITED to 4 KB (compact_line;
round 5's 20 KB line was dropped by the driver): kernel tables, per-step host times, parity detail and workload prose go to
bench_detail.json beside this file and to stderr.  --warmup W means W: nothing runs in front of it (round 5 primed both branches of
the teacher-forcing coin; the one-time host work of a kernel's first launch is now done by inet_preload() at trainer construction).
"""
import argparse
import collections
import csv
import gc
import json
import os
import random
import socket
import subprocess
import sys
import tempfile
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")      # kernel arguments in device memory (-0.5 ms/step)

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 FLOP/clk
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 (16x the f32-input rate)
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s HBM3E (6.3 TB/s achievable)
NUM_NOTES = 48
VAE_BATCH_PER_GPU = 256
LATENT_SEQ_PER_GPU = 128
PMC_FILE = os.path.join(REPO, "profiles", "r06_pmc_traffic.json")


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """Parent of a multi-GPU run: never initialises the GPU (device_count() does not, on this image)."""
    have = torch.cuda.device_count()
    if args.gpus > have and os.environ.get("INET_BENCH_SHARE_GPU") != "1":
        print(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible; refusing to report a "
              f"{have}-GPU number as a {args.gpus}-GPU result", file=sys.stderr)
        return 2
    cmd, env = rank_command(args.gpus, argv)
    return subprocess.call(cmd, env=env)


def rank_command(gpus, argv, environ=None):
    """(command, environment) of the rank launcher: one process per GPU over RCCL, rendezvous on 127.0.0.1 (the container's
    hostname may not resolve), dmabuf IPC (the host driver has no legacy IPC: RCCL's hipIpcGetMemHandle fails without it)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ if environ is None else environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return cmd, env


# ------------------------------------------------------------------------------------------------ CPU legs (un-timed)
def _time_steps(fn, max_seconds, max_steps):
    fn()                                            # warm-up
    n, t0 = 0, time.time()
    while True:
        fn()
        n += 1
        if time.time() - t0 > max_seconds or n >= max_steps:
            break
    return n, time.time() - t0


def cpu_baseline(batch, seconds=20.0):
    """Reported baseline only: the oracle's CPU port of the same training step, at the best of a few thread counts
    (64 threads oversubscribe aten::gru at this size: VERDICT r01 weak 7)."""
    from oracle import torch_ref as O
    from inpaintnet_amd import layout, synthetic
    cores = os.cpu_count() or 1
    shapes = layout.vae_param_shapes(NUM_NOTES)
    P = {k: torch.from_numpy(synthetic.det_param(k, s)) for k, s in shapes.items()}
    tok = torch.from_numpy(synthetic.det_tokens("bench/cpu", (batch, 24), NUM_NOTES))
    cands = sorted({t for t in (8, 16, 32, 64) if t <= cores} | {min(cores, 8)})
    best = None
    tried = {}
    for th in cands:                                 # a short scan for the best thread count ...
        torch.set_num_threads(th)
        stepper = O.CpuVaeTrainStep(P, dropout=0.5)
        rng = random.Random(0)
        n, dt = _time_steps(lambda: stepper.step(tok, rng.random() < 0.5), seconds / (2 * len(cands)), 2)
        tried[th] = round(batch * n / dt, 2)
        if best is None or tried[th] > best[0]:
            best = (tried[th], th, n)
    torch.set_num_threads(best[1])                   # ... then the sample that is reported: at least 8 timed steps there
    stepper = O.CpuVaeTrainStep(P, dropout=0.5)
    rng = random.Random(0)
    n, dt = _time_steps(lambda: stepper.step(tok, rng.random() < 0.5), seconds / 2, 8)
    n2 = 0
    if n < 8:                                        # (a slow host: keep going until eight steps are in)
        n2, dt2 = _time_steps(lambda: stepper.step(tok, rng.random() < 0.5), 1e9, 8 - n)
        n, dt = n + n2, dt + dt2
    out = {"value": round(batch * n / dt, 2), "unit": "measures/s", "cores": best[1], "kind": "port",
           "sample": f"{n} timed training steps of batch {batch} at {best[1]} threads "
                     f"(oracle/torch_ref.CpuVaeTrainStep, fused aten::gru, dropout 0.5); thread count chosen by a scan of 2 steps "
                     f"each: {tried} measures/s on {cores} host cores"}
    # the other configurations BASELINE.md section 3 promises next to the MI355X numbers (bounded samples)
    legs = {}
    torch.set_num_threads(best[1])
    try:
        st = O.CpuVaeTrainStep(P, dropout=0.5)
        tok2 = tok[:2]
        n, dt = _time_steps(lambda: st.step(tok2, True), 2.0, 10)
        legs["cfg1_vae_b2"] = {"measures_per_s": round(2 * n / dt, 2), "ms_per_step": round(1e3 * dt / n, 2)}
        # cfg3 on a quarter batch (32 of the 128 sequences: the CPU step is linear in the batch and a full one takes
        # tens of seconds to warm up), cfg5 at its own batch of 32
        lat = O.CpuLatentTrainStep(NUM_NOTES, dropout=0.5)
        n, dt = _time_steps(lambda: lat.step(32), 6.0, 2)
        legs["cfg3_latent"] = {"sequences_per_s": round(32 * n / dt, 2), "measures_per_s": round(16 * 32 * n / dt, 2),
                               "ms_per_step": round(1e3 * dt / n, 1), "sample": f"{n} steps of 32 sequences x 16 measures"}
        ar = O.CpuArnnTrainStep(NUM_NOTES)
        n, dt = _time_steps(lambda: ar.step(32), 6.0, 2)
        legs["cfg5_arnn_b32"] = {"sequences_per_s": round(32 * n / dt, 2), "measures_per_s": round(16 * 32 * n / dt, 2),
                                 "ms_per_step": round(1e3 * dt / n, 1), "sample": f"{n} steps of 32 sequences x 384 ticks"}
    except Exception as e:                           # a baseline leg must never take the headline down with it
        legs["error"] = repr(e)
    out["other_configs"] = legs
    return out


def parity_check(model, tokens_dev):
    """One un-timed training step at the bench's own batch and on the weights the timed run just trained (both coin
    values), through the C-ABI, against the oracle on the same weights, tokens, eps and dropout masks.  The oracle is
    only the checker here: nothing it computes is timed or reported as throughput.  SELU / ReLU branches of
    pre-activations within fp32 noise of 0 are aligned with the GPU's (oracle `kinks`, tests/test_gpu_bench_sizes.py)."""
    from oracle import torch_ref as O
    from inpaintnet_amd import ops
    ops.side_defer(False)
    cfg, params = model.cfg, model.flat
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tok = tokens_dev.cpu()
    B, T = tok.shape
    V, nb, He, Hd, Z = cfg.num_notes, cfg.beats, cfg.enc_hidden, cfg.dec_hidden, cfg.z_dim
    dev = tokens_dev.device
    eps = torch.randn(B, Z, generator=torch.Generator().manual_seed(7))
    m_enc = ops.dropout_mask((T, B, 2 * He), 0.5, 4242, 0, dev)
    m_beat = ops.dropout_mask((nb, B, Hd), 0.5, 4242, 10 ** 8, dev)
    m_tick = ops.dropout_mask((T, B, Hd), 0.5, 4242, 2 * 10 ** 8, dev)
    om = {"enc": m_enc.cpu().permute(1, 0, 2), "beat": m_beat.cpu().permute(1, 0, 2), "tick": m_tick.cpu().permute(1, 0, 2)}
    grads = torch.zeros_like(params)
    worst = {}
    for tf in (True, False):
        tag = "tf" if tf else "fr"
        grads.zero_()
        mu, ls, ews = ops.encoder_fwd(cfg, tokens_dev, params, mask=m_enc, save=True)
        acc3 = torch.zeros(3, device=dev)
        z, _ = ops.reparam_kl(mu, ls, eps.to(dev), kl_sum=acc3[2:3])
        w, s, dws = ops.decoder_fwd(cfg, z, tokens_dev, tf, params, m_beat, m_tick, save=True)
        fld = lambda ws, which, name, shape: ops.ws_field(cfg, ws, B, which, name).view(shape).cpu() > 0
        kinks = {"a_mu": fld(ews, 0, "a_mu", (B, 2 * He)), "a_ls": fld(ews, 0, "a_ls", (B, 2 * He)),
                 "hb0": fld(dws, 1, "hb0", (B, 2 * Hd)), "ht0": fld(dws, 1, "ht0", (nb, B, 2 * Hd)),
                 "c_all": fld(dws, 1, "c_all", (nb, B, Hd)), "relu": w.cpu() > 0}
        dW = torch.empty_like(w)
        ops.cross_entropy(w.view(B * T, V), tokens_dev.reshape(-1), acc3, dW=dW.view(B * T, V), scale=1.0 / (B * T))
        dz = ops.decoder_bwd(cfg, dW, w, s, params, grads, m_beat, m_tick, dws)
        dmu, dls = ops.latent_bwd(dz, mu, ls, eps.to(dev), 1e-3 / B)
        ops.encoder_bwd(cfg, tokens_dev, params, grads, m_enc, dmu, dls, ews)
        a = acc3.cpu().double()
        loss = float(a[0] / (B * T) + 1e-3 * a[2] / B)
        Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        O.kink_stats_reset()
        wr, sr, mur, lsr, zr = O.vae_forward(Pr, tok, eps, tf, om, feed_tokens=None if tf else s.cpu()[:, 0], kinks=kinks)
        # the oracle followed the GPU's branch only where ITS pre-activation is within KINK_TOL of 0; a branch that
        # differs anywhere else is a `violation` and fails the check
        worst[f"kink_flips_{tag}"] = int(O.KINK_STATS["flips"])
        worst[f"kink_violations_{tag}"] = int(O.KINK_STATS["violations"])
        worst[f"kink_elements_{tag}"] = int(O.KINK_STATS["elements"])
        worst[f"kink_max_abs_{tag}"] = float(O.KINK_STATS["max_abs_flip"])
        lr, cer, klr, accr = O.vae_loss(wr, tok, mur, lsr)
        lr.backward()
        worst[f"loss_{tag}"] = abs(loss - lr.item()) / abs(lr.item())
        worst[f"logits_{tag}"] = float((w.cpu() - wr.detach()).abs().max() / wr.detach().abs().max())
        gerr, gname, gnext = 0.0, "", 0.0                       # (worst tensor, and the worst of all the others)
        for name, off, shape in model._table:
            gr = Pr[name].grad
            e = float((grads[off:off + gr.numel()].view(shape).cpu() - gr).abs().max() / (gr.abs().max() + 1e-12))
            if e > gerr:
                gerr, gname, gnext = e, name, gerr
            elif e > gnext:
                gnext = e
        worst[f"grads_{tag}"] = gerr
        worst[f"grads_{tag}_worst_tensor"] = gname
        worst[f"grads_{tag}_all_other_tensors"] = gnext
        # token indices: exact on every row whose top-2 margin exceeds 1e-4 of the row's own maximum (rows whose two best
        # post-ReLU logits are both 0 cannot be compared); the fraction actually compared is reported
        top2 = torch.topk(wr.detach(), 2, dim=-1).values
        ok = (top2[..., 0] - top2[..., 1]) > 1e-4 * top2[..., 0].clamp_min(1e-30)
        worst[f"token_mismatch_{tag}"] = int((s.cpu()[:, 0][ok] != sr[:, 0][ok]).sum())
        worst[f"token_rows_compared_{tag}"] = int(ok.sum())
        worst[f"token_rows_compared_frac_{tag}"] = float(ok.float().mean())
        # (on these weights -- synthetic initial values plus a few hundred optimizer steps -- most rows have NO positive logit: every
        #  post-ReLU logit is 0 and the token is the tie-break index 0 in both implementations; such rows cannot be told apart from
        #  a row whose best pre-activation sits within rounding of 0, so they are not counted as compared.  With trained-looking
        #  logits the tests compare 99.9 % of the rows: tests/test_gpu_bench_sizes.py)
        worst[f"token_rows_all_zero_{tag}"] = int((top2[..., 0] <= 0).sum())
    passed = all(worst[f"loss_{t}"] <= 1e-4 and worst[f"logits_{t}"] <= 1e-4 and worst[f"grads_{t}"] <= 5e-4 and
                 worst[f"token_mismatch_{t}"] == 0 and worst[f"kink_violations_{t}"] == 0 and
                 worst[f"kink_flips_{t}"] <= 8 + 1e-5 * worst[f"kink_elements_{t}"] for t in ("tf", "fr"))
    return {"parity_checked": bool(passed),
            "max_rel_err": round(max(v for k, v in worst.items() if isinstance(v, float) and not k.startswith(("kink_", "token_"))), 8),
            "kink_flips": worst["kink_flips_tf"] + worst["kink_flips_fr"],
            "kink_violations": worst["kink_violations_tf"] + worst["kink_violations_fr"],
            "parity_detail": {k: (round(v, 8) if isinstance(v, float) else v) for k, v in worst.items()},
            "parity_tolerance": "loss 1e-4 rel, logits 1e-4 of max, every gradient tensor 5e-4 of its max, sampled "
                                "tokens exact on rows with top-2 margin > 1e-4 of the row's maximum (north_star; "
                                "token_rows_compared_frac_* = the share of rows that was compared); SELU/ReLU branches "
                                "aligned with the GPU's only where the oracle's pre-activation is within 1e-5 of 0 "
                                "(kink_flips = how many; kink_violations = branch differences outside that band, must be 0)"}


# ------------------------------------------------------------------------------------------------ workloads
class VaeWorkload:
    name = "vae"
    units_per_step = VAE_BATCH_PER_GPU
    unit = "measures/s"

    def __init__(self, dev, rank):
        from inpaintnet_amd import synthetic
        from inpaintnet_amd.measure_vae import MeasureVAE
        from inpaintnet_amd.vae_trainer import VAETrainer
        self.ds = synthetic.SyntheticFolkDataset(num_notes=NUM_NOTES)
        self.model = MeasureVAE(self.ds)                       # reference defaults: E=10,H=512,Z=256, dropout 0.5
        sd = {k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape))) for k, v in self.model.state_dict().items()}
        self.model.load_state_dict(sd)                         # identical initial weights on every rank
        self.trainer = VAETrainer(self.ds, self.model, lr=1e-4)
        self.trainer.overlap_backward = True                   # as in the epoch loop (Trainer.loss_and_acc_on_epoch)
        self.model.train()
        self.tokens = torch.from_numpy(synthetic.det_tokens(f"bench/rank{rank}", (VAE_BATCH_PER_GPU, 24), NUM_NOTES)).to(dev)
        self.arena_mb = self.model.flat.numel() * 4 / 1e6

    def step(self):
        t = self.trainer
        t.zero_grad()
        loss, acc = t.loss_and_acc_for_batch(self.tokens, 0, train=True)
        loss.backward()
        t.step()
        return loss

    def describe(self, world):
        return {"workload": "MeasureVAE training, synthetic FolkDB-shaped tokens, 256 measures per GPU "
                            "(BASELINE.json configs[1]); V=48,E=10,H=512,Z=256, dropout 0.5, "
                            "teacher-forcing coin per step, Adam lr=1e-4",
                "batch_per_gpu": VAE_BATCH_PER_GPU, "global_batch": world * VAE_BATCH_PER_GPU,
                "parallelism": f"dp{world}"}


class LatentWorkload:
    name = "latent"
    units_per_step = 16 * LATENT_SEQ_PER_GPU
    unit = "measures/s"

    def __init__(self, dev, rank, vae=None, ds=None, auto_reg=False, encode_all=True):
        from inpaintnet_amd import synthetic
        from inpaintnet_amd.latent_rnn import LatentRNN
        from inpaintnet_amd.latent_rnn_trainer import LatentRNNTrainer
        from inpaintnet_amd.measure_vae import MeasureVAE
        self.ds = ds or synthetic.SyntheticFolkDataset(num_notes=NUM_NOTES)
        if vae is None:
            vae = MeasureVAE(self.ds)
            vae.load_state_dict({k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape)))
                                 for k, v in vae.state_dict().items()})
        self.model = LatentRNN(self.ds, vae, num_rnn_layers=2, rnn_hidden_size=512, dropout=0.5, rnn_class=torch.nn.GRU,
                               auto_reg=auto_reg, teacher_forcing=True)
        self.auto_reg = auto_reg
        # encode_all: the reference's work measure for measure -- it encodes the target measures in every forward pass and reads
        # the result only when the auto-regressive generator is teacher-forced (latent_rnn.py:133, 148-149).  The package drops
        # that dead encode by default (LatentRNN.encode_unused_target = False); SURVEY.md section 8(d) prices 16 encodes per
        # sequence, so the LatentRNN lines of this file time all 16 and `latent_rnn_train_default` the package's default.
        self.model.encode_unused_target = bool(encode_all)
        self.encode_all = bool(encode_all)
        own = {k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape))) for k, v in self.model.named_parameters()}
        for k, v in own.items():
            self.model.param(k).copy_(v)
        self.trainer = LatentRNNTrainer(self.ds, self.model, lr=1e-4)
        self.trainer.overlap_backward = True
        self.model.train()
        score = torch.from_numpy(synthetic.folk_score(LATENT_SEQ_PER_GPU, NUM_NOTES, seed=9 + rank))
        self.batch = LatentRNNTrainer.split_score(score, 6, 6, 4, 24)
        self.arena_mb = self.model.flat.numel() * 4 / 1e6

    def step(self):
        t = self.trainer
        t.zero_grad()
        loss, acc = t.loss_and_acc_for_batch(self.batch, 0, train=True)
        loss.backward()
        t.step()
        return loss

    def describe(self, world):
        if self.auto_reg:
            return {"workload": "LatentRNN with auto_reg=True (the script default, train_inpaintnet.py:53; teacher-forcing coin per "
                                "step: the generator runs measure by measure and, when free-running, decodes and re-encodes "
                                "each generated measure), frozen MeasureVAE, 128 sequences x 16 measures per GPU, 6/4/6",
                    "batch_per_gpu": LATENT_SEQ_PER_GPU, "global_batch": world * LATENT_SEQ_PER_GPU, "parallelism": f"dp{world}"}
        return {"workload": "LatentRNN (non-AR) training with the frozen MeasureVAE, 128 sequences x 16 measures per "
                            "GPU, past/target/future 6/4/6, dropout 0.5 (BASELINE.json configs[2]; configs[3] when "
                            "data-parallel: 1024 sequences global at 8 GPUs)",
                "batch_per_gpu": LATENT_SEQ_PER_GPU, "global_batch": world * LATENT_SEQ_PER_GPU,
                "parallelism": f"dp{world}"}


def timed(step, steps, warmup, fence, trace=None):
    """W un-timed steps, a fence, EXACTLY `steps` timed steps, a fence.  trace (a dict): also the first 32 timed steps one by one --
    host time to queue each and the GPU's time between events recorded behind consecutive steps -- and Python's garbage
    collections inside the region (a full collection over the import-time heap is a 40 ms host stall: profiles/r05_cold_start.txt)."""
    for _ in range(warmup):
        step()
    ntr = min(steps, 32) if trace is not None else 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(ntr + 1)]
    host, gcs, g0 = [], [], [0.0]

    def on_gc(phase, info):
        if phase == "start":
            g0[0] = time.perf_counter()
        else:
            gcs.append((info["generation"], round(1e3 * (time.perf_counter() - g0[0]), 3)))
    if trace is not None:
        gc.callbacks.append(on_gc)
    fence()
    t0 = time.perf_counter()
    if ntr:
        ev[0].record()
    for i in range(steps):
        if i < ntr:
            h0 = time.perf_counter()
            loss = step()
            ev[i + 1].record()
            host.append(time.perf_counter() - h0)
        else:
            loss = step()
    fence()
    dt = time.perf_counter() - t0
    if trace is not None:
        gc.callbacks.remove(on_gc)
        trace["first_steps_ms"] = [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(ntr)]
        trace["first_steps_host_ms"] = [round(1e3 * h, 3) for h in host]
        trace["gc_in_timed_region"] = {"collections": len(gcs), "generation2": sum(1 for g, _ in gcs if g == 2),
                                       "longest_ms": max([m for _, m in gcs], default=0.0)}
    return dt, loss


def exchange_report(wl, dp, step_s, fence, dev, steps=40):
    """What the data-parallel exchange of one step looks like and what it costs: the ranges reduced under the backward pass
    ("bucket") and in step() ("final": includes the step-flag word in front of the arena), in MB, and the exposed time = the
    step with the exchange minus the same step with every gradient collective switched off (dp.set_exchange), MAX over ranks."""
    wl.step()
    torch.cuda.synchronize()
    ranges = [{"kind": k, "mbytes": round(4 * (b - a) / 1e6, 3)} for k, a, b in dp.last_ranges]
    dp.set_exchange(False)
    try:
        dt_off, _ = timed(wl.step, steps, 5, fence)
    finally:
        dp.set_exchange(True)
    dp.broadcast_params(wl.model.flat)                         # (the ranks drifted apart while nothing was exchanged:
    for t in (wl.trainer.adam_m, wl.trainer.adam_v):           #  weights AND Adam moments, for whatever the ranks run next)
        dp.broadcast_params(t)
    t = torch.tensor([dt_off / steps], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return {"ranges": ranges, "ms_per_step_without_exchange": round(1e3 * float(t.item()), 4),
            "exposed_exchange_ms_per_step": round(1e3 * (step_s - float(t.item())), 4)}


def allreduce_ms(grad, iters=10):
    """Cost of the gradient exchange alone: all-reduce(sum) of the flat arena over RCCL, averaged."""
    if not torch.distributed.is_initialized():
        return None
    buf = torch.zeros_like(grad)
    for _ in range(2):
        torch.distributed.all_reduce(buf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        torch.distributed.all_reduce(buf)
    torch.cuda.synchronize()
    return round(1e3 * (time.perf_counter() - t0) / iters, 4)


# ------------------------------------------------------------------------------------------------ roofline
def kernel_table(step, nprof=4):
    """Per-launch HIP-event timing (inet_prof_*) of separate, un-timed steps with the side stream off, so that every
    duration is the kernel alone; grouped by the label that names the template instantiation and shape."""
    from inpaintnet_amd import ops
    ops.set_option(0, 0)
    step()
    torch.cuda.synchronize()
    ops.prof_enable(True)
    for _ in range(nprof):
        step()
    torch.cuda.synchronize()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "launches.csv")
        ops.prof_dump(path)
        rows = list(csv.DictReader(open(path)))
    ops.prof_enable(False)
    ops.set_option(0, 1)
    groups = {}
    for r in rows:
        g = groups.setdefault(r["label"] or f"class{r['class']}", {"n": 0, "us": 0.0, "gflop": 0.0, "mb": 0.0})
        g["n"] += 1
        g["us"] += float(r["us"])
        g["gflop"] += float(r["gflop"])
        g["mb"] += float(r["mbytes"])
    table = []
    for label, g in groups.items():
        us, n = g["us"], g["n"]
        tflops = g["gflop"] * 1e9 / (us * 1e-6) / 1e12 if us > 0 else 0.0
        gbps = g["mb"] * 1e6 / (us * 1e-6) / 1e9 if us > 0 else 0.0
        row = {"kernel": label, "launches_per_step": round(n / nprof, 2), "avg_us": round(us / n, 3),
               "ms_per_step": round(us / nprof / 1e3, 4), "gflop_per_launch": round(g["gflop"] / n, 4),
               "mbytes_per_launch": round(g["mb"] / n, 4), "tflops": round(tflops, 2), "gbps": round(gbps, 1),
               "frac_hbm": round(gbps / PEAK_HBM_GBPS, 4)}
        npieces = piece_products(label)
        if npieces:
            # the kernel runs its products on the bf16 matrix cores, `npieces` bf16 MFMA products per f32 product: its MFMA
            # roof in ALGORITHMIC (f32-product) FLOP/s is the bf16 dense peak / npieces
            row["mfma_pipe"] = f"bf16 x{npieces}"
            row["peak_tflops"] = round(PEAK_BF16_MFMA_TFLOPS / npieces, 1)
            row["executed_bf16_tflops"] = round(tflops * npieces, 1)
            row["frac_of_f32_input_peak"] = round(tflops / PEAK_F32_MFMA_TFLOPS, 4)
        else:
            row["mfma_pipe"] = "f32"
            row["peak_tflops"] = PEAK_F32_MFMA_TFLOPS
        row["frac_mfma"] = round(tflops / row["peak_tflops"], 4)
        table.append(row)
    table.sort(key=lambda r: -r["ms_per_step"])
    return table


def secondary_table(step, top=8, ms_per_step=None):
    """The per-kernel roofline table of a secondary workload (LatentRNN, AnticipationRNN): the same rows as roofline.kernels,
    without PMC traffic (the committed PMC passes profile the headline step).  ms_per_step: the un-profiled step the table must
    be consistent with -- a table whose kernels sum to more than twice that step is not evidence (BENCH_r05: lstm_chain_bwd read
    470 us per launch under per-launch events on the driver's box, 110 everywhere else) and is reported as "inconsistent" with
    what the library's slow-wait recorder saw during the profiled steps."""
    from inpaintnet_amd import ops
    keep = ("kernel", "launches_per_step", "avg_us", "ms_per_step", "tflops", "mfma_pipe", "peak_tflops", "frac_mfma", "gbps", "frac_hbm")
    ops.slow_waits(reset=True)
    table = kernel_table(step, nprof=3)
    slow = ops.slow_waits(reset=True)
    out = {"step_kernel_ms": round(sum(r["ms_per_step"] for r in table), 4),
           "step_gflop": round(sum(r["gflop_per_launch"] * r["launches_per_step"] for r in table), 2),
           "launches_per_step": round(sum(r["launches_per_step"] for r in table), 1),
           "slow_waits_while_profiled": slow["count"], "waits_noted_while_profiled": slow["noted"]}
    if ms_per_step is not None and out["step_kernel_ms"] > 2.0 * ms_per_step:
        out["kernels"] = "inconsistent"
        out["unprofiled_ms_per_step"] = ms_per_step
        out["slow_wait_entries"] = slow["entries"][:8]
        out["largest"] = {k: table[0][k] for k in ("kernel", "launches_per_step", "avg_us", "ms_per_step")} if table else None
        return out
    out["top"] = [{k: r[k] for k in keep if k in r} for r in table[:top]]
    return out


def piece_products(label):
    """bf16 piece products per f32 product of a kernel label (csrc/gemm_bf3.hip 'bf3p9', csrc/gru_chain2.hip 'v2w4 p9'), 0 for
    the f32-input MFMA kernels."""
    import re
    m = re.search(r"\bbf3p(\d)\b", label) or re.search(r" v2w\d+e? p(\d) ", label) or re.search(r"^gru_step_bf3(?:_bwd)? p(\d) ", label)
    return int(m.group(1)) if m else 0


def dump_kernel_sequences(step, path, nsteps=6):
    """For tools/pmc_summary.py: one template instantiation + grid can serve several shapes (the direct weight-gradient
    kernel runs M1536 N512 and M1536 N1024 with the same 256 workgroups).  The per-step launch ORDER of the labels behind
    each kernel|grid key lets the PMC summary split such a key by shape.  Only keys whose order is the same in every
    profiled step are written."""
    from inpaintnet_amd import ops
    seqs = []
    for _ in range(nsteps):
        torch.cuda.synchronize()
        ops.prof_enable(True)
        step()
        torch.cuda.synchronize()
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "l.csv")
            ops.prof_dump(f)
            rows = list(csv.DictReader(open(f)))
        ops.prof_enable(False)
        per = {}
        for r in rows:
            key = pmc_key(r["label"]) if r["label"] else None
            if key:
                per.setdefault(key, []).append(r["label"])
        seqs.append(per)
    out = {}
    for key in seqs[0]:
        if all(s_.get(key) == seqs[0][key] for s_ in seqs) and len(set(seqs[0][key])) > 1:
            out[key] = seqs[0][key]
    # every label of the profiled steps that maps to a PMC key: tools/pmc_summary.py copies the list (and its hash) into the
    # traffic file, and roofline() refuses the file's figures when the library launches a label the file has never seen
    out["__labels__"] = sorted({l for s_ in seqs for ls in s_.values() for l in ls})
    json.dump(out, open(path, "w"), indent=1)
    return out


def label_hash(labels):
    import hashlib
    return hashlib.sha1("\n".join(sorted(set(labels))).encode()).hexdigest()[:16]


def attach_traffic(table, pmc):
    """Per-launch memory-side traffic (committed rocprofv3 PMC summary) for the rows of a kernel table.  Freshness: the file
    lists the labels its passes saw (`labels`, hashed in `label_hash`); if the library now launches a label with a PMC key
    that the file does not list -- a kernel was renamed, re-tiled or replaced since the passes -- NO figure is attached and the
    missing labels are returned: a stale file is refused, not quoted.  Files without the list (rounds 2-3) are used as before."""
    kern = pmc.get("kernels", {})
    listed = set(pmc.get("labels", []))
    current = {row["kernel"] for row in table if pmc_key(row["kernel"])}
    missing = sorted(current - listed) if listed else []
    if missing:
        return missing
    shared = collections.Counter(pmc_key(row["kernel"]) for row in table)
    for row in table:
        key = pmc_key(row["kernel"])
        hit = None
        if key:
            hit = kern.get(key + "#" + row["kernel"])
            # a kernel|grid key behind several shapes whose launch order differs between steps (teacher-forced steps add
            # T = 6 launches of the forward chain kernel) has no per-shape PMC figure: null rather than the mixture
            if hit is None and shared[key] == 1:
                hit = kern.get(key) or next((v for k, v in kern.items() if k.startswith(key)), None)
        if hit:
            row["traffic_mbytes_per_launch"] = hit.get("hbm_mbytes_per_launch")
            row["pmc_key"] = key
            for extra in ("mfma_busy_frac", "gpu_busy_frac"):          # MFMA-busy counter pass (tools/profile.sh), when merged in
                if extra in hit:
                    row[extra] = hit[extra]
    return []


def pmc_key(label):
    """Kernel-name|grid key of tools/pmc_summary.py (rocprofv3 PMC pass) for a profile label of the library."""
    import math
    import re
    f = {k: int(v) for k, v in re.findall(r"\b(ms|np|nc|pk|x|T|B|H|M|N|K|s)(\d+)", label)}
    tf = lambda v: "true" if v else "false"
    m2 = re.match(r"gru_chain_(fwd|bwd) v2w(\d+)(e?) p(\d+) np(\d+) T(\d+) B(\d+) H(\d+)", label)
    if m2:                                               # second generation (csrc/gru_chain2.hip): waves per workgroup, piece products
        em = m2.group(3) == "e"                          # ("e": the build that writes the ChainEmit piece outputs)
        kind, (wv, npp, nprob, _, B_, H) = m2.group(1), [int(x) for x in (m2.group(2),) + m2.groups()[3:]]
        groups = nprob * math.ceil(math.ceil(B_ / 16) / wv)
        grid = 64 * wv * 8 * (H // 16) * math.ceil(groups / 8)
        s32 = H // 32 if kind == "fwd" else 3 * H // 32
        return f"gru_chain2_{kind}_kernel<{wv}, {s32}, {npp}, {tf(em)}>|g{grid}"
    if label.startswith("gru_chain_fwd") or label.startswith("gru_chain_bwd"):
        fwd = label.startswith("gru_chain_fwd")
        H, ms = f["H"], f["ms"]
        groups = f["np"] * math.ceil(f["B"] / (16 * ms))
        grid = 256 * 8 * (H // 16) * math.ceil(groups / 8)
        if fwd:                                          # third parameter: the build for two launches per CU ("ms4x2")
            return f"gru_chain_fwd_kernel<{ms}, {H // 64}, {2 if re.search(r'ms[0-9]+x2', label) else 1}>|g{grid}"
        emr = re.search(r" ms\d+e ", label) is not None    # ("ms4e": the build that writes the row pieces of dgi)
        return f"gru_chain_bwd_kernel<{ms}, {3 * H // 64}, {tf(emr)}>|g{grid}"
    if label.startswith("gru_fwd"):
        grid = 256 * f["np"] * math.ceil(f["B"] / (16 * f["ms"])) * (f["H"] // 16)
        return f"gru_step_fwd_kernel<{tf(f['x'])}, {f['ms']}, {tf(f['pk'])}>|g{grid}"
    if label.startswith("gru_bwd"):
        grid = 256 * f["np"] * math.ceil(f["B"] / (16 * f["ms"])) * (f["H"] // (16 * f["nc"]))
        return f"gru_step_bwd_kernel<{f['ms']}, {f['nc']}, {tf(f['pk'])}>|g{grid}"
    if label.startswith("decode_chain"):
        H, ms = f["H"], f["ms"]
        grid = 256 * 8 * (H // 16) * math.ceil(math.ceil(f["B"] / (16 * ms)) / 8)
        train = label.startswith("decode_chain_train")
        return f"decode_chain_kernel<{ms}, {H // 64}, {tf(train)}, 0, 0>|g{grid}"
    if label == "adam":
        return "adam_kernel|"
    mb = re.match(r"M(\d+) N(\d+) K(\d+) bf3p(\d) t(\d+)x(\d+) s(\d+)(?: e\d+)?(?: x(\d+))?", label)
    if mb:                                               # csrc/gemm_bf3.hip: 512 threads; 192x192 = waves 2x4 of 6x3 tiles, 192x128 = 4x2 of 3x4
        M, N, K, npp, bm, bn, sp, nb = mb.groups()
        grid = 512 * (int(M) // int(bm)) * (int(N) // int(bn)) * int(sp) * int(nb or 1)
        cfg = "2, 4, 6, 3" if int(bn) == 192 else "4, 2, 3, 4"
        return f"gemm_bf3_kernel<{cfg}, {npp}>|g{grid}"
    m = re.match(r"M(\d+) N(\d+) K(\d+) ([TN])([TN]) ([tdk])(\d+)x(\d+) s(\d+)(?: e\d+)?(?: x(\d+))?", label)
    if m:
        M, N, K, a, b, kind, bm, bn, sp, nb = m.groups()
        grid = 256 * math.ceil(int(N) / int(bn)) * math.ceil(int(M) / int(bm)) * int(sp) * int(nb or 1)   # xN: N products per launch
        if kind == "t":                                  # LDS-tiled
            return f"gemm_kernel<{int(bm) // 64}, {int(bn) // 64}, {tf(a == 'T')}, {tf(b == 'N')}>|g{grid}"
        if kind == "k":                                  # workgroup split-K
            return f"gemm_ks_kernel<{int(bm) // 16}, {int(bn) // 16}, {tf(a == 'T')}, {tf(b == 'N')}>|g{grid}"
        if a == "T":                                     # direct, k-major x k-major
            return f"gemm_tn_direct_kernel<{int(bm) // 64}, {int(bn) // 64}>|g{grid}"
        return f"gemm_kc_direct_kernel<{int(bm) // 32}, {int(bn) // 32}, {tf(b == 'N')}>|g{grid}"
    return None


def by_instantiation(table):
    """The same rows grouped by TEMPLATE INSTANTIATION (several shapes can share one: VERDICT r04 -- `roofline.kernel` is the
    largest group by label; by instantiation gemm_bf3_kernel<4, 2, 3, 4, 9> with its three launches is larger)."""
    groups = {}
    for r in table:
        key = pmc_key(r["kernel"])
        name = key.split("|")[0] if key else r["kernel"]
        g = groups.setdefault(name, {"instantiation": name, "labels": [], "launches_per_step": 0.0, "ms_per_step": 0.0, "gflop": 0.0,
                                     "peak_tflops": r["peak_tflops"], "mfma_pipe": r["mfma_pipe"]})
        g["labels"].append(r["kernel"])
        g["launches_per_step"] += r["launches_per_step"]
        g["ms_per_step"] += r["ms_per_step"]
        g["gflop"] += r["gflop_per_launch"] * r["launches_per_step"]
    out = []
    for g in groups.values():
        tf = g["gflop"] / g["ms_per_step"] if g["ms_per_step"] > 0 else 0.0          # GFLOP / ms = TFLOP/s
        out.append({"instantiation": g["instantiation"], "labels": g["labels"], "launches_per_step": round(g["launches_per_step"], 2),
                    "ms_per_step": round(g["ms_per_step"], 4), "tflops": round(tf, 2), "mfma_pipe": g["mfma_pipe"],
                    "peak_tflops": g["peak_tflops"], "frac_mfma": round(tf / g["peak_tflops"], 4)})
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def roofline(step):
    table = kernel_table(step)
    top = table[0]
    pmc = {}
    if os.path.exists(PMC_FILE):
        try:
            pmc = json.load(open(PMC_FILE))
        except Exception:
            pmc = {}
    stale = attach_traffic(table, pmc)
    mfma_bound = top["frac_mfma"] >= top["frac_hbm"]
    out = {"bound": "mfma" if mfma_bound else "hbm", "kernel": top["kernel"],
           "achieved": top["tflops"] if mfma_bound else top["gbps"],
           "peak": top["peak_tflops"] if mfma_bound else PEAK_HBM_GBPS,
           "mfma_pipe": top["mfma_pipe"],
           # continuity with rounds 1-2, which priced every kernel against the f32-input MFMA peak
           "frac_of_f32_input_peak": round(top["tflops"] / PEAK_F32_MFMA_TFLOPS, 4),
           "unit": "TFLOP/s" if mfma_bound else "GB/s",
           "frac": top["frac_mfma"] if mfma_bound else top["frac_hbm"],
           "traffic": top.get("traffic_mbytes_per_launch"),
           "traffic_unit": "MB of HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, gfx950-corrected; "
                           f"{os.path.relpath(PMC_FILE, REPO)})",
           "launches_per_step": top["launches_per_step"], "avg_launch_us": top["avg_us"],
           "algorithmic_gflop_per_launch": top["gflop_per_launch"],
           "algorithmic_mbytes_per_launch": top["mbytes_per_launch"],
           "step_gflop": round(sum(r["gflop_per_launch"] * r["launches_per_step"] for r in table), 2),
           "step_kernel_ms": round(sum(r["ms_per_step"] for r in table), 4),
           "largest_by_instantiation": by_instantiation(table)[:3],
           "kernels": table[:8]}
    out["traffic_label_hash"] = pmc.get("label_hash")
    if stale:
        out["traffic_stale"] = {"reason": "the library launches labels the committed PMC file has never seen: its figures are "
                                          "not quoted (regenerate with tools/profile.sh)", "missing_labels": stale[:8]}
    return out


# ------------------------------------------------------------------------------------------------ extras
def decode_latency_extra(vae, iters=20):
    """Inference-side number for the tick decode (SURVEY 8d 'K7'): full HierarchicalDecoder.forward in eval mode at
    small batch.  The HBM figure counts the 26.44 MB of decoder weights ONCE per call, as SURVEY.md 8d prescribes; the
    north_star target for this call is 40 % of the 8 TB/s roofline."""
    out = {}
    vae.eval()
    for b in (1, 2, 4, 16, 256):
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
        gbps = 26.44e6 / (ms * 1e-3) / 1e9
        out[f"b{b}"] = {"ms_per_call": round(ms, 4), "measures_per_s": round(b / ms * 1e3, 1),
                        "weights_once_GBps": round(gbps, 2), "frac_hbm_roofline": round(gbps / PEAK_HBM_GBPS, 5)}
    out["north_star_target_frac"] = 0.40
    # What bounds the b = 1 call is a chain of dependent hand-offs AND the work between them, not bytes.  Per tick (in-kernel stamps of
    # workgroup CB_0, profiles/r06_decode_stamps.txt: tick period 3.28 us): ONE hand-off on the critical path -- the all-gather of h1_t
    # among the 16 layer-1 workgroups of csrc/decode_b1.hip's merged build, 0.8 us now that they share an XCD and write XCD-local
    # copies (1.54 us with agent-scope stores across XCDs in round 5; a one-to-one hand-off inside an XCD measures 0.6) -- plus
    # 2.47 us of LOCAL work no hand-off hides: the token's rows of the gather table and layer 0's cells 0.50, barrier + the W_ih1
    # product + layer 1's cells 0.67, barrier + head + barrier 0.62, argmax 0.19, the next tick's recurrent summands 0.25, the rest
    # stamps' own cost.  In front of the 24 ticks: five hand-offs of beat 0 (z2b -> beat layer 0 -> layer 1 -> projection -> cgi)
    # with one 512-wide product behind each.  The floor of THIS design is therefore hand-offs x 0.6 us + 24 x the measured local
    # work -- not the 0.023 ms round 5 printed from the hand-offs alone, and nowhere near the 40 % HBM target (8.3 us per call).
    handoffs, local_us = 24 * 1 + 5, 2.47
    out["b1"]["latency_floor_ms"] = round((handoffs * 0.6 + 24 * local_us + 5 * 0.4) * 1e-3, 4)
    out["b1"]["latency_floor"] = (f"{handoffs} dependent hand-offs x 0.6 us (one per tick + five for beat 0, same XCD) + 24 ticks x {local_us} us of "
                                  "local work between them (cells, two products, three barriers, argmax: profiles/r06_decode_stamps.txt) + five "
                                  "beat-path products x 0.4 us")
    vae.train()
    return {"decoder_eval": out}


def epoch_loop_extra(wl, batches=100):
    """The same step driven by Trainer.loss_and_acc_on_epoch over the input feed (feed.BatchLoader: pinned staging, shuffle;
    feed.DeviceFeed: copy stream two batches ahead; on-device int32 -> int64 widening): 16 sequences = 256 measures per
    batch, every batch different.  SURVEY.md section 8 f2: the feed must not cost throughput against the resident batch."""
    from inpaintnet_amd import synthetic
    from inpaintnet_amd.feed import BatchLoader
    score, md = synthetic.SyntheticFolkDataset(num_notes=NUM_NOTES, n_seq=16 * batches, seed=3).tensors()
    loader = BatchLoader((torch.from_numpy(score), torch.from_numpy(md)), 16, shuffle=True)
    wl.trainer.dataset.n_bars = 16
    wl.model.train()
    wl.trainer.loss_and_acc_on_epoch(loader, 0, train=True)           # warm-up epoch (allocator, staging ring)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss, acc = wl.trainer.loss_and_acc_on_epoch(loader, 0, train=True)
    dt = time.perf_counter() - t0                                     # (the epoch ends with its one host sync)
    return {"epoch_loop": {"measures_per_s": round(256 * batches / dt, 1), "ms_per_step": round(1e3 * dt / batches, 4),
                           "batches": batches, "mean_loss": round(loss, 4),
                           "workload": "VAETrainer.loss_and_acc_on_epoch over feed.BatchLoader/DeviceFeed, 256 measures "
                                       "per batch from pinned host memory, shuffled"}}


def vae4096_extra(wl, batch=4096, steps=6, warmup=2):
    """The reference's default MeasureVAE step (train_measure_vae.py:33: batch_size 256 sequences x 16 bars = 4096 measures,
    vae_trainer.py:49-52) on the same model: more rows than one resident chain launch holds.  Up to INET_CHAIN_CHUNK_MAX
    (1024) rows the recurrent layers would run as chain launches over row chunks (csrc/seq.hip chain_chunk_rows); at 4096 the
    H = 512 layers run one launch per time step over all rows, the faster form there (`per_step_gru_launches` counts them).
    Secondary number; the headline stays BASELINE.json's configs[1] (256 measures per GPU)."""
    from inpaintnet_amd import ops, synthetic
    dev = wl.tokens.device
    tok = torch.from_numpy(synthetic.det_tokens("bench/4096", (batch, 24), NUM_NOTES)).to(dev)
    t = wl.trainer

    def step():
        t.zero_grad()
        loss, acc = t.loss_and_acc_for_batch(tok, 0, train=True)
        loss.backward()
        t.step()
        return loss
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # which kernels ran: per-step launches of the H = 512 layers would mean the chunked chain path was not taken
    ops.prof_enable(True)
    step()
    torch.cuda.synchronize()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "l.csv")
        ops.prof_dump(path)
        labels = [r["label"] for r in csv.DictReader(open(path))]
    ops.prof_enable(False)
    per_step = sum(1 for l in labels if l.startswith("gru_fwd") or l.startswith("gru_bwd"))
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"vae_train_4096": {"measures_per_s": round(batch * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3),
                               "batch": batch, "launches_per_step": len(labels), "per_step_gru_launches": per_step,
                               # per-kernel table (side streams off, per-launch events; mean over the coin of the profiled steps)
                               "kernels": secondary_table(step, top=12, ms_per_step=1e3 * dt / steps),
                               "final_loss": round(float(loss.detach()), 5),
                               "workload": "MeasureVAE training at the reference's default batch: 256 sequences x 16 bars = "
                                           "4096 measures per step (train_measure_vae.py:33); above INET_CHAIN_CHUNK_MAX = 1024 rows the H = 512 "
                                           "layers run one launch per time step (per_step_gru_launches), which is the "
                                           "faster form there (profiles/r03_e_batch_crossover.txt)"}}


def vocab_extra(num_notes=61, steps=40, warmup=8):
    """The headline step with a vocabulary that is not a multiple of 16 (the real one is data-derived,
    MeasureVAE/measure_vae.py:56; 48 is this bench's stand-in): the fused decode kernel pads its last column block and the
    token segment-sum kernels take up to 128 rows, so such a vocabulary stays on the fast paths."""
    from inpaintnet_amd import ops, synthetic
    from inpaintnet_amd.measure_vae import MeasureVAE
    from inpaintnet_amd.vae_trainer import VAETrainer
    ds = synthetic.SyntheticFolkDataset(num_notes=num_notes)
    model = MeasureVAE(ds)
    model.load_state_dict({k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape))) for k, v in model.state_dict().items()})
    tr = VAETrainer(ds, model, lr=1e-4)
    tr.overlap_backward = True
    model.train()
    tok = torch.from_numpy(synthetic.det_tokens("bench/vocab", (VAE_BATCH_PER_GPU, 24), num_notes)).cuda()

    def step():
        tr.zero_grad()
        loss, acc = tr.loss_and_acc_for_batch(tok, 0, train=True)
        loss.backward()
        tr.step()
    random.seed(77)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ops.prof_enable(True)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "l.csv")
        ops.prof_dump(path)
        labels = [r["label"] for r in csv.DictReader(open(path))]
    ops.prof_enable(False)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {f"vae_train_v{num_notes}": {"measures_per_s": round(VAE_BATCH_PER_GPU * steps / dt, 1),
                                        "ms_per_step": round(1e3 * dt / steps, 4), "num_notes": num_notes,
                                        "fused_decode_launches": sum(l.startswith("decode_chain") for l in labels),
                                        "per_tick_gru_launches": sum(l.startswith("gru_fwd") or l.startswith("gru_bwd") for l in labels),
                                        "workload": f"the headline step with V = {num_notes} (not a multiple of 16)"}}


def chain_generations_extra(wl, steps=60, warmup=10):
    """The headline step under the other forms of its products (inet_set_option keys 7: chain kernels, 8: large products):
    everything on the f32-input MFMA (round 2's arithmetic), and the chain kernels alone on the bf16 pipe."""
    from inpaintnet_amd import ops
    out = {}
    try:
        for chain, gemm, key in ((0, 0, "f32_input_mfma"), (9, 0, "bf16_split_chains_only")):
            ops.set_option(7, chain)
            ops.set_option(8, gemm)
            for _ in range(warmup):
                wl.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                wl.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[key] = {"measures_per_s": round(wl.units_per_step * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 4)}
    finally:
        ops.set_option(7, 9)
        ops.set_option(8, 9)
    return {"chain_generations": out}


def arnn_extra(batch=32, steps=30, warmup=4, tables=True, free_steps=10):
    """Secondary number (BASELINE.json configs[4]): AnticipationRNN gauss-reg model, teacher-forced training step,
    batch 32 sequences of 384 ticks, script defaults of train_arnn_reg.py.  Not the headline metric."""
    import types
    from inpaintnet_amd import synthetic
    from inpaintnet_amd.arnn import AnticipationRNNGaussianRegTrainer, ConstraintModelGaussianReg, free_positions
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
        free = free_positions(data[2])
        loss, acc = trainer.mean_crossentropy_loss_and_accuracy_voices(weights, data[0][:, :, free].transpose(0, 1))
        loss.backward()
        trainer.step()
    dt, _ = timed(step, steps, warmup, torch.cuda.synchronize)

    def step_fr():                                   # the coin's other side (the reference draws it per batch, p = 0.5)
        trainer.zero_grad()
        weights, _ = model(data[0], data[1], data[2], data[3], data[4], train=True, teacher_forcing=False)
        free = free_positions(data[2])
        loss, acc = trainer.mean_crossentropy_loss_and_accuracy_voices(weights, data[0][:, :, free].transpose(0, 1))
        loss.backward()
        trainer.step()
    dt_fr, _ = timed(step_fr, free_steps, 2, torch.cuda.synchronize)

    # ... and as AnticipationRNNGaussianRegTrainer.loss_and_acc_for_batch calls the model (trim=True): the generation LSTMs stop behind the
    # last unconstrained tick and the head runs on the unconstrained ticks only -- what the reference computes there is read by nobody
    # (identical weights and gradients: tests/test_gpu_arnn.py); the lines above time ALL 384 ticks, as the reference computes them
    torch.manual_seed(6)
    many = [trainer.process_batch_data((score, md)) for _ in range(16)]      # sixteen windows drawn by the trainer's own sampler
    turn = [0]

    def step_trim(tf):
        d = many[turn[0] % len(many)]
        turn[0] += 1
        trainer.zero_grad()
        weights, _ = model(d[0], d[1], d[2], d[3], d[4], train=True, teacher_forcing=tf, trim=True)
        free = free_positions(d[2])
        loss, acc = trainer.mean_crossentropy_loss_and_accuracy_voices(weights, d[0][:, :, free].transpose(0, 1))
        loss.backward()
        trainer.step()
    nst = max(steps, 2 * len(many))
    dt_t, _ = timed(lambda: step_trim(True), nst, len(many), torch.cuda.synchronize)
    turn[0] = 0
    dt_tf, _ = timed(lambda: step_trim(False), 2 * len(many), len(many), torch.cuda.synchronize)
    trainer.finish()
    return {"anticipation_rnn_train": {"sequences_per_s": round(batch * steps / dt, 1),
                                       "measures_per_s": round(16 * batch * steps / dt, 1),
                                       "ms_per_step": round(1e3 * dt / steps, 3),
                                       "ms_per_step_free_running": round(1e3 * dt_fr / free_steps, 3),
                                       "ms_per_step_mean_of_the_coin": round(0.5e3 * (dt / steps + dt_fr / free_steps), 3),
                                       "default_trimmed": {
                                           "ms_per_step": round(1e3 * dt_t / nst, 3),
                                           "ms_per_step_free_running": round(1e3 * dt_tf / (2 * len(many)), 3),
                                           "ms_per_step_mean_of_the_coin": round(0.5e3 * (dt_t / nst + dt_tf / (2 * len(many))), 3),
                                           "window_end_ticks": [int(d[4]) for d in many],
                                           "what": "the trainer's own call (trim=True) over sixteen windows drawn by its sampler: the generation "
                                                   "LSTMs and the head skip the ticks behind the window, which nobody reads (window_end_ticks .. "
                                                   "383; the lines above use ONE batch whose window ends at tick %d and compute all 384 ticks)" % int(data[4])},
                                       "kernels": secondary_table(step, ms_per_step=1e3 * dt / steps) if tables else None,
                                       "workload": "AnticipationRNN gauss-reg (LSTM 2x2 layers, H=256), teacher-forced "
                                                   "train step, 32 sequences x 384 ticks"}}


# ------------------------------------------------------------------------------------------------ main
LINE_LIMIT = 4096        # bytes: the driver's capture dropped round 5's 20 KB line (BENCH_r05.json: parsed null)
DETAIL_FILE = os.environ.get("INET_BENCH_DETAIL", os.path.join(REPO, "bench_detail.json"))


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(out):
    """The ONE stdout line: the contract's keys, the dominant kernel's roofline row, the CPU baseline and one scalar per secondary
    workload -- hard limit LINE_LIMIT bytes (tests/test_bench_keys.py).  Kernel tables, per-step host times, parity detail and
    workload prose stay in `out`, which main() writes to DETAIL_FILE and stderr."""
    ex = out.get("extras") or {}
    cfg = out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg[k] for k in ("workload", "batch_per_gpu", "global_batch", "parallelism", "final_loss") if k in cfg}
    r = out.get("roofline")
    line["roofline"] = None if not r else {k: r.get(k) for k in (
        "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "mfma_pipe", "frac_of_f32_input_peak",
        "algorithmic_mbytes_per_launch", "algorithmic_gflop_per_launch", "avg_launch_us", "launches_per_step", "step_gflop",
        "step_kernel_ms")}
    c = out.get("cpu_baseline")
    line["cpu_baseline"] = None if not c else {k: c.get(k) for k in ("value", "unit", "cores", "kind", "sample")}
    for k in ("parity_checked", "max_rel_err", "kink_violations", "chain_timeouts", "slow_waits", "waits_noted"):
        if k in out:
            line[k] = out[k]
    if out.get("first_steps_ms"):
        line["first_steps_ms"] = out["first_steps_ms"][:20]
    if _get(out, "gc_in_timed_region", "collections") is not None:
        line["gc_in_timed_region"] = out["gc_in_timed_region"]["collections"]
    for k in ("per_rank_units_per_s", "allreduce_ms_per_step", "allreduce_mbytes"):
        if k in out:
            line[k] = out[k]
    if out.get("dp"):
        line["dp"] = {k: out["dp"].get(k) for k in ("ms_per_step_without_exchange", "exposed_exchange_ms_per_step",
                                                     "chain_timeouts_per_rank", "skipped_steps_per_rank")}
    scal = {
        "latent_ms": _get(ex, "latent_rnn_train", "ms_per_step"),
        "latent_seq_per_s": _get(ex, "latent_rnn_train", "sequences_per_s"),
        "latent_default_ms": _get(ex, "latent_rnn_train_default", "ms_per_step"),
        "latent_ar_ms": _get(ex, "latent_rnn_train_auto_reg", "ms_per_step"),
        "latent_ar_fr_ms": _get(ex, "latent_rnn_train_auto_reg", "ms_per_step_free_running"),
        "latent_dp_ms": _get(ex, "latent_rnn_train_dp", "ms_per_step"),
        "latent_dp_seq_per_s": _get(ex, "latent_rnn_train_dp", "sequences_per_s"),
        "arnn_tf_ms": _get(ex, "anticipation_rnn_train", "ms_per_step"),
        "arnn_fr_ms": _get(ex, "anticipation_rnn_train", "ms_per_step_free_running"),
        "arnn_trimmed_tf_ms": _get(ex, "anticipation_rnn_train", "default_trimmed", "ms_per_step"),
        "decode_b1_ms": _get(ex, "decoder_eval", "b1", "ms_per_call"),
        "decode_b2_ms": _get(ex, "decoder_eval", "b2", "ms_per_call"),
        "decode_b4_ms": _get(ex, "decoder_eval", "b4", "ms_per_call"),
        "decode_b16_ms": _get(ex, "decoder_eval", "b16", "ms_per_call"),
        "decode_b256_ms": _get(ex, "decoder_eval", "b256", "ms_per_call"),
        "decode_b1_frac_hbm": _get(ex, "decoder_eval", "b1", "frac_hbm_roofline"),
        "decode_b1_floor_ms": _get(ex, "decoder_eval", "b1", "latency_floor_ms"),
        "epoch_loop_measures_per_s": _get(ex, "epoch_loop", "measures_per_s"),
        "vae4096_measures_per_s": _get(ex, "vae_train_4096", "measures_per_s"),
        "vae4096_ms": _get(ex, "vae_train_4096", "ms_per_step"),
        "vae_v61_ms": _get(ex, "vae_train_v61", "ms_per_step"),
        "f32_input_chains_ms": _get(ex, "chain_generations", "f32_input_mfma", "ms_per_step"),
    }
    scal = {k: v for k, v in scal.items() if v is not None}
    if scal:
        line["extras"] = scal
    line["detail"] = os.path.basename(DETAIL_FILE)
    # the limit is hard: optional keys go first, then the config's prose, before a line the driver cannot read is printed
    for drop in ("first_steps_ms", "extras", "dp", "per_rank_units_per_s"):
        if len(json.dumps(line)) < LINE_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) >= LINE_LIMIT:
        line["config"]["workload"] = line["config"].get("workload", "")[:120]
        if line.get("cpu_baseline"):
            line["cpu_baseline"].pop("sample", None)
    text = json.dumps(line)
    assert len(text) < LINE_LIMIT, len(text)
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)         # >= 2 s of timed GPU work at ~5 ms per step
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=("vae", "latent"), default="vae")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))            # nothing above touched the GPU
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}: refusing to report a mismatched run",
              file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # INET_BENCH_SHARE_GPU=1: a FUNCTIONAL check of the N > 1 code path on a box with one GPU -- all ranks use cuda:0 and gloo
    # carries the exchange (RCCL wants one device per rank); the line it prints says so and is not a scaling number
    share_gpu = os.environ.get("INET_BENCH_SHARE_GPU") == "1" and world_env > 1
    if share_gpu:
        local_rank = 0
        # Every persistent launch must be resident in full, and two processes on one GPU cannot promise each other that (a first
        # version gave each rank half the CUs, INET_CHAIN_CUS=128: a chain still ran into its bounded spin -- and both ranks
        # reported it and left together, which is the failure path working on a real timeout): the per-step kernels here.
        os.environ.setdefault("INET_CHAIN", "0")
    if local_rank >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)",
              file=sys.stderr)
        sys.exit(2)

    from inpaintnet_amd import dp, ops
    world = dp.init_from_env(backend="gloo" if share_gpu else "nccl") if world_env > 1 else 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # what RCCL really connected: one element per rank summed over the ring
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)
        world = int(ones.item())
        if world != args.gpus:
            print(f"bench.py: RCCL reduced over {world} ranks, --gpus says {args.gpus}", file=sys.stderr)
            sys.exit(2)

    coin = os.environ.get("INET_BENCH_COIN")                   # profiling aid (tools/profile_r03.sh): every step teacher-forced
    if coin in ("tf", "fr"):                                   # ("tf") or free-running ("fr") instead of the per-step coin, so that
        random.random = (lambda: 0.0) if coin == "tf" else (lambda: 0.99)   # every step launches the same kernel sequence
    dp.seed_rank(1234, rank)                                   # per-rank eps / dropout streams
    dp.seed_shared(4321)                                       # the teacher-forcing coin is shared by all ranks
    wl = (VaeWorkload if args.workload == "vae" else LatentWorkload)(dev, rank)
    dp.broadcast_params(wl.model.flat)
    ops.slow_waits(reset=True)                                 # the recorder covers this process's steps from here on

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    try:
        trace = {}
        dt, loss = timed(wl.step, args.steps, args.warmup, fence, trace)
    except ops.ChainTimeoutError as e:
        # raised by Trainer.step() on EVERY rank at the same step (the decision travels with the gradients): all ranks leave
        print(f"[bench] rank {rank}: {e}\n[bench] result invalid", file=sys.stderr)
        sys.exit(3)
    dt_local = dt
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    final_loss = float(loss.detach())
    from inpaintnet_amd import ops as _ops
    # Persistent-kernel health of the timed region, per rank: workgroups that gave up waiting (inet_chain_status) and optimizer
    # steps that skipped themselves (the step reports; under data parallelism every rank skips a step ANY rank failed in).  The
    # verdict is collective: all ranks exit together, nobody is left waiting in the next all-reduce.
    skipped_steps = 0
    try:
        wl.trainer.check_steps(wait_all=True)
    except _ops.ChainTimeoutError:
        skipped_steps = len(wl.trainer._lost)
    chain_timeouts = max(_ops.chain_status(reset=True), 0)
    slow = _ops.slow_waits(reset=True)                         # waits > ~50 us inside persistent kernels during warm-up + timed steps
    health = torch.zeros(2 * world, dtype=torch.float64, device=dev)
    health[2 * rank], health[2 * rank + 1] = float(chain_timeouts), float(skipped_steps)
    if world > 1:
        torch.distributed.all_reduce(health)
    health = [int(v) for v in health.tolist()]
    timeouts_per_rank, skipped_per_rank = health[0::2], health[1::2]
    if sum(health) > 0:
        print(f"[bench] rank {rank}: chain-kernel workgroups timed out in the timed region (per rank {timeouts_per_rank}, "
              f"skipped optimizer steps per rank {skipped_per_rank}): result invalid", file=sys.stderr)
        if world > 1:
            torch.distributed.destroy_process_group()
        sys.exit(3)

    per_rank = None
    ar_ms = None
    dp_report = None
    if world > 1:
        mine = torch.zeros(world, dtype=torch.float64, device=dev)
        mine[rank] = wl.units_per_step * args.steps / dt_local
        torch.distributed.all_reduce(mine)
        per_rank = [round(v, 1) for v in mine.tolist()]
        ar_ms = allreduce_ms(wl.model.grad)
        dp_report = exchange_report(wl, dp, dt / args.steps, fence, dev)
        dp_report["chain_timeouts_per_rank"] = timeouts_per_rank
        # (dp.one_side_stream: a process with an exchange keeps ONE side stream in rotation -- DESIGN.md section 6)
        dp_report["side_streams_in_rotation"] = int(os.environ.get("INET_DP_SIDE_STREAMS", "1"))
        dp_report["skipped_steps_per_rank"] = skipped_per_rank

    extras = {}
    roof = None
    if rank == 0 and os.environ.get("INET_BENCH_SEQ"):         # tools/profile_r02.sh: launch order of shape-sharing kernels
        dump_kernel_sequences(wl.step, os.environ["INET_BENCH_SEQ"])
    if rank == 0 and not args.no_roofline:
        roof = roofline(wl.step)
    if world > 1:
        torch.distributed.barrier()

    if not args.no_extras:
        if world > 1 and args.workload == "vae":
            # BASELINE.json configs[3]: the LatentRNN step under the same data-parallel exchange (159.6 MB arena)
            lw = LatentWorkload(dev, rank, vae=wl.model, ds=wl.ds)
            dp.broadcast_params(lw.model.flat)
            ldt, _ = timed(lw.step, 10, 3, fence)
            t = torch.tensor([ldt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            lar = allreduce_ms(lw.model.grad)
            extras["latent_rnn_train_dp"] = {
                "sequences_per_s": round(world * LATENT_SEQ_PER_GPU * 10 / float(t.item()), 1),
                "measures_per_s": round(world * 16 * LATENT_SEQ_PER_GPU * 10 / float(t.item()), 1),
                "ms_per_step": round(1e3 * float(t.item()) / 10, 3), "allreduce_ms": lar,
                "arena_mbytes": round(lw.arena_mb, 1), **lw.describe(world)}
        elif rank == 0 and world == 1:
            if args.workload == "vae":
                lw = LatentWorkload(dev, rank, vae=wl.model, ds=wl.ds)
                ldt, _ = timed(lw.step, 20, 4, fence)
                extras["latent_rnn_train"] = {"sequences_per_s": round(LATENT_SEQ_PER_GPU * 20 / ldt, 1),
                                              "measures_per_s": round(16 * LATENT_SEQ_PER_GPU * 20 / ldt, 1),
                                              "ms_per_step": round(1e3 * ldt / 20, 3), **lw.describe(1),
                                              "kernels": secondary_table(lw.step, ms_per_step=1e3 * ldt / 20)}
                del lw
                ld = LatentWorkload(dev, rank, vae=wl.model, ds=wl.ds, encode_all=False)
                ddt, _ = timed(ld.step, 20, 4, fence)
                extras["latent_rnn_train_default"] = {
                    "sequences_per_s": round(LATENT_SEQ_PER_GPU * 20 / ddt, 1), "measures_per_s": round(16 * LATENT_SEQ_PER_GPU * 20 / ddt, 1),
                    "ms_per_step": round(1e3 * ddt / 20, 3),
                    "workload": "the same step as latent_rnn_train with the package's default: the target measures are NOT encoded when "
                                "nothing reads their latents (auto_reg=False: never; the reference encodes them in every forward pass and "
                                "drops the result, latent_rnn.py:133,148-149): 12 instead of 16 frozen-encoder passes per sequence, "
                                "identical outputs, gradients and weights",
                    "kernels": secondary_table(ld.step, ms_per_step=1e3 * ddt / 20)}
                del ld
                random.seed(99)
                la = LatentWorkload(dev, rank, vae=wl.model, ds=wl.ds, auto_reg=True)
                adt, _ = timed(la.step, 20, 4, fence)
                # the two sides of the teacher-forcing coin on their own (the coin is LatentRNN.forward's `random.random() < 0.5`): the
                # teacher-forced side is the non-auto-regressive step plus the target encodes, the free-running side decodes and
                # re-encodes measure by measure at 128 rows -- 280 launches, GPU-bound (profiles/r06_e_latent_ar_fr_table.txt)
                sides = {}
                real_random = random.random
                try:
                    for name, val in (("teacher_forced", 0.0), ("free_running", 0.99)):
                        random.random = (lambda v=val: v)
                        sdt, _ = timed(la.step, 10, 2, fence)
                        sides[name] = round(1e3 * sdt / 10, 3)
                finally:
                    random.random = real_random
                extras["latent_rnn_train_auto_reg"] = {"sequences_per_s": round(LATENT_SEQ_PER_GPU * 20 / adt, 1),
                                                       "measures_per_s": round(16 * LATENT_SEQ_PER_GPU * 20 / adt, 1),
                                                       "ms_per_step": round(1e3 * adt / 20, 3),
                                                       "ms_per_step_teacher_forced": sides.get("teacher_forced"),
                                                       "ms_per_step_free_running": sides.get("free_running"), **la.describe(1),
                                                       # (the mean over the teacher-forcing coin of the four profiled steps)
                                                       "kernels": secondary_table(la.step, ms_per_step=1e3 * adt / 20)}
                del la
                wl.model.trainable = True                      # (LatentRNN froze the shared VAE)
                wl.model.train()
            # (AnticipationRNN: 1,536 dependent hand-offs per step.  It used to measure 1 ms more here than alone -- the hardware queue
            #  its second chain's stream got depended on which workload had created its streams first; csrc/side.hip creates them
            #  in one order now: DESIGN.md section 8, tools/arnn_order.py)
            extras.update(arnn_extra())
            vae = wl.model if args.workload == "vae" else wl.model.vae_model
            extras.update(decode_latency_extra(vae))
            if args.workload == "vae":
                wl.model.train()
                extras.update(epoch_loop_extra(wl))
                for fn, key in ((lambda: chain_generations_extra(wl), "chain_generations"),
                                (lambda: vae4096_extra(wl), "vae_train_4096"), (vocab_extra, "vae_train_v61")):
                    try:
                        extras.update(fn())
                    except Exception as e:                   # a secondary number must never take the headline down
                        extras[key] = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": "measures/sec training (MeasureVAE: fwd + CE/KL + bwd + Adam)" if args.workload == "vae" else
                      "measures/sec training (LatentRNN + frozen MeasureVAE: fwd + CE + bwd + Adam)",
            "value": round(world * wl.units_per_step * args.steps / dt, 2),
            "unit": wl.unit,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "chain_timeouts": chain_timeouts,                      # inet_chain_status after the timed region: 0 = healthy
            "slow_waits": slow["count"],                           # the recorder over warm-up + timed steps: waits of 16384+ polls (~6 ms) / given up
            "waits_noted": slow["noted"],                          # ... of 64+ polls (overlapping launches becoming resident: normal)
            "slow_wait_entries": slow["entries"][:16],
            # the timed steps one by one (the first 32): GPU time between events behind consecutive steps, host time to queue each,
            # Python garbage collections inside the region -- a cold-start transient shows up HERE (profiles/r05_cold_start.txt)
            **trace,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "arithmetic": "f32 throughout.  The recurrent contractions of the GRU chain kernels (csrc/gru_chain2.hip) and the "
                          "encoder's large products (csrc/gemm_bf3.hip: layer-1 input products, their data gradient, the weight "
                          "gradients) split every f32 operand EXACTLY into three bf16 pieces and accumulate all nine piece products "
                          "in f32 on the bf16 MFMA: the products of fp32 arithmetic, only the f32 summation order differs (error "
                          "against float64 equal to the f32-input form: tests/test_gpu_kernels.py::test_chain_generations_against_"
                          "float64, ::test_gemm_bf3_layouts); every other product runs on the f32-input MFMA (v_mfma_f32_*_f32).  "
                          "INET_CHAIN2=0 / INET_GEMM_BF3=0 select the f32-input forms (extras.chain_generations times them).",
            "data": "synthetic",
            "config": dict(wl.describe(world), final_loss=round(final_loss, 5)),
            "roofline": roof,
        }
        if world > 1:
            out["per_rank_units_per_s"] = per_rank
            out["allreduce_ms_per_step"] = ar_ms
            out["allreduce_mbytes"] = round(wl.arena_mb, 1)
            out["dp"] = dp_report
            if share_gpu:
                out["data"] = "synthetic; FUNCTIONAL CHECK ONLY: all ranks share one GPU and gloo carries the exchange"
        cpu = None
        if world == 1 and args.workload == "vae":
            if not args.no_parity:
                wl.model.trainable = True
                out.update(parity_check(wl.model, wl.tokens))
            if not args.no_cpu_baseline:
                cpu = cpu_baseline(VAE_BATCH_PER_GPU)
        out["cpu_baseline"] = cpu
        out["extras"] = extras or None
        detail = json.dumps(out)
        try:
            with open(DETAIL_FILE, "w") as f:                  # tables, per-step traces, parity detail, workload prose
                f.write(detail + "\n")
        except OSError as e:
            print(f"[bench] could not write {DETAIL_FILE}: {e}", file=sys.stderr)
        print("[bench detail] " + detail, file=sys.stderr, flush=True)
        print(compact_line(out), flush=True, file=_REAL_STDOUT)   # the ONE line of the contract (< LINE_LIMIT bytes)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


# stdout carries exactly one line, the JSON result: whatever the library prints on the way (the reference's own "Freeze the ..."
# message of LatentRNN, warnings of extensions) goes to stderr
_REAL_STDOUT = sys.stdout


if __name__ == "__main__":
    sys.stdout = sys.stderr
    try:
        main()
    finally:
        sys.stdout = _REAL_STDOUT
