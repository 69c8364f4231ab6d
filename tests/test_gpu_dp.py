"""Data-parallel path on the GPU: two ranks (sharing cuda:0; gloo carries the exchange, since RCCL wants one device per
rank) run the real HIP training step through VAETrainer with the epoch loop's deferred side-stream joins, the decoder
bucket started from inside backward and the rest summed in step().  Both ranks must end with identical parameters, equal
to a single process that takes the same two half-batches as one global batch."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(V=20, E=6, H=256, Z=24)     # H % 256 == 0: fragment-major step kernels
B, STEPS = 12, 3


def _build():
    from inpaintnet_amd import synthetic
    from inpaintnet_amd.measure_vae import MeasureVAE
    from inpaintnet_amd.vae_trainer import VAETrainer
    ds = synthetic.SyntheticFolkDataset(num_notes=CFG["V"])
    model = MeasureVAE(ds, note_embedding_dim=CFG["E"], encoder_hidden_size=CFG["H"], latent_space_dim=CFG["Z"],
                       decoder_hidden_size=CFG["H"], encoder_dropout_prob=0.0, decoder_dropout_prob=0.0)
    sd = {k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape))) for k, v in model.state_dict().items()}
    model.load_state_dict(sd)
    trainer = VAETrainer(ds, model, lr=1e-3)
    trainer.overlap_backward = True
    model.train()
    return model, trainer


def _first_grad(model, trainer, tokens, eps_all, coins, lo, hi):
    """Averaged gradient arena of step 0, exchanged exactly as Trainer.step() does it (join, all-reduce, 1/world)."""
    from inpaintnet_amd import dp, ops
    trainer.zero_grad()
    w, smp, zd, pd, z, zp = model(tokens[lo:hi].cuda(), train=True, eps=eps_all[0][lo:hi].cuda(), teacher_forced=coins[0])
    ce, acc = trainer.mean_crossentropy_loss_and_accuracy(w, tokens[lo:hi].cuda())
    (ce + trainer.compute_kld_loss(zd, pd)).backward()
    ops.side_join()
    scale = dp.allreduce_grads(model.grad)
    g = (model.grad * scale).cpu().numpy()
    ops.side_defer(False)
    return g


def _run_steps(model, trainer, tokens, eps_all, coins, lo, hi):
    losses = []
    for s in range(STEPS):
        eps = eps_all[s][lo:hi].cuda()
        trainer.zero_grad()
        w, smp, zd, pd, z, zp = model(tokens[lo:hi].cuda(), train=True, eps=eps, teacher_forced=coins[s])
        ce, acc = trainer.mean_crossentropy_loss_and_accuracy(w, tokens[lo:hi].cuda())
        loss = ce + trainer.compute_kld_loss(zd, pd)
        loss.backward()
        trainer.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    return losses


def _inputs():
    from inpaintnet_amd import synthetic
    tokens = torch.from_numpy(synthetic.det_tokens("dpgpu/tokens", (B, 24), CFG["V"]))
    eps_all = [torch.from_numpy(synthetic.det_normal(f"dpgpu/eps{s}", (B, CFG["Z"]))) for s in range(STEPS)]
    return tokens, eps_all, [True, False, True]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from inpaintnet_amd import dp
    assert dp.init_from_env(backend="gloo") == world
    torch.cuda.set_device(0)
    model, trainer = _build()
    dp.broadcast_params(model.flat)
    tokens, eps_all, coins = _inputs()
    lo, hi = dp.shard(B)
    grad0 = _first_grad(model, trainer, tokens, eps_all, coins, lo, hi)
    losses = _run_steps(model, trainer, tokens, eps_all, coins, lo, hi)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), flat=model.flat.cpu().numpy(), losses=np.array(losses), grad0=grad0)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_match_single_process_global_batch(tmp_path):
    port = 29100 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    r1 = np.load(tmp_path / "r1.npz")
    assert np.array_equal(r0["flat"], r1["flat"])                 # ranks stay bit-identical over 3 optimizer steps
    assert np.array_equal(r0["grad0"], r1["grad0"])
    # mean of the two half-batch gradients == gradient of the global batch in one process (compared before Adam, which
    # would turn rounding noise on near-zero entries into lr-sized differences)
    model, trainer = _build()
    tokens, eps_all, coins = _inputs()
    ref = _first_grad(model, trainer, tokens, eps_all, coins, 0, B)
    err = np.abs(ref - r0["grad0"]).max() / np.abs(ref).max()
    assert err < 2e-5, err
    # and the trajectories agree to the extent Adam allows
    single = _run_steps(model, trainer, tokens, eps_all, coins, 0, B)
    assert np.isfinite(r0["losses"]).all() and abs(single[0] - 0.5 * (r0["losses"][0] + r1["losses"][0])) < 1e-4 * abs(single[0])


# ---- LatentRNNTrainer under data parallelism (BASELINE.json configs[3], per-rank shape scaled down) -------------------
LB = 4                                      # global batch of sequences; 2 per rank


def _build_latent():
    from inpaintnet_amd import synthetic
    from inpaintnet_amd.latent_rnn import LatentRNN
    from inpaintnet_amd.latent_rnn_trainer import LatentRNNTrainer
    from inpaintnet_amd.measure_vae import MeasureVAE
    ds = synthetic.SyntheticFolkDataset(num_notes=CFG["V"])
    vae = MeasureVAE(ds, note_embedding_dim=CFG["E"], encoder_hidden_size=CFG["H"], latent_space_dim=CFG["Z"],
                     decoder_hidden_size=CFG["H"], encoder_dropout_prob=0.0, decoder_dropout_prob=0.0)
    vae.load_state_dict({k: torch.from_numpy(synthetic.det_param(k, tuple(v.shape))) for k, v in vae.state_dict().items()})
    model = LatentRNN(ds, vae, num_rnn_layers=2, rnn_hidden_size=CFG["H"], dropout=0.0, rnn_class=torch.nn.GRU,
                      auto_reg=False, teacher_forcing=True)
    for k, v in model.named_parameters():
        v.copy_(torch.from_numpy(synthetic.det_param(k, tuple(v.shape))))
    trainer = LatentRNNTrainer(ds, model, lr=1e-3)
    trainer.overlap_backward = True
    model.train()
    return model, trainer


def _latent_inputs():
    from inpaintnet_amd import synthetic
    score = torch.from_numpy(synthetic.folk_score(LB, CFG["V"], seed=17))
    eps = [torch.from_numpy(synthetic.det_normal(f"dpgpu/leps{i}", (LB, n, CFG["Z"]))) for i, n in enumerate((6, 6, 4))]
    return score, eps


def _latent_first_grad(model, trainer, score, eps, lo, hi):
    from inpaintnet_amd import dp, ops
    from inpaintnet_amd.latent_rnn_trainer import LatentRNNTrainer
    past, future, target = LatentRNNTrainer.split_score(score[lo:hi], 6, 6, 4, 24)
    trainer.zero_grad()
    w, s, gz = model(past, future, target, 4, train=True, eps=tuple(e[lo:hi].cuda() for e in eps))
    loss, acc = trainer.mean_crossentropy_loss_and_accuracy(w, target)
    loss.backward()
    ops.side_join()
    scale = dp.allreduce_grads(model.grad)
    g = (model.grad * scale).cpu().numpy()
    ops.side_defer(False)
    return g, float(loss.detach())


def _latent_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from inpaintnet_amd import dp
    assert dp.init_from_env(backend="gloo") == world
    torch.cuda.set_device(0)
    model, trainer = _build_latent()
    dp.broadcast_params(model.flat)
    score, eps = _latent_inputs()
    lo, hi = dp.shard(LB)
    grad0, loss0 = _latent_first_grad(model, trainer, score, eps, lo, hi)
    # two real optimizer steps through Trainer.step(): the generator bucket starts inside backward, the rest in step()
    from inpaintnet_amd.latent_rnn_trainer import LatentRNNTrainer
    past, future, target = LatentRNNTrainer.split_score(score[lo:hi], 6, 6, 4, 24)
    for _ in range(2):
        trainer.zero_grad()
        w, s, gz = model(past, future, target, 4, train=True, eps=tuple(e[lo:hi].cuda() for e in eps))
        loss, acc = trainer.mean_crossentropy_loss_and_accuracy(w, target)
        loss.backward()
        trainer.step()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"l{rank}.npz"), flat=model.flat.cpu().numpy(), grad0=grad0, loss0=loss0)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_latent_trainer_two_ranks_match_single_process_global_batch(tmp_path):
    port = 29400 + (os.getpid() % 2000)
    mp.spawn(_latent_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "l0.npz"), np.load(tmp_path / "l1.npz")
    assert np.array_equal(r0["flat"], r1["flat"])                 # ranks stay bit-identical through two Adam steps
    assert np.array_equal(r0["grad0"], r1["grad0"])
    model, trainer = _build_latent()
    score, eps = _latent_inputs()
    ref, loss = _latent_first_grad(model, trainer, score, eps, 0, LB)
    err = np.abs(ref - r0["grad0"]).max() / np.abs(ref).max()
    assert err < 2e-5, err
    assert abs(loss - 0.5 * (float(r0["loss0"]) + float(r1["loss0"]))) < 1e-5 * abs(loss)


# ---- a chain-kernel timeout on ONE of two ranks ------------------------------------------------
NB = 6                                      # global batches of 4 sequences x 16 bars: 32 measures per rank and step


def _fault_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from inpaintnet_amd import dp, ops, synthetic
    from inpaintnet_amd.measure_vae import MeasureVAE
    from inpaintnet_amd.vae_trainer import VAETrainer
    assert dp.init_from_env(backend="gloo") == world
    torch.cuda.set_device(0)

    class FaultyTrainer(VAETrainer):
        seen = 0

        def process_batch_data(self, batch):
            if self.seen == 1 and dp.rank() == 1 and self.chain_fallbacks == 0:
                ops.set_option(6, 1)        # rank 1 only: the next forward chain launch loses a workgroup (~0.4 s spin)
            self.seen += 1
            return super().process_batch_data(batch)

    ds = synthetic.SyntheticFolkDataset(num_notes=48)
    ds.n_bars = 16
    model = MeasureVAE(ds)                  # reference defaults (H = 512): the chain kernels run
    trainer = FaultyTrainer(ds, model, lr=1e-3)
    trainer.report_lag = 2                  # (default 12: the host stays far ahead; 2 puts the detection in the middle of this short epoch)
    dp.seed_shared(3)
    dp.seed_rank(3)
    dp.broadcast_params(model.flat)
    model.train()
    start = model.flat.clone()
    score, md = synthetic.SyntheticFolkDataset(num_notes=48, n_seq=4 * NB, seed=2).tensors()
    loader = [(torch.from_numpy(score[4 * i:4 * i + 4]), torch.from_numpy(md[4 * i:4 * i + 4])) for i in range(NB)]
    assert ops.chain_status(reset=True) >= 0
    try:
        loss, acc = trainer.loss_and_acc_on_epoch(loader, 0, train=True)
        torch.cuda.synchronize()
        np.savez(os.path.join(out_dir, f"f{rank}.npz"), flat=model.flat.cpu().numpy(), loss=loss, acc=acc,
                 adam_t=trainer.adam_t, fallbacks=trainer.chain_fallbacks, lost=trainer.lost_steps, seen=trainer.seen,
                 own_timeouts=ops.chain_status(), moved=float((model.flat - start).abs().max()),
                 m=trainer.adam_m.cpu().numpy())
    finally:
        ops.set_option(6, 0)
        ops.set_option(4, 1)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_chain_timeout_on_one_rank_is_handled_by_all_ranks_together(tmp_path):
    """inet_set_option(6, 1) on rank 1 only, in the middle of an epoch.  The failing rank's flag travels with the gradients, so
    BOTH ranks' optimizer kernels skip that step and the next ones, both read the same step reports at the same step, both
    fall back to the per-step kernels and run the same lost batches again: the replicas end bit-identical, with the same
    number of applied updates (= batches), and nobody is left waiting in a collective."""
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(_fault_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "f0.npz"), np.load(tmp_path / "f1.npz")
    assert np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["m"], r1["m"])
    assert np.isfinite(r0["flat"]).all() and float(r0["moved"]) > 0
    assert int(r0["adam_t"]) == int(r1["adam_t"]) == NB                       # every batch applied exactly once
    assert int(r0["fallbacks"]) == int(r1["fallbacks"]) == 1
    assert int(r0["lost"]) == int(r1["lost"]) == 3                            # report_lag + 1 steps were skipped and repeated
    assert int(r0["seen"]) == int(r1["seen"]) == NB
    assert float(r0["loss"]) == float(r1["loss"]) and np.isfinite(float(r0["loss"]))   # summed over ranks: identical means


def test_bench_two_ranks_functional(tmp_path):
    """`bench.py --gpus 2` end to end on ONE GPU (INET_BENCH_SHARE_GPU=1: both ranks on cuda:0, gloo carries the exchange): the
    N > 1 code path the driver's scaling run takes -- rank spawn, world check, seeded shared draws, bucketed exchange with the
    step flag, collective health verdict, exchange report, the LatentRNN step under the same exchange (BASELINE.json
    configs[3]) -- prints one JSON line with the data-parallel fields.  A functional check, not a scaling number."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    detail = str(tmp_path / "bench_detail.json")
    env = dict(os.environ, INET_BENCH_SHARE_GPU="1", INET_BENCH_DETAIL=detail)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "3",
                        "--no-cpu-baseline", "--no-roofline", "--no-parity"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    assert len(line[0]) < 4096                                  # the stdout line stays under bench.py's limit with the dp fields on it
    d = json.loads(line[0])
    for k in ("n_gpus", "per_rank_units_per_s", "allreduce_ms_per_step", "allreduce_mbytes", "dp", "scaling", "value"):
        assert k in d, k
    assert d["dp"]["chain_timeouts_per_rank"] == [0, 0] and d["extras"]["latent_dp_ms"] > 0
    d = json.load(open(detail))                                 # ... and everything else is in the detail file
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 512
    assert d["chain_timeouts"] == 0 and d["dp"]["chain_timeouts_per_rank"] == [0, 0] and d["dp"]["skipped_steps_per_rank"] == [0, 0]
    kinds = [x["kind"] for x in d["dp"]["ranges"]]
    assert "bucket" in kinds and kinds.count("final") >= 1
    # the whole arena (+ the 4-float head that carries the step flag) is exchanged exactly once per step
    assert abs(sum(x["mbytes"] for x in d["dp"]["ranges"]) - (d["allreduce_mbytes"] + 16e-6)) < 0.05
    assert d["dp"]["ms_per_step_without_exchange"] > 0 and len(d["per_rank_units_per_s"]) == 2
    lat = d["extras"]["latent_rnn_train_dp"]
    assert lat["global_batch"] == 256 and lat["sequences_per_s"] > 0
    assert "FUNCTIONAL CHECK ONLY" in d["data"]
