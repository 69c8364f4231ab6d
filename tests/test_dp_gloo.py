"""Data-parallel path, world_size 2, gloo on CPU (no GPU needed): the exchange the trainer performs
(inpaintnet_amd.dp: one all-reduce of the flat gradient arena, 1/world folded into Adam, shared coins/splits)
reproduces the single-process global-batch step.  Compute stand-in for the HIP kernels: the CPU oracle."""
import os
import random

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from inpaintnet_amd import dp, layout, synthetic
from oracle import torch_ref as O

CFG = dict(V=12, E=4, H=16, Z=8)


def _params():
    shapes = layout.vae_param_shapes(CFG["V"], CFG["E"], CFG["H"], CFG["Z"], CFG["H"])
    return shapes, {k: torch.from_numpy(synthetic.det_param(k, s)) for k, s in shapes.items()}


def _flat_grads(P, shapes, tokens, eps, coin):
    for p in P.values():
        p.requires_grad_(True)
        p.grad = None
    w, s, mu, ls, z = O.vae_forward(P, tokens, eps, coin)
    loss, ce, kl, acc = O.vae_loss(w, tokens, mu, ls)
    loss.backward()
    offs, total = layout.arena_offsets(shapes)
    flat = torch.zeros(total)
    for k, (off, shp) in offs.items():
        flat[off:off + P[k].numel()] = P[k].grad.reshape(-1)
    return flat, float(loss)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    assert dp.init_from_env(backend="gloo") == world
    assert dp.world_size() == world and dp.rank() == rank
    dp.seed_shared(77)
    coins = [random.random() < 0.5 for _ in range(6)]
    split = [int(torch.randint(2, 7, (1,)).item()) for _ in range(6)]
    shapes, P = _params()
    B = 8
    tokens = torch.from_numpy(synthetic.det_tokens("dp/tokens", (B, 24), CFG["V"]))
    eps = torch.from_numpy(synthetic.det_normal("dp/eps", (B, CFG["Z"])))
    lo, hi = dp.shard(B)
    flat, loss = _flat_grads(P, shapes, tokens[lo:hi], eps[lo:hi], coins[0])
    if rank == 1:
        flat_b = torch.full_like(flat, 123.0)          # broadcast must overwrite this
    else:
        flat_b = flat.clone()
    dp.broadcast_params(flat_b, src=0)
    # bucketed exchange: the tail of the arena starts early (as the decoder's gradients do), the rest at step()
    cut = (flat.numel() // 3) // 4 * 4
    dp.start_bucket(flat, cut, flat.numel())
    gscale = dp.allreduce_grads(flat)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), grad=(flat * gscale).numpy(), coins=np.array(coins),
             split=np.array(split), loss=loss, bcast=flat_b.numpy(), lo=lo, hi=hi)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_allreduce_equals_global_batch_step(tmp_path):
    port = 29000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "r0.npz")
    r1 = np.load(tmp_path / "r1.npz")
    # shared host draws: same coin and split sequence on both ranks
    assert np.array_equal(r0["coins"], r1["coins"]) and np.array_equal(r0["split"], r1["split"])
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 4, 4, 8)
    # every rank ends with the same averaged gradient ...
    assert np.array_equal(r0["grad"], r1["grad"])
    assert np.array_equal(r0["bcast"], r1["bcast"])
    # ... equal to the gradient of the global batch computed in one process
    shapes, P = _params()
    tokens = torch.from_numpy(synthetic.det_tokens("dp/tokens", (8, 24), CFG["V"]))
    eps = torch.from_numpy(synthetic.det_normal("dp/eps", (8, CFG["Z"])))
    full, loss = _flat_grads(P, shapes, tokens, eps, bool(r0["coins"][0]))
    assert np.abs(r0["grad"] - full.numpy()).max() <= 1e-5 * np.abs(full.numpy()).max() + 1e-8
    assert abs(0.5 * (float(r0["loss"]) + float(r1["loss"])) - loss) < 1e-5 * abs(loss)


def test_single_process_defaults():
    assert dp.world_size() == 1 and dp.rank() == 0
    g = torch.ones(8)
    assert dp.allreduce_grads(g) == 1.0 and torch.equal(g, torch.ones(8))
    assert dp.shard(10, 1, 4) == (2, 4)


# ---- epoch loop under data parallelism: every rank must see the same statistics and stop on the same epoch ----------
class _StubModel:
    """Host-only stand-in: the epoch loop's control flow (feed sharding, statistic reduction, early stopping, save on
    rank 0 only) is what is under test here, not the kernels."""

    def __init__(self, out_dir):
        self.flat = torch.zeros(8)
        self.grad = torch.zeros(8)
        self.filepath = os.path.join(out_dir, f"model_r{dp.rank()}")
        self.saved = 0

    def train(self, mode=True):
        return self

    def zero_grad(self):
        self.grad.zero_()

    def save(self):
        self.saved += 1

    def save_checkpoint(self, epoch):
        pass


def _epoch_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    assert dp.init_from_env(backend="gloo") == world
    from inpaintnet_amd.trainer import Trainer

    class StubTrainer(Trainer):
        feed_fields = (0,)

        def __init__(self, dataset, model):
            self.seen = []
            super().__init__(dataset, model, early_stopping=True)
            self.early_stopper.patience = 2

        def process_batch_data(self, batch):
            return batch[0]

        def loss_and_acc_for_batch(self, batch, epoch_num=None, train=True):
            self.seen.append(batch[:, 0, 0].clone())
            # a loss that differs between ranks (it depends on the shard) and stops improving after epoch 1
            base = 1.0 if epoch_num < 2 else 2.0
            return (batch.float().mean() * 1e-3 + base - 0.1 * min(epoch_num, 1)).requires_grad_(train), torch.tensor(0.5)

        def zero_grad(self):                 # (the real one also arms the HIP side stream)
            self.model.zero_grad()

        def step(self):
            self.adam_t += 1

        def update_scheduler(self, epoch_num):
            return

        def save_training_state(self, path, next_epoch=0):
            pass

    ds = synthetic.SyntheticFolkDataset(num_notes=12, n_seq=40)
    model = _StubModel(out_dir)
    tr = StubTrainer(ds, model)
    tr.train_model(batch_size=8, num_epochs=10, seed=5)
    seen = torch.stack(tr.seen).numpy()
    np.savez(os.path.join(out_dir, f"e{rank}.npz"), seen=seen, saved=model.saved, steps=tr.adam_t,
             counter=tr.early_stopper.counter, stop=tr.early_stopper.early_stop)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_epoch_loop_ranks_agree_on_statistics_and_early_stop(tmp_path):
    port = 29300 + (os.getpid() % 2000)
    mp.spawn(_epoch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "e0.npz"), np.load(tmp_path / "e1.npz")
    # both ranks ran the same number of optimizer steps and stopped on the same epoch (no rank is left hanging in an
    # all-reduce): epochs 0,1 improve, 2 and 3 do not -> stop after epoch 3 with patience 2
    assert int(r0["steps"]) == int(r1["steps"]) == 4 * 3
    assert bool(r0["stop"]) and bool(r1["stop"]) and int(r0["counter"]) == int(r1["counter"]) == 2
    assert int(r0["saved"]) == 4 and int(r1["saved"]) == 0          # only rank 0 writes checkpoints
    # each rank saw its own half of every global batch: same shuffle on both ranks, disjoint rows
    assert r0["seen"].shape == r1["seen"].shape and r0["seen"].shape[1] == 4
    assert not np.array_equal(r0["seen"], r1["seen"])
    full_score, _ = synthetic.SyntheticFolkDataset(num_notes=12, n_seq=40).tensors()
    first_tokens = set(full_score[:28, 0, 0].tolist())
    assert set(r0["seen"][0].tolist()) <= first_tokens


# ---- the rank-coordinated failure protocol of Trainer.step / check_steps / _run_batch on host stand-ins (round 4) --------------
NBATCH = 7


def _protocol_worker(rank, world, port, out_dir, mode="timeout"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    assert dp.init_from_env(backend="gloo") == world
    from inpaintnet_amd.trainer import Trainer

    class Arena:                                     # the layout of model.Model: [4-float head | arena | tail], flag = head[0]
        _HEAD = 4

        def __init__(self):
            self.flat = torch.zeros(8)
            self._grad_store = torch.zeros(4 + 8 + 4)
            self.grad = self._grad_store[4:12]
            self.step_flag = self._grad_store[0:2]

        def train(self, mode=True):
            return self

        def zero_grad(self):
            self._grad_store.zero_()

    class ProtoTrainer(Trainer):
        """The real step() / check_steps() / _run_batch() / _replay() / _fall_back_from_chains(); the five device-touching hooks
        replaced by host code: a 'chain timeout' is a sticky local flag, the 'optimizer kernel' an SGD update that honours the
        SUMMED flag, the 'step report' a dict."""
        feed_fields = (0,)

        def __init__(self, dataset, model):
            super().__init__(dataset, model)
            self.report_lag = 2
            self.local_fail = False
            self.local_badtok = False
            self.chains_off = False
            self.records, self.applied, self.seen, self.tokrec = {}, [], 0, {}

        def process_batch_data(self, batch):
            if self.seen == 2 and dp.rank() == 1 and not self.chains_off and mode == "timeout":
                self.local_fail = True               # rank 1 only: "a persistent kernel gave up" while batch 2 is computed
            if self.seen == 2 and dp.rank() == 1 and mode == "token":
                self.local_badtok = True             # rank 1 only: "a prologue kernel met a token outside the vocabulary"
            self.seen += 1
            return batch[0]

        def loss_and_acc_for_batch(self, batch, epoch_num=None, train=True):
            self.model.grad.copy_(batch.float().mean().expand(8))      # (as if the backward kernels had filled the arena)
            self.current = int(batch[0, 0, 0])
            return (torch.zeros(1, requires_grad=True) + float(batch.float().mean())).sum(), torch.tensor(0.5)

        def zero_grad(self):
            self.model.zero_grad()

        def update_scheduler(self, epoch_num):
            return

        def _reports_on(self):
            return True

        def _join_side_work(self):
            pass

        def _export_flag(self, flag):
            flag[0] = 1.0 if self.local_fail else 0.0
            flag[1] = 1.0 if self.local_badtok else 0.0

        def _launch_optimizer(self, tag, gscale, flag):
            skipped = float(flag[0]) != 0.0          # the ranks' SUM: identical everywhere
            self.records[tag] = skipped
            self.tokrec[tag] = float(flag[1]) != 0.0
            if not skipped:
                self.model.flat -= 0.1 * gscale * self.model.grad
                self.applied.append(self.current)

        def _read_report(self, tag):
            return True, self.records[tag], False, self.tokrec[tag]

        def _device_sync(self):
            pass

        def _chains_off(self):
            self.local_fail, self.chains_off = False, True
            return 1

    model = Arena()
    tr = ProtoTrainer(None, model)
    # global batches of 4 rows x [1, 4]; row value = batch id, so a rank's shard mean identifies the batch
    loader = [(torch.full((4, 1, 4), i, dtype=torch.int32), torch.zeros(4, 1, 4, dtype=torch.int32)) for i in range(NBATCH)]
    if mode == "token":
        from inpaintnet_amd import ops
        raised_at = -1
        try:
            tr.loss_and_acc_on_epoch(loader, 0, train=True)
        except ops.TokenRangeError:
            raised_at = tr._tag                      # how many optimizer launches this rank had issued when it raised
        np.savez(os.path.join(out_dir, f"p{rank}.npz"), raised_at=raised_at, applied=np.array(tr.applied))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        return
    loss, acc = tr.loss_and_acc_on_epoch(loader, 0, train=True)
    np.savez(os.path.join(out_dir, f"p{rank}.npz"), flat=model.flat.numpy(), applied=np.array(tr.applied), adam_t=tr.adam_t,
             fallbacks=tr.chain_fallbacks, lost=tr.lost_steps, loss=loss, seen=tr.seen,
             skipped=np.array(sorted(t for t, s in tr.records.items() if s)))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_failure_protocol_keeps_ranks_in_step(tmp_path):
    """Rank 1 'times out' while batch 2 is computed.  The flag travels in the head of the gradient store through the SAME
    all-reduce as the gradients, so both ranks' optimizers skip step 2 -- and 3 and 4, the flag being sticky until the fallback --,
    both read the report of step 2 when step 4 has been queued (report_lag = 2), both fall back and run batches 2, 3, 4 again, in
    order; every batch is applied exactly once, weights are bit-identical, the epoch mean counts every batch once."""
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_protocol_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "p0.npz"), np.load(tmp_path / "p1.npz")
    for r in (r0, r1):
        assert r["applied"].tolist() == list(range(NBATCH))
        assert r["skipped"].tolist() == [2, 3, 4]
        assert int(r["adam_t"]) == NBATCH and int(r["fallbacks"]) == 1 and int(r["lost"]) == 3
    assert np.array_equal(r0["flat"], r1["flat"]) and float(r0["loss"]) == float(r1["loss"])
    # the update is the mean of the two ranks' gradients (= the batch id): sum_i 0.1 * i
    assert np.allclose(r0["flat"], -0.1 * sum(range(NBATCH)), rtol=1e-6)
    assert abs(float(r0["loss"]) - np.mean(range(NBATCH))) < 1e-6     # every batch in the epoch mean exactly once


def test_bad_token_on_one_rank_raises_on_all_ranks_at_the_same_step(tmp_path):
    """the token-range error (decoder.py:36-45 check_index) used to be read from a rank-local counter, so only the
    rank that saw the token raised and the others waited in the next all-reduce.  The token word now travels in the head of the
    gradient store next to the step flag (word 1) and comes back in the step report: rank 1 'sees' a bad token in batch 2, BOTH
    ranks raise TokenRangeError when they read that report -- after the same number of optimizer launches."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_protocol_worker, args=(2, port, str(tmp_path), "token"), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "p0.npz"), np.load(tmp_path / "p1.npz")
    assert int(r0["raised_at"]) == int(r1["raised_at"]) == 2 + 2 + 1      # report of step 2 read behind step 4 (report_lag 2)
    assert r0["applied"].tolist() == r1["applied"].tolist()
