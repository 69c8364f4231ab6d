"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no distributed code; BASELINE.json's config 4 asks for data-parallel
training with a gradient all-reduce.  Samples are independent through forward/backward and every loss is a
mean over rows, so with equal per-rank batches the global-batch gradient is the mean of the rank gradients:
ONE all-reduce(sum) of the flat fp32 gradient arena per step, the 1/world folded into the Adam kernel.
What has to be identical on every rank, because the reference draws it once per (global) batch:
the teacher-forcing coin (decoder.py:432, latent_rnn.py:143) and the past/target/future split
(latent_rnn_trainer.py:99-117) -- both come from host generators seeded identically on all ranks
(seed_shared); dropout masks and eps are per-rank streams (measure_vae.set_dropout_seed, torch.manual_seed).
"""
import os
import random

import torch


def is_distributed():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def world_size():
    return torch.distributed.get_world_size() if is_distributed() else 1


def rank():
    return torch.distributed.get_rank() if is_distributed() else 0


def init_from_env(backend=None):
    """Initialise the process group from RANK / WORLD_SIZE / MASTER_* (as set by torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or is_distributed():
        return world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    torch.distributed.init_process_group(backend=backend, **kw)
    return world


def seed_shared(seed):
    """Host generators that must agree on all ranks: Python's `random` (teacher-forcing coins) and torch's CPU
    generator (the stochastic past/target/future split)."""
    random.seed(seed)
    torch.manual_seed(seed)


def broadcast_params(flat, src=0):
    """Identical initial weights on every rank."""
    if world_size() > 1:
        torch.distributed.broadcast(flat, src=src)


_pending = []        # [(start, stop, work handle)] buckets of the arena already being summed


def start_bucket(grad, start, stop):
    """Begin summing grad[start:stop] over ranks asynchronously (RCCL runs on its own stream, ordered after the
    work already queued on the current stream).  Called as soon as a contiguous part of the arena is final --
    the decoder's gradients are complete while the encoder is still back-propagating -- so that part of the
    exchange hides behind the rest of backward.  No-op for a single process."""
    if world_size() > 1 and stop > start:
        work = torch.distributed.all_reduce(grad[start:stop], op=torch.distributed.ReduceOp.SUM, async_op=True)
        _pending.append((start, stop, work))


def allreduce_grads(grad):
    """Sum the (rest of the) flat gradient arena over ranks and wait for the buckets started earlier; returns the
    scale (1/world) the optimizer kernel applies to it."""
    w = world_size()
    if w > 1:
        done = sorted((a, b) for a, b, _ in _pending)
        pos = 0
        for a, b in done + [(grad.numel(), grad.numel())]:
            if a > pos:
                torch.distributed.all_reduce(grad[pos:a], op=torch.distributed.ReduceOp.SUM)
            pos = max(pos, b)
        for _, _, work in _pending:
            work.wait()
    _pending.clear()
    return 1.0 / w


def shard(n_items, r=None, w=None):
    """Contiguous shard [lo, hi) of n_items for this rank (equal sizes: drop the remainder)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    per = n_items // w
    return r * per, (r + 1) * per
