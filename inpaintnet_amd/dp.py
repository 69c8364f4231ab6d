"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no distributed code; BASELINE.json's config 4 asks for data-parallel
training with a gradient all-reduce.  Samples are independent through forward/backward and every loss is a
mean over rows, so with equal per-rank batches the global-batch gradient is the mean of the rank gradients:
ONE all-reduce(sum) of the flat fp32 gradient arena per step, the 1/world folded into the Adam kernel.

Random draws:
  * identical on every rank, because the reference draws them once per (global) batch: the teacher-forcing coin
    (decoder.py:432, latent_rnn.py:143 -- Python's `random`) and the past/target/future split
    (latent_rnn_trainer.py:99-117 -- torch's CPU generator).  seed_shared(seed) seeds exactly those two HOST
    generators and nothing else;
  * distinct per rank: eps (torch.randn_like on the device -> the CUDA generator) and the dropout masks (the
    counter-based stream of measure_vae.set_dropout_seed).  seed_rank(seed) seeds those with a rank offset and
    does not touch the host generators.
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
    if backend == "nccl":
        one_side_stream()
    return world


def seed_shared(seed):
    """Host generators that must agree on all ranks: Python's `random` (teacher-forcing coins) and torch's CPU
    generator (the stochastic past/target/future split, the constraint window).  The device generator is left alone."""
    random.seed(seed)
    torch.default_generator.manual_seed(seed)


def seed_rank(seed, r=None):
    """Per-rank noise: eps draws (device generator) and dropout masks (counter-based stream)."""
    from .measure_vae import set_dropout_seed
    r = rank() if r is None else r
    if torch.cuda.is_available():
        torch.cuda.manual_seed(int(seed) + r)
    set_dropout_seed(seed, r)


def one_side_stream():
    """One side stream instead of two when a gradient exchange runs beside the steps (inet_set_option key 13): the process then keeps
    caller, side, bucket and process-group streams busy and the runtime has four hardware queues to deal.  On one GPU, with a stand-in
    all-reduce that really runs on its own stream, the B = 256 step measured 4.94 ms with two side streams and 3.87 ms with one (3.65
    without any exchange; tools/dp_streams_1gpu.py, profiles/r04_c_arnn_xcd.txt).  INET_DP_SIDE_STREAMS=n overrides (0: all)."""
    if torch.cuda.is_available():
        from . import ops
        ops.set_option(13, int(os.environ.get("INET_DP_SIDE_STREAMS", "1")))


def broadcast_params(flat, src=0):
    """Identical initial weights on every rank (and, on the way, the library's stream budget for a process with an exchange)."""
    if world_size() > 1:
        if flat.is_cuda:
            one_side_stream()
        torch.distributed.broadcast(flat, src=src)


# Buckets of a gradient arena whose sum over ranks has already been started, keyed by the arena's data pointer so that
# two models in one process (or a stale entry after an exception) can never be mistaken for each other.
_pending = {}        # data_ptr -> [(start, stop, work handle)]


def reset_buckets(grad=None):
    """Forget started buckets (Trainer.zero_grad calls this: a step that died between backward and step() must not
    leave handles behind).  Outstanding work is waited for first so that no all-reduce is left writing the arena."""
    keys = list(_pending) if grad is None else [grad.data_ptr()]
    del last_ranges[:]
    for k in keys:
        for _, _, work in _pending.pop(k, []):
            work.wait()


_prep_streams = {}   # device index -> the stream the bucket all-reduces are issued from

# Measurement aids (bench.py --gpus N): the ranges of the last step's exchange, in floats of the allocation that was reduced
# ("bucket": started under the backward pass; "final": issued by allreduce_grads, what a step can expose), and a switch that
# turns every gradient collective into a no-op so that the same step can be timed without its exchange.
last_ranges = []
_exchange = True


def set_exchange(on):
    """False: start_bucket / allreduce_grads skip their collectives (timing aid; the gradients stay rank-local)."""
    global _exchange
    _exchange = bool(on)



def _prep_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _prep_streams:
        _prep_streams[key] = torch.cuda.Stream(device=device)
    return _prep_streams[key]


def start_bucket(grad, start, stop, join_side=False):
    """Begin summing grad[start:stop] over ranks asynchronously.  Called as soon as a contiguous part of the arena is
    final -- the decoder's gradients are complete while the encoder is still back-propagating -- so that part of the
    exchange hides behind the rest of backward.  The all-reduce is issued from a dedicated stream that waits for the work
    queued so far on the current stream and, with `join_side`, for the library's side streams (the leaf weight-gradient
    GEMMs that fill the bucket): the backward pass itself is not held up.  No-op for a single process.  A range may be
    started only once per step (a second backward() before step() would otherwise be summed twice)."""
    if world_size() > 1 and stop > start:
        last_ranges.append(("bucket", int(start), int(stop)))
        if not _exchange:
            return
        mine = _pending.setdefault(grad.data_ptr(), [])
        for a, b, _ in mine:
            if start < b and a < stop:
                raise RuntimeError(f"dp.start_bucket: [{start},{stop}) overlaps the bucket [{a},{b}) already being "
                                   "reduced for this arena (backward() twice without step()/zero_grad()?)")
        if grad.is_cuda:
            from . import ops
            prep = _prep_stream(grad.device)
            prep.wait_stream(torch.cuda.current_stream(grad.device))
            if join_side:
                ops.side_join_on(prep)
            with torch.cuda.stream(prep):
                work = torch.distributed.all_reduce(grad[start:stop], op=torch.distributed.ReduceOp.SUM, async_op=True)
        else:
            work = torch.distributed.all_reduce(grad[start:stop], op=torch.distributed.ReduceOp.SUM, async_op=True)
        mine.append((start, stop, work))


def allreduce_grads(grad, store=None, head=0):
    """Sum the (rest of the) flat gradient arena over ranks and wait for the buckets started earlier; returns the
    scale (1/world) the optimizer kernel applies to it.  `store` / `head`: the allocation the arena is a view of and the
    floats in front of the arena inside it (Model._grad_store, Model._HEAD: the step flag).  The head is never part of a
    bucket, so it always travels with the first range that is left for this call -- by default no extra collective (MeasureVAE:
    the layer-0 range at the start of the arena) -- and every rank's optimizer kernel sees the sum of all ranks' flags."""
    w = world_size()
    mine = _pending.pop(grad.data_ptr(), [])
    if w > 1:
        if store is None or head <= 0:
            store, head = grad, 0
        else:
            assert store.data_ptr() + 4 * head == grad.data_ptr(), "dp.allreduce_grads: grad is not store[head:]"
        pos = 0                                       # positions in `store`; buckets are in arena coordinates
        end = head + grad.numel()
        started = sorted((a + head, b + head) for kind, a, b in last_ranges if kind == "bucket") if not _exchange else \
            sorted((a + head, b + head) for a, b, _ in mine)
        for a, b in started + [(end, end)]:
            if a > pos:
                last_ranges.append(("final", int(pos), int(a)))
                if _exchange:
                    torch.distributed.all_reduce(store[pos:a], op=torch.distributed.ReduceOp.SUM)
            pos = max(pos, b)
        for _, _, work in mine:
            work.wait()
    return 1.0 / w


def allreduce_sum_(t):
    """In-place sum of a small tensor over ranks (epoch statistics)."""
    if world_size() > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.SUM)
    return t


def shard(n_items, r=None, w=None):
    """Contiguous shard [lo, hi) of n_items for this rank (equal sizes: drop the remainder)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    per = n_items // w
    return r * per, (r + 1) * per


def shard_batch(batch):
    """This rank's rows of a global batch (a tensor or a tuple/list of tensors sharing dim 0)."""
    if world_size() == 1:
        return batch
    if torch.is_tensor(batch):
        lo, hi = shard(batch.shape[0])
        return batch[lo:hi]
    return type(batch)(shard_batch(b) if torch.is_tensor(b) else b for b in batch)
