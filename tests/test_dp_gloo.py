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
