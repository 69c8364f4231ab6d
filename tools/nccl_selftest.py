#!/usr/bin/env python3
"""Single-rank RCCL self-test of the exact torch.distributed calls bench.py / dp.py make with N > 1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
torch.cuda.set_device(0)
torch.distributed.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
from inpaintnet_amd import dp  # noqa: E402

g = torch.ones(1024, device="cuda")
w = torch.distributed.all_reduce(g[256:], op=torch.distributed.ReduceOp.SUM, async_op=True)
torch.distributed.all_reduce(g[:256])
w.wait()
torch.distributed.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
dp.broadcast_params(g)
torch.cuda.synchronize()
print("nccl selftest ok", float(g.sum()), float(t))
torch.distributed.destroy_process_group()
