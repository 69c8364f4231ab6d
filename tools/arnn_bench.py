"""AnticipationRNN training-step time with and without chain kernels (bench.py arnn_extra)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from inpaintnet_amd import ops
for chain in (1, 0, 1):
    ops.set_option(4, chain)
    t0 = time.time()
    r = bench.arnn_extra(steps=6, warmup=2)
    torch.cuda.synchronize()
    print("chain", chain, r["anticipation_rnn_train"]["ms_per_step"], "ms/step", "status", ops.chain_status(), "wall %.1fs" % (time.time() - t0), flush=True)
