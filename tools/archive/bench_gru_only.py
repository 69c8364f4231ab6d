#!/usr/bin/env python3
"""A few launches of the fused GRU step kernel at the training shape (for rocprofv3 passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from inpaintnet_amd import ops  # noqa: E402

B, H = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 512
save = len(sys.argv) > 3 and sys.argv[3] == "save"
gi = torch.randn(B, 3 * H, device="cuda")
h = torch.randn(B, H, device="cuda")
W = torch.randn(3 * H, H, device="cuda") / H ** 0.5
b = torch.randn(3 * H, device="cuda")
for _ in range(50):
    ops.gru_step(gi, h, W, b, save=save)
torch.cuda.synchronize()
