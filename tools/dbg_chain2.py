import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inpaintnet_amd import ops
torch.manual_seed(0)
B, T, K, H = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 3, 32, 256
n_w = 0
shapes = []
for l in range(2):
    for d in range(2):
        kin = K if l == 0 else 2 * H
        shapes += [(3 * H, kin), (3 * H, H), (3 * H,), (3 * H,)]
tot = sum(int(torch.tensor(s).prod()) for s in shapes)
tot = (tot + 3) // 4 * 4
W = (torch.randn(tot) * 0.05).cuda()
x = torch.randn(B, T, K).cuda()
dout = torch.randn(B, T, 2 * H).cuda()
dhn = torch.randn(4, B, H).cuda()
res = {}
for mode in (0, 9):
    ops.set_option(7, mode)
    out, hn, ws = ops.bigru2_fwd(x, None, W, H, B, T, K, save=True)
    g = torch.zeros_like(W)
    dx, dh0 = ops.bigru2_bwd(x, None, W, g, H, B, T, K, None, dout, dhn, ws, want_dx=True, want_dh0=True)
    ops.side_join(); torch.cuda.synchronize()
    res[mode] = (out.cpu(), hn.cpu(), dx.cpu(), dh0.cpu(), g.cpu())
    print("mode", mode, "status", ops.chain_status())
names = ["out", "hn", "dx", "dh0", "grads"]
for n, a, b in zip(names, res[0], res[9]):
    print(n, "max|v1|", float(a.abs().max()), "max diff", float((a - b).abs().max()))
g0, g9 = res[0][4], res[9][4]
off = 0
for i, s in enumerate(shapes):
    n = int(torch.tensor(s).prod())
    print(i, s, float((g0[off:off+n]-g9[off:off+n]).abs().max()), float(g0[off:off+n].abs().max()))
    off += n
