"""bigru2 forward + backward of a given shape under the chain generations (inet_set_option key 7) against a float64 CPU run:
    python tools/bigru2_vs_float64.py B T K H   -- how far f32-input and bf16-piece arithmetic sit from float64 and from each other"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inpaintnet_amd import ops, layout
B, T, K, H = [int(a) for a in sys.argv[1:5]]
g = torch.Generator().manual_seed(B * 100 + T * 10 + K)
shapes = layout._gru("g", K, H, 2, True)
offs, total = layout.arena_offsets(dict(shapes))
P = {k: (torch.randn(*s, generator=g) * (0.3 if "weight" in k else 0.1)) for k, s in shapes}
flat = torch.zeros(total)
for k, (off, s) in offs.items():
    flat[off:off + P[k].numel()] = P[k].reshape(-1)
flat = flat.cuda()
h0 = torch.randn(4, B, H, generator=g).cuda()
mask = ((torch.rand(T, B, 2 * H, generator=g) > 0.5).float() * 2.0).cuda()
x = torch.randn(B, T, K, generator=g).cuda()
from oracle import torch_ref as O
P64 = {k: v.double() for k, v in P.items()}
out64, hn64 = O.gru_stack(x.cpu().double(), h0.cpu().double(), P64, "g", 2, True, [mask.cpu().double().permute(1, 0, 2)])
out32, hn32 = O.gru_stack(x.cpu(), h0.cpu(), P, "g", 2, True, [mask.cpu().permute(1, 0, 2)])
print("cpu fp32 oracle vs float64:", float((out32.double() - out64).abs().max()))
res = {}
for mode in (0, 9):
    ops.set_option(7, mode)
    o, h, ws = ops.bigru2_fwd(x, None, flat, H, B, T, K, h0=h0, mask=mask, save=True)
    torch.cuda.synchronize()
    res[mode] = (o.cpu(), h.cpu())
    print("mode", mode, "vs float64:", float((o.cpu().double() - out64).abs().max()), " vs cpu fp32:", float((o.cpu() - out32).abs().max()))
d = (res[0][0] - res[9][0]).abs()
print("out diff max", float(d.max()), "per row block:", [round(float(d[i:i+16].max()), 6) for i in range(0, B, 16)])
print("per t:", [round(float(d[:, t].max()), 6) for t in range(T)])
print("fwd dir / bwd dir:", float(d[..., :H].max()), float(d[..., H:].max()))
dh = (res[0][1] - res[9][1]).abs()
print("hn diff", [float(dh[i].max()) for i in range(4)])
