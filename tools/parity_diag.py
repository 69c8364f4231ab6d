"""Diagnostic: per-tensor gradient error of the HIP step and of the fp32 CPU oracle, both against a float64 run of the
oracle, (a) on trained weights for the MeasureVAE at B=256 (free-running), (b) for the LatentRNN at B=128.
    python tools/parity_diag.py [train_steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import torch_ref as O
from inpaintnet_amd import ops, dp


def vae_case(train_steps):
    dev = torch.device("cuda", 0)
    dp.seed_rank(1234, 0); dp.seed_shared(4321)
    wl = bench.VaeWorkload(dev, 0)
    for _ in range(train_steps):
        wl.step()
    model, trainer, tok_dev = wl.model, wl.trainer, wl.tokens
    P = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tok = tok_dev.cpu(); B = tok.shape[0]
    eps = torch.randn(B, 256, generator=torch.Generator().manual_seed(7))
    real = ops.dropout_mask
    for tf in (True, False):
        rec = []
        ops.dropout_mask = lambda *a, _r=rec: (_r.append(real(*a)) or _r[-1])
        trainer.zero_grad()
        w, s, zd, pd, z, zp = model(tok_dev, train=True, eps=eps.to(dev), teacher_forced=tf)
        ce, acc = trainer.mean_crossentropy_loss_and_accuracy(w, tok_dev)
        (ce + trainer.compute_kld_loss(zd, pd)).backward()
        ops.side_defer(False)
        ops.dropout_mask = real
        masks = [m.cpu().permute(1, 0, 2) for m in rec]
        res = {}
        for dt in (torch.float32, torch.float64):
            Pr = {k: v.to(dt).clone().requires_grad_(True) for k, v in P.items()}
            om = {"enc": masks[0].to(dt), "beat": masks[1].to(dt), "tick": masks[2].to(dt)}
            wr, sr, mu, ls, zr = O.vae_forward(Pr, tok, eps.to(dt), tf, om, feed_tokens=None if tf else s.cpu()[:, 0])
            lr, *_ = O.vae_loss(wr, tok, mu, ls)
            lr.backward()
            res[dt] = {k: Pr[k].grad.double() for k in P}
        print(f"--- VAE tf={tf} after {train_steps} training steps: tensor, |g|max, hip-vs-f64, cpu32-vs-f64")
        rows = []
        for k in P:
            t = res[torch.float64][k]; m = float(t.abs().max()) + 1e-30
            eh = float((model.param_grad(k).cpu().double() - t).abs().max()) / m
            ec = float((res[torch.float32][k] - t).abs().max()) / m
            rows.append((eh, ec, m, k))
        for eh, ec, m, k in sorted(rows, reverse=True)[:8]:
            print(f"  {k:55s} {m:10.3e} {eh:10.3e} {ec:10.3e}")


if __name__ == "__main__":
    vae_case(int(sys.argv[1]) if len(sys.argv) > 1 else 420)
