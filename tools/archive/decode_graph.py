#!/usr/bin/env python3
"""b = 1 / 16 free-running decode as a captured hipGraph (torch.cuda.CUDAGraph) against the eager call: python tools/decode_graph.py"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
sys.stdout = sys.stderr
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
vae = wl.model
vae.eval()
for b in (1, 16):
    z = torch.randn(b, vae.latent_space_dim, device="cuda")
    dummy = torch.zeros(b, 24, device="cuda")
    with torch.no_grad():
        for _ in range(5): w0, s0 = vae.decoder(z, dummy, train=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): vae.decoder(z, dummy, train=False)
        torch.cuda.synchronize()
        eager = 1e3 * (time.perf_counter() - t0) / 200
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): vae.decoder(z, dummy, train=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                w1, s1 = vae.decoder(z, dummy, train=False)
        except Exception as e:
            print(f"b={b}: capture failed: {e!r}"); continue
        g.replay(); torch.cuda.synchronize()
        same = bool((s1 == s0).all())
        t0 = time.perf_counter()
        for _ in range(200): g.replay()
        torch.cuda.synchronize()
        graph = 1e3 * (time.perf_counter() - t0) / 200
    print(f"b={b}: eager {eager:.4f} ms  graph {graph:.4f} ms  tokens equal {same}")
