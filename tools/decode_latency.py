"""eval-mode decode latency at small batch (bench.decode_latency_extra): python tools/decode_latency.py"""
import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
sys.stdout = sys.stderr
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
r = bench.decode_latency_extra(wl.model, iters=200)["decoder_eval"]
print({k: v["ms_per_call"] for k, v in r.items() if isinstance(v, dict)})
# b = 1 under the decode kernels (inet_set_option key 15: 0 = decode_chain.hip's exchange kernel, 1 / 2 = decode_b1.hip behind the beat
# path's launches, 3 = decode_b1.hip with the beat path folded in)
from inpaintnet_amd import ops
import time
vae = wl.model
vae.eval()
z = torch.randn(1, vae.latent_space_dim, device="cuda")
dummy = torch.zeros(1, 24, device="cuda")
for mode in (0, 1, 2, 3, 4):
    ops.set_option(15, mode)
    with torch.no_grad():
        for _ in range(5):
            vae.decoder(z, dummy, train=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300):
            vae.decoder(z, dummy, train=False)
        torch.cuda.synchronize()
    print(f"b = 1, decode kernel mode {mode}: {1e3 * (time.perf_counter() - t0) / 300:.4f} ms per call, chain status {ops.chain_status()}")
ops.set_option(15, 4)

# two to four measures (the reference's non-auto-regressive inpainting call) on the register-resident kernel (mode 3) and on
# decode_chain.hip's exchange kernel (mode 0)
for b in (2, 3, 4, 5, 6, 8, 10, 11, 12, 16):
    z = torch.randn(b, vae.latent_space_dim, device="cuda")
    dummy = torch.zeros(b, 24, device="cuda")
    line = []
    for mode in (0, 3, 4):
        ops.set_option(15, mode)
        with torch.no_grad():
            for _ in range(5):
                vae.decoder(z, dummy, train=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                vae.decoder(z, dummy, train=False)
            torch.cuda.synchronize()
        line.append(f"mode {mode}: {1e3 * (time.perf_counter() - t0) / 200:.4f} ms")
    print(f"b = {b}: " + ", ".join(line) + f", chain status {ops.chain_status()}")
ops.set_option(15, 4)

if os.environ.get("INET_DECODE_B1_STAMPS") == "1":
    # anatomy of a tick from in-kernel wall-clock stamps (10 ns units), workgroup C and workgroup TBi_0
    from inpaintnet_amd import ops as _o
    for b in (1, 2, 4):
        z = torch.randn(b, vae.latent_space_dim, device="cuda")
        with torch.no_grad():
            for _ in range(3):
                w, s_, ws = _o.decoder_fwd(vae.cfg, z, None, False, vae.flat)
        torch.cuda.synchronize()
        st = _o.ws_field(vae.cfg, ws, b, 2, "b1stamps").cpu().view(torch.int64).view(2, 32, 8).double() * 0.01
        c, t = st[0, 2:23], st[1, 2:23]
        merged = b == 1 and vae.num_notes <= 64 or b == 2 and vae.num_notes <= 32       # decode_b1.hip's MG build: stamps of CB_0 only
        if merged:
            names = ["table rows + layer-0 cells (+ publish h0 for TA)", "barrier + W_ih1 product + layer-1 cell + publish h1",
                     "wait for h1 of all 16 workgroups (the tick's ONE hand-off)", "barrier + head product(s) + barrier",
                     "argmax", "look at the next tick's gh0"]
        else:
            names = ["table rows + cells + publish h0", "wait for h1 (TBi: product, cell, two hand-offs)", "barrier", "head product(s) + barrier",
                     "argmax + barrier", "look at the next tick's gh0"]
        print(f"b = {b}: {'CB_0' if merged else 'C'}, mean us per phase over ticks 2..22 (tick period {float((st[0, 3:24, 0] - st[0, 2:23, 0]).mean()):.2f} us)")
        for i, n in enumerate(names):
            print(f"  {float((c[:, i + 1] - c[:, i]).mean()):6.2f}  {n}")
        if merged:
            continue
        names_t = ["wait for gh1 (requested early)", "wait for h0", "barrier", "W_ih1 product(s) + cells + publish h1"]
        print("  TBi_0:")
        for i, n in enumerate(names_t):
            print(f"  {float((t[:, i + 1] - t[:, i]).mean()):6.2f}  {n}")
