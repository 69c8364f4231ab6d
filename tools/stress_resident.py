"""Stress of the register-resident persistent launches that need most of the chip at once (decode teams: up to 245 workgroups of 512
threads x 256 registers = one per CU; AnticipationRNN's token pass with XCD-local stores): thousands of calls, interleaved with
other work on the stream, chain status checked throughout.  python3 tools/stress_resident.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from inpaintnet_amd import ops, synthetic
from inpaintnet_amd.measure_vae import MeasureVAE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
ds = synthetic.SyntheticFolkDataset(num_notes=48)
vae = MeasureVAE(ds)
vae.eval()
bad = 0
t0 = time.perf_counter()
filler = torch.randn(4096, 4096, device=dev)
for i in range(n):
    b = (1, 2, 3, 4, 5, 6, 8, 9, 10, 12, 16)[i % 11]
    z = torch.randn(b, vae.latent_space_dim, device=dev)
    with torch.no_grad():
        vae.decoder(z, torch.zeros(b, 24, device=dev), train=False)
        if i % 7 == 0:
            filler = filler @ filler * 1e-4          # something else on the stream between the persistent launches
    if i % 200 == 199:
        torch.cuda.synchronize()
        st = ops.chain_status()
        if st:
            bad += 1
            print(f"call {i}: chain status {st} (batch sizes cycle 1..16)")
            ops.chain_status(reset=True)
torch.cuda.synchronize()
print(f"decode: {n} calls of 1..16 measures in {time.perf_counter() - t0:.1f} s, windows with a non-zero chain status: {bad}")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_arnn import _token_pass_inputs          # noqa: E402
x = _token_pass_inputs()
bad2 = 0
t0 = time.perf_counter()
ref = None
for i in range(max(200, n // 4)):
    tok = ops.arnn_generate(x["emb"], x["oc"], x["W_ih0"], x["b_ih0"], x["W_hh0"], x["b_hh0"], x["W_ih1"], x["b_ih1"], x["W_hh1"],
                            x["b_hh1"], x["W1"], x["b1"], x["W2"], x["b2"])
    if i % 5 == 0:
        filler = filler @ filler * 1e-4
    if ref is None:
        torch.cuda.synchronize(); ref = tok.clone()
    if i % 50 == 49:
        torch.cuda.synchronize()
        st = ops.chain_status()
        if st or not torch.equal(tok, ref):
            bad2 += 1
            print(f"token pass {i}: chain status {st}, tokens equal {torch.equal(tok, ref)}")
            ops.chain_status(reset=True)
torch.cuda.synchronize()
print(f"token pass: {max(200, n // 4)} calls in {time.perf_counter() - t0:.1f} s, bad windows: {bad2}")
print("stress ok" if bad == 0 and bad2 == 0 else "STRESS FAILED")
