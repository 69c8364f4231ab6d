"""eval-mode decode latency at small batch (bench.decode_latency_extra): python tools/decode_latency.py"""
import os, sys
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
sys.stdout = sys.stderr
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
r = bench.decode_latency_extra(wl.model, iters=200)["decoder_eval"]
print({k: v["ms_per_call"] for k, v in r.items() if isinstance(v, dict)})
