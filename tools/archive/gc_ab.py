import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
sys.stdout = sys.stderr
dev = torch.device("cuda", 0)
wl = bench.VaeWorkload(dev, 0)
for gc_on in ("1", "0", "1", "0"):
    os.environ["INET_BENCH_GC"] = gc_on
    dt, _ = bench.timed(wl.step, 150, 15, torch.cuda.synchronize)
    print(f"VAE step, gc {'on ' if gc_on == '1' else 'off'}: {1e3 * dt / 150:.4f} ms", flush=True)
lw = bench.LatentWorkload(dev, 0, vae=wl.model, ds=wl.ds)
for gc_on in ("1", "0", "1", "0"):
    os.environ["INET_BENCH_GC"] = gc_on
    dt, _ = bench.timed(lw.step, 20, 4, torch.cuda.synchronize)
    print(f"latent step, gc {'on ' if gc_on == '1' else 'off'}: {1e3 * dt / 20:.4f} ms", flush=True)
for gc_on in ("1", "0", "1", "0"):
    os.environ["INET_BENCH_GC"] = gc_on
    r = bench.arnn_extra(tables=False)["anticipation_rnn_train"]
    print(f"arnn step (after the others), gc {'on ' if gc_on == '1' else 'off'}: {r['ms_per_step']:.4f} ms", flush=True)
