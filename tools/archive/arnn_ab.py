"""One-box A/B for the AnticipationRNN step (8.05 -> 9.01 ms between BENCH_r02 and BENCH_r03 with no
change to lstm.hip / arnn.py): time bench.arnn_extra() alone, with more steps, and again after each of the extras that
bench.py now runs in front of it (LatentRNN workload, chain_generations, vae_train_4096, vocab), then after
torch.cuda.empty_cache().  Prints one line per arm."""
import os
import sys

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

sys.stdout = sys.stderr


def arm(name, **kw):
    r = bench.arnn_extra(**kw)["anticipation_rnn_train"]
    free, total = torch.cuda.mem_get_info()
    print(f"{name:<44} {r['ms_per_step']:8.3f} ms/step   reserved {torch.cuda.memory_reserved() / 2**30:6.2f} GiB  "
          f"device free {free / 2**30:6.1f} GiB", flush=True)


torch.cuda.set_device(0)
arm("alone, 8 steps (bench default)")
arm("alone, 8 steps again")
arm("alone, 40 steps / 5 warmup", steps=40, warmup=5)
wl = bench.VaeWorkload(torch.device("cuda", 0), 0)
for _ in range(30):
    wl.step()
torch.cuda.synchronize()
arm("after 30 VAE steps")
lw = bench.LatentWorkload(torch.device("cuda", 0), 0, vae=wl.model, ds=wl.ds)
for _ in range(13):
    lw.step()
torch.cuda.synchronize()
wl.model.trainable = True
wl.model.train()
arm("after LatentRNN workload")
bench.chain_generations_extra(wl, steps=20, warmup=5)
arm("after chain_generations_extra")
bench.vae4096_extra(wl)
arm("after vae4096_extra")
arm("after vae4096_extra, 40 steps", steps=40, warmup=5)
bench.vocab_extra()
arm("after vocab_extra")
torch.cuda.empty_cache()
arm("after empty_cache()")
arm("after empty_cache(), 40 steps", steps=40, warmup=5)
