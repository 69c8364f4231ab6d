#!/usr/bin/env python3
"""Headline step with torch's stream pool created BEFORE the library's streams (what a data-parallel run does: the process group's
first collective takes a stream from the pool) or after: python tools/stream_order.py [before|after|hp-before]"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
what = sys.argv[1] if len(sys.argv) > 1 else "after"
sys.stdout = sys.stderr
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
keep = []
if what == "before":
    keep.append(torch.cuda.Stream(device=dev))
elif what == "hp-before":
    keep.append(torch.cuda.Stream(device=dev, priority=-1))
elif what.startswith("used"):                      # used2: two pool streams that have run a kernel before the library starts
    for i in range(int(what[4:])):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            torch.zeros(16, device=dev).add_(1)
        keep.append(st)
    torch.cuda.synchronize()
wl = bench.VaeWorkload(dev, 0)
if what == "after":
    for _ in range(3): wl.step()
    keep.append(torch.cuda.Stream(device=dev))
dt, _ = bench.timed(wl.step, 200, 20, torch.cuda.synchronize)
print(f"torch stream pool created {what:<10}: {256 * 200 / dt:9.1f} measures/s  {1e3 * dt / 200:.3f} ms/step")
