#!/usr/bin/env python3
"""The stream structure of a data-parallel run on ONE GPU: an NCCL (= RCCL) process group of one rank, dp.world_size() reporting 2 so
that the bucketed gradient exchange runs on the streams a multi-GPU run uses (prep stream + the process group's own stream); the
all-reduce over one rank moves nothing.  Not a scaling number: it shows what the extra busy streams cost the step.
python tools/dp_streams_1gpu.py [vae|latent] [dp|plain]"""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
which = sys.argv[1] if len(sys.argv) > 1 else "vae"
mode = sys.argv[2] if len(sys.argv) > 2 else "dp"
sys.stdout = sys.stderr
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if mode == "dp":
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from inpaintnet_amd import dp
    dp.world_size = lambda: 2
    if os.environ.get("X_ONE_SIDE", "1") == "1":
        dp.one_side_stream()                                   # what dp.init_from_env / dp.broadcast_params do in a real run
    if (len(sys.argv) > 3 and sys.argv[3] == "twin") or os.environ.get("X_PREP_TWIN") == "1":            # the bucket all-reduces issued from the library's twin stream
        from inpaintnet_amd import ops
        dp._prep_stream = lambda device: ops.twin_stream(device)
    if len(sys.argv) > 3 and sys.argv[3].startswith("busy"):   # a stand-in all-reduce that really runs: a kernel over the bucket
        nccl_like = torch.cuda.Stream(device=dev, priority=-1)    # on a fifth stream (where the process group's own kernels run)
        reps = int(sys.argv[3][4:] or 8)

        class _Work:
            def wait(self):
                torch.cuda.current_stream().wait_stream(nccl_like)

        def fake_all_reduce(t, op=None, async_op=False):
            nccl_like.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(nccl_like):
                for _ in range(reps):
                    t.mul_(1.0)
            w = _Work()
            if not async_op:
                w.wait()
            return w
        torch.distributed.all_reduce = fake_all_reduce
    if len(sys.argv) > 3 and sys.argv[3] == "noex":            # the same code path without the collectives
        dp.set_exchange(False)
import bench
wl = (bench.VaeWorkload if which == "vae" else bench.LatentWorkload)(dev, 0)
n = 200 if which == "vae" else 40
for _ in range(10): wl.step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(n): wl.step()
    torch.cuda.synchronize()
    print(f"{which} {mode} {sys.argv[3] if len(sys.argv) > 3 else str()}: {1e3 * (time.perf_counter() - t0) / n:.3f} ms per step")
if mode == "dp":
    torch.distributed.destroy_process_group()
