"""Where do the first steps of a fresh process go?  

For the first N steps of the bench's MeasureVAE workload in THIS process: the coin of every step, the host time to queue it, the
GPU time between an event in front of it and one behind it, and the wall time of blocks of steps -- then the same region timed
as bench.py times it (warmup W, steps K) in the same process.

    python3 tools/cold_start.py [--steps 40] [--sync-each] [--prewarm]

--sync-each   synchronise after every step (GPU time of a step alone, no queue)
--prewarm     call Trainer.prewarm() before the first step (what the fix does)
"""
import argparse
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--sync-each", action="store_true")
    ap.add_argument("--prewarm", action="store_true")
    ap.add_argument("--bench-region", type=str, default="5,20")
    ap.add_argument("--gc", choices=("default", "off", "freeze"), default="default",
                    help="Python's cyclic collector: as is / disabled / gc.collect() + gc.freeze() after construction")
    args = ap.parse_args()

    import gc
    gc_log, gc_t0, cur_step = [], [0.0], [-1]

    def on_gc(phase, info):                                  # every collection, with its generation and duration
        if phase == "start":
            gc_t0[0] = time.perf_counter()
        else:
            gc_log.append((cur_step[0], info["generation"], 1e3 * (time.perf_counter() - gc_t0[0]), info["collected"]))
    gc.callbacks.append(on_gc)

    t_proc = time.perf_counter()
    import bench
    from inpaintnet_amd import dp
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dp.seed_rank(1234, 0)
    dp.seed_shared(4321)
    t0 = time.perf_counter()
    wl = bench.VaeWorkload(dev, 0)
    torch.cuda.synchronize()
    print(f"construct workload: {1e3 * (time.perf_counter() - t0):.1f} ms (process so far {time.perf_counter() - t_proc:.1f} s)")
    if args.prewarm and hasattr(wl.trainer, "prewarm"):
        t0 = time.perf_counter()
        wl.trainer.prewarm(wl.tokens)
        torch.cuda.synchronize()
        print(f"prewarm: {1e3 * (time.perf_counter() - t0):.1f} ms")

    if args.gc == "off":
        gc.disable()
    elif args.gc == "freeze":
        gc.collect()
        gc.freeze()
    print(f"gc: {args.gc}; counts {gc.get_count()}, thresholds {gc.get_threshold()}, frozen {gc.get_freeze_count()}, "
          f"tracked objects {len(gc.get_objects())}")
    # the coin sequence the steps will see (random.random() < 0.5 = teacher-forced), without consuming it
    st = random.getstate()
    coins = ["TF" if random.random() < 0.5 else "FR" for _ in range(args.steps)]
    random.setstate(st)

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host = []
    wall = []
    mallocs = []                                     # hipMalloc calls of the caching allocator (segments) and reserved MB behind each step
    torch.cuda.synchronize()
    t_begin = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        t0 = time.perf_counter()
        cur_step[0] = i
        wl.step()
        ev[i + 1].record()
        host.append(time.perf_counter() - t0)
        ms_ = torch.cuda.memory_stats()
        mallocs.append((ms_.get("num_device_alloc", 0), ms_.get("reserved_bytes.all.current", 0) >> 20))
        if args.sync_each:
            torch.cuda.synchronize()
        wall.append(time.perf_counter() - t_begin)
    torch.cuda.synchronize()
    t_end = time.perf_counter() - t_begin
    gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    print(f"{'step':>4} {'coin':>4} {'host_ms':>9} {'gpu_ms':>9} {'wall_at_queue_ms':>17} {'hipMallocs':>10} {'reserved_MB':>11}")
    for i in range(args.steps):
        print(f"{i:>4} {coins[i]:>4} {1e3 * host[i]:>9.3f} {gpu[i]:>9.3f} {1e3 * wall[i]:>17.3f} {mallocs[i][0]:>10} {mallocs[i][1]:>11}")
    cur_step[0] = 10 ** 6
    for st_, gen, ms, n in gc_log:
        if gen >= 1 or ms > 0.5:
            print(f"gc: generation {gen} collection in step {st_}: {ms:.2f} ms, {n} collected")
    print(f"gc: {len(gc_log)} collections so far, {sum(1 for g in gc_log if g[1] == 2)} of generation 2")
    import json
    print("JSON " + json.dumps({"coins": coins, "host_ms": [round(1e3 * h, 3) for h in host], "gpu_ms": [round(g, 3) for g in gpu],
                                "wall_at_queue_ms": [round(1e3 * w, 3) for w in wall], "total_wall_ms": round(1e3 * t_end, 3),
                                "gc": [[a, b, round(c, 3)] for a, b, c, _ in gc_log if a >= 0]}))
    print(f"total wall {1e3 * t_end:.2f} ms for {args.steps} steps; sum host {1e3 * sum(host):.2f}; sum gpu {sum(gpu):.2f}")

    if args.bench_region == "none":
        wl.trainer.check_steps(wait_all=True)
        return
    w, k = (int(v) for v in args.bench_region.split(","))
    for rep in range(3):
        def fence():
            torch.cuda.synchronize()
        dt, _ = bench.timed(wl.step, k, w, fence)
        print(f"bench-style region (warmup {w}, steps {k}) in the warm process, rep {rep}: {1e3 * dt / k:.4f} ms/step")
    dt, _ = bench.timed(wl.step, 400, 20, lambda: torch.cuda.synchronize())
    print(f"bench-style region (warmup 20, steps 400): {1e3 * dt / 400:.4f} ms/step")
    wl.trainer.check_steps(wait_all=True)


if __name__ == "__main__":
    main()
