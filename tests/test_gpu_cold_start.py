"""The first steps of a FRESH process cost what later steps cost (profiles/r05_cold_start.txt).

Round 4's driver line (bench.py --steps 20 --warmup 5 in a fresh process) read 6.31 ms per step against 3.65 in steady state:
CPython's full garbage collection over the import-time heap (~170 k tracked objects, 36-40 ms) fell into the first dozen steps,
the launch queue drained and the GPU idled.  Trainer.__init__ freezes that heap now (trainer.settle_python_heap).  Here: a
fresh interpreter (no bytecode caches written or needed: -B) runs 125 steps of the bench's workload; the wall time of steps
5..24 -- the driver's timed region -- must be within 15 % of steps 100..119, and no collection inside the steps may take > 5 ms."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cold(extra_env=None):
    env = dict(os.environ, **(extra_env or {}))
    out = subprocess.run([sys.executable, "-B", os.path.join(REPO, "tools", "cold_start.py"), "--steps", "125",
                          "--bench-region", "none"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("JSON ")][-1][5:])


@pytest.mark.gpu
def test_fresh_process_first_steps_cost_what_later_steps_cost():
    d = run_cold()
    wall = d["wall_at_queue_ms"]
    gpu = d["gpu_ms"]
    early = sum(gpu[5:25])                      # GPU timeline of the driver's timed region (idle gaps included)
    late = sum(gpu[100:120])
    print(f"steps 5..24: {early / 20:.3f} ms per step (GPU timeline), steps 100..119: {late / 20:.3f}; "
          f"host: {(wall[24] - wall[4]) / 20:.3f} / {(wall[119] - wall[99]) / 20:.3f} ms per step; gc events {d['gc'][:8]}")
    assert early <= 1.15 * late, (early, late)
    # no collection-sized host stall in the region either (a full collection is 36-40 ms; the host runs ~2.5 ms per step ahead of
    # the GPU, so a stall below ~20 ms is absorbed by the queue -- one 11 ms hiccup at step 5 was seen once in a dozen runs, with the
    # GPU timeline unaffected: 3.634 vs 3.622 ms per step)
    assert max(d["host_ms"][5:25]) < 20.0, d["host_ms"][5:25]
    assert all(ms < 5.0 for _, _, ms in d["gc"]), d["gc"]
