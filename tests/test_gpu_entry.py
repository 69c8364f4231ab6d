"""The driver's entry points in ONE fresh process, in the order that used to fail: build() loads libinpaintnet_hip.so before
anything has touched the GPU; smoke() then runs a training step.  (PyTorch-ROCm ships its own HIP runtime: the library must bind
to that one -- inpaintnet_amd/_lib.py imports torch before it loads the library; loaded first it pulled in /opt/rocm's copy as a
second runtime and the first launch failed with hipErrorNoDevice.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loaded_before_torch_then_smoke():
    code = ("from inpaintnet_amd import _lib\n"
            "L = _lib.lib()\n"                       # (what build() does after compiling; without the rebuild: seconds, not minutes)
            "assert L.inet_abi_version() == 1\n"
            "import __graft_entry__ as g\n"
            "g.smoke()\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "[smoke] ok" in r.stdout


def test_bench_prints_exactly_one_small_line():
    """The driver's command (short form: no extras, no CPU baseline) in a subprocess: stdout is ONE JSON line under bench.py's
    4 KB limit with the contract's keys; the detail went to bench_detail.json (round 5's 20 KB line was dropped by the driver)."""
    import json
    detail = os.path.join(ROOT, "gpurun_out", "bench_detail_test.json")
    os.makedirs(os.path.dirname(detail), exist_ok=True)
    env = dict(os.environ, INET_BENCH_DETAIL=detail)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--no-extras", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[:2000]
    assert len(lines[0]) < 4096
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1 and line["chain_timeouts"] == 0
    assert line["parity_checked"] is True
    assert "priming_steps" not in line["config"]               # --warmup 5 means five
    assert line["roofline"]["bound"] in ("mfma", "hbm") and 0.0 < line["roofline"]["frac"] < 1.0
    full = json.load(open(detail))
    assert full["value"] == line["value"] and len(full["roofline"]["kernels"]) >= 4
