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
