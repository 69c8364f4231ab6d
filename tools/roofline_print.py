#!/usr/bin/env python3
"""Print the per-kernel table of a bench.py JSON line read from stdin: `python bench.py ... | python tools/roofline_print.py [n]`."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
print(d["value"], d["ms_per_step"], d["roofline"]["step_kernel_ms"])
for k in d["roofline"]["kernels"][:n]:
    print(f"{k['kernel']:<46} n={k['launches_per_step']:<5} {k['avg_us']:>8.1f} us {k['ms_per_step']:.3f} ms  {k['tflops']:>6.1f} TF")
