#!/bin/bash
# Headline step in several checkouts on ONE box, alternating (box-to-box variance is +-5 %): tools/ab_worktrees.sh <dir> <dir> ...
# each <dir> holds a built copy of the repo (git worktree add build/wt_x <sha>; python -c "from inpaintnet_amd import _lib; _lib.build()")
for rep in 1 2 3; do
  for wt in "$@"; do
    out=$(cd "$wt" && timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --no-parity --no-roofline 2>/dev/null | tail -1)
    python3 - "$wt" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); print(f"{sys.argv[1]:<24} {d['value']:>10.1f} measures/s  {d['ms_per_step']:.4f} ms/step")
PY
  done
done
