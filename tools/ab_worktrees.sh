#!/bin/bash
# Headline step in several checkouts on ONE box, alternating (box-to-box variance is +-5 %): tools/ab_worktrees.sh <dir> <dir> ...
# each <dir> holds a built copy of the repo (git worktree add build/wt_x <sha>; python -c "from inpaintnet_amd import _lib; _lib.build()")
# Every checkout is measured with the driver's command (20 / 5, fresh process) and with a long region (200 / 20).
for rep in 1 2 3; do
  for wt in "$@"; do
    for sw in "20 5" "200 20"; do
      read -r K W <<< "$sw"
      out=$(cd "$wt" && timeout 300 python3 bench.py --gpus 1 --steps $K --warmup $W --no-cpu-baseline --no-extras --no-parity --no-roofline 2>/dev/null | tail -1)
      python3 - "$wt" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); print(f"{sys.argv[1]:<24} steps {d['steps']:>3} warmup {d['warmup']:>2} {d['value']:>10.1f} measures/s  {d['ms_per_step']:.4f} ms/step")
PY
    done
  done
done
