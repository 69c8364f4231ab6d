#!/bin/bash
# A/B the headline bench under environment switches, on ONE box:  tools/ab_bench.sh "NAME=VAL ..." "NAME=VAL ..." ...
# Each configuration is measured twice: with the DRIVER's command (--steps 20 --warmup 5 in a fresh process: what BENCH_rNN.json
# records -- round 4 never ran it and missed a cold-start transient, profiles/r05_cold_start.txt) and with a long region (400 / 20).
for cfg in "$@"; do
  for sw in "20 5" "400 20"; do
    read -r K W <<< "$sw"
    out=$(env $cfg timeout 300 python3 bench.py --gpus 1 --steps $K --warmup $W --no-cpu-baseline --no-extras --no-parity --no-roofline 2>/dev/null | tail -1)
    python3 - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
gc = d.get("gc_in_timed_region", 0)
print(f"{sys.argv[1]:<48} steps {d['steps']:>3} warmup {d['warmup']:>2}  {d['value']:>10.1f} measures/s  {d['ms_per_step']:.4f} ms/step"
      f"  slowest of the first steps {max(d.get('first_steps_ms', [0])):.2f} ms, collections {gc}, slow waits {d.get('slow_waits')}")
PY
  done
done
