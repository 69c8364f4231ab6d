#!/bin/bash
# A/B the headline bench under environment switches: tools/ab_bench.sh "NAME=VAL ..." "NAME=VAL ..." ...
for cfg in "$@"; do
  out=$(env $cfg timeout 300 python bench.py --steps 150 --warmup 15 --no-cpu-baseline --no-extras --no-parity --no-roofline 2>/dev/null | tail -1)
  python3 - "$cfg" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); print(f"{sys.argv[1]:<60} {d['value']:>10.1f} measures/s  {d['ms_per_step']:.4f} ms/step")
PY
done
