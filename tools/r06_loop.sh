#!/bin/bash
# Robustness hunt with the recorder armed: the GPU suite N times back to back on one lease, then one soak; logs in
# FULL (not tails).  usage: tools/r06_loop.sh <suite repetitions> [soak: 0/1]
mkdir -p gpurun_out
N=${1:-5}; SOAK=${2:-1}
tag=$(date +%s)
log=gpurun_out/r06_loop_$tag.log
echo "lease $tag: $N suite repetitions, soak $SOAK" > $log
for i in $(seq 1 $N); do
  echo "=== suite repetition $i" >> $log
  python -m pytest tests -m gpu -q -p no:cacheprovider >> $log 2>&1
  echo "=== repetition $i rc $?" >> $log
done
if [ "$SOAK" = "1" ]; then
  echo "=== soak (tools/soak.py 20000 3000 3000) + recorder" >> $log
  python - >> $log 2>&1 <<'PY'
import runpy, sys
sys.argv = ["tools/soak.py", "20000", "3000", "3000"]
try:
    runpy.run_path("tools/soak.py", run_name="__main__")
finally:
    from inpaintnet_amd import ops
    r = ops.slow_waits()
    print("recorder after the soak:", {"count": r["count"], "noted": r["noted"], "entries": r["entries"][:8]}, file=sys.stderr)
PY
  echo "=== soak rc $?" >> $log
fi
grep -E "=== |passed|failed|soak ok|recorder after" $log
