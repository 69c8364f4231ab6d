#!/bin/bash
# kernel trace of a short bench run + per-step timeline (tools/timeline.py).  usage: tools/trace_r02.sh <tag> [env assignments...]
TAG=${1:-trace}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline > $OUT/trace.log 2>&1
cd $ROOT
TR=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $TR 5 > $OUT/timeline_step5.txt 2>&1
python3 tools/timeline.py $TR 6 full > $OUT/timeline_step6.txt 2>&1
python3 tools/pmc_summary.py stats $TR > $OUT/kernel_stats.txt 2>&1
rm -rf $OUT/tr
cat $OUT/timeline_step5.txt; tail -2 $OUT/trace.log
