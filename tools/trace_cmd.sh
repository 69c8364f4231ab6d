#!/bin/bash
# kernel trace of an arbitrary python script + per-step timeline: tools/trace_cmd.sh <tag> <step> <script> [args...]
TAG=$1; STEP=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -o t -- python3 $ROOT/$@ > $OUT/trace.log 2>&1
cd $ROOT
TR=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $TR $STEP full > $OUT/timeline.txt 2>&1
python3 tools/pmc_summary.py stats $TR > $OUT/kernel_stats.txt 2>&1
rm -rf $OUT/tr
