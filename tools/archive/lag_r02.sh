#!/bin/bash
# kernel trace + HIP runtime trace of a short bench run -> tools/launch_lag.py.  usage: tools/lag_r02.sh <tag>
TAG=${1:-lag}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/tr -o t -- python3 $ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --no-parity --no-roofline > $OUT/trace.log 2>&1
cd $ROOT
KT=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
AT=$(find $OUT/tr -name "*hip_api_trace.csv" | head -1)
ls $OUT/tr/* | head
for st in 6 7 8; do python3 tools/launch_lag.py $KT $AT $st > $OUT/lag_step$st.txt 2>&1; done
rm -rf $OUT/tr
cat $OUT/lag_step6.txt $OUT/lag_step7.txt | tail -60; tail -2 $OUT/trace.log
