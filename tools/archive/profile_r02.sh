#!/bin/bash
# rocprofv3 passes of the headline bench on the GPU box (run through gpurun from the repo root):
#   kernel-trace + stats, then FETCH_SIZE and WRITE_SIZE PMC passes (separate: TCC slots), summaries into gpurun_out/.
# usage: tools/profile_r02.sh <tag>
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline"
export INET_BENCH_SEQ=$OUT/kernel_sequences.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
unset INET_BENCH_SEQ
# the same command with the side streams off: every kernel alone on the chip, the durations bench.py's roofline table quotes
INET_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o s -- python3 $ROOT/bench.py $ARGS > $OUT/serial.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
cd $ROOT
ST=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
FE=$(find $OUT/fetch -name "*counter_collection.csv" | head -1)
WR=$(find $OUT/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_summary.py stats $ST $OUT/kernel_sequences.json > $OUT/kernel_stats.txt 2>&1
SE=$(find $OUT/serial -name "*kernel_trace.csv" | head -1)
python3 tools/pmc_summary.py stats $SE $OUT/kernel_sequences.json > $OUT/kernel_stats_side_streams_off.txt 2>&1
python3 tools/pmc_summary.py pmc $FE $WR $OUT/pmc_traffic.json $OUT/kernel_sequences.json > $OUT/pmc_summary.txt 2>&1
# the raw traces are large: keep only the summaries for the merge back
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/serial
head -25 $OUT/kernel_stats.txt; cat $OUT/pmc_summary.txt; tail -3 $OUT/stats.log $OUT/fetch.log $OUT/write.log
