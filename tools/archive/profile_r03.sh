#!/bin/bash
# rocprofv3 passes of the headline bench on the GPU box (run through gpurun from the repo root):
#   kernel-trace + stats (as benched, and with the side streams off), then FETCH_SIZE and WRITE_SIZE PMC passes (separate: TCC
#   slots) once with every step teacher-forced and once free-running (INET_BENCH_COIN), so that every step of a pass launches
#   the same kernel sequence and kernels that share an instantiation + grid can be told apart by launch order.
# usage: tools/profile_r03.sh <tag>
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-parity --no-roofline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
INET_SIDE_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o s -- python3 $ROOT/bench.py $ARGS > $OUT/serial.log 2>&1
for coin in tf fr; do
  export INET_BENCH_COIN=$coin
  INET_BENCH_SEQ=$OUT/kernel_sequences_$coin.json timeout 600 python3 $ROOT/bench.py $ARGS > $OUT/seq_$coin.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$coin -o f -- python3 $ROOT/bench.py $ARGS > $OUT/fetch_$coin.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$coin -o w -- python3 $ROOT/bench.py $ARGS > $OUT/write_$coin.log 2>&1
  unset INET_BENCH_COIN
done
cd $ROOT
ST=$(find $OUT/stats -name "*kernel_trace.csv" | head -1)
SE=$(find $OUT/serial -name "*kernel_trace.csv" | head -1)
python3 tools/pmc_summary.py stats $ST > $OUT/kernel_stats.txt 2>&1
python3 tools/pmc_summary.py stats $SE > $OUT/kernel_stats_side_streams_off.txt 2>&1
python3 tools/timeline.py $ST 6 full > $OUT/timeline_full_step.txt 2>&1
for coin in tf fr; do
  FE=$(find $OUT/fetch_$coin -name "*counter_collection.csv" | head -1)
  WR=$(find $OUT/write_$coin -name "*counter_collection.csv" | head -1)
  python3 tools/pmc_summary.py pmc $FE $WR $OUT/pmc_traffic_$coin.json $OUT/kernel_sequences_$coin.json > $OUT/pmc_summary_$coin.txt 2>&1
done
python3 tools/pmc_summary.py merge $OUT/pmc_traffic.json $OUT/pmc_traffic_tf.json $OUT/pmc_traffic_fr.json >> $OUT/pmc_summary_tf.txt 2>&1
rm -rf $OUT/stats $OUT/fetch_tf $OUT/write_tf $OUT/fetch_fr $OUT/write_fr $OUT/serial
head -20 $OUT/kernel_stats.txt; cat $OUT/pmc_summary_tf.txt | head -30; tail -3 $OUT/stats.log $OUT/fetch_tf.log
