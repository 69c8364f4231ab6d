#!/bin/bash
# Memory-side traffic of the gemm_bf3 shapes under both tile orders (tools/bf3_map_ab.py child = the product loop):
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (TCC slots), summarised per kernel|grid by tools/pmc_summary.py.
# usage (from the repo root, through gpurun): tools/bf3_map_pmc.sh <outdir>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/${1:-gpurun_out/bf3map}
mkdir -p $OUT
python3 $ROOT/tools/bf3_map_ab.py > $OUT/timing.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  export INET_BF3_MAP=$m
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch$m -o f -- python3 $ROOT/tools/bf3_map_ab.py child > $OUT/fetch$m.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write$m -o w -- python3 $ROOT/tools/bf3_map_ab.py child > $OUT/write$m.log 2>&1
  FE=$(find $OUT/fetch$m -name "*counter_collection.csv" | head -1)
  WR=$(find $OUT/write$m -name "*counter_collection.csv" | head -1)
  (cd $ROOT && python3 tools/pmc_summary.py pmc $FE $WR $OUT/traffic_map$m.json) > $OUT/traffic_map$m.txt 2>&1
  rm -rf $OUT/fetch$m $OUT/write$m
done
unset INET_BF3_MAP
cat $OUT/timing.txt; grep gemm_bf3 $OUT/traffic_map0.txt $OUT/traffic_map1.txt
