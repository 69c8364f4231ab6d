#!/bin/bash
mkdir -p gpurun_out/r03c build
OUT=gpurun_out/r03c/exp3.txt; : > $OUT
B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -I inpaintnet_amd/csrc tools/exp_chain2.hip -DCSTRIDE_=64 -DONLY_FULL_ -DALSO_NOSYNC_"
for v in "-DMS_=4" "-DMS_=2" "-DMS_=1"; do
  echo "== $v" >> $OUT
  $B $v -o build/exp_c 2>> $OUT && timeout 120 build/exp_c >> $OUT 2>&1
done
cat $OUT
