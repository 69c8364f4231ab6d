#!/bin/bash
mkdir -p gpurun_out/r03c build
OUT=gpurun_out/r03c/exp4.txt; : > $OUT
for v in "-DRING_=8" "-DRING_=4" "-DRING_=12"; do
  echo "== exp_chain3 $v" >> $OUT
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I inpaintnet_amd/csrc tools/exp_chain3.hip $v -o build/exp_c3 2>> $OUT && timeout 120 build/exp_c3 >> $OUT 2>&1
done
cat $OUT
