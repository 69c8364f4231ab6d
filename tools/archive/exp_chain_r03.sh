#!/bin/bash
# Chain-step experiments of round 3 (GPU box, from the repo root; results: profiles/r03_c_chain_two_per_cu.txt, r03_j_*):
#   tools/exp_chain2.hip  the library kernel's step in isolation: one 64-row workgroup per CU (MS_=4) vs two 32-row workgroups
#                         per CU (MS_=2), in phase, de-phased at start (SKEW_), at different wave priorities (PRIO_), free-running
#   tools/exp_chain3.hip  second design: W_hh slice in LDS, one row block per wave, no barrier in the step
mkdir -p gpurun_out/r03c build
OUT=gpurun_out/r03c/exp_all.txt; : > $OUT
B="hipcc --offload-arch=gfx950 -O3 -std=c++17 -I inpaintnet_amd/csrc tools/exp_chain2.hip -DCSTRIDE_=64 -DONLY_FULL_"
for v in "-DMS_=4 -DALSO_NOSYNC_" "-DMS_=2 -DALSO_NOSYNC_" "-DMS_=2 -DSKEW_=350" "-DMS_=2 -DPRIO_=3" "-DMS_=1 -DALSO_NOSYNC_"; do
  echo "== exp_chain2 $v" >> $OUT
  $B $v -o build/exp_c 2>> $OUT && timeout 120 build/exp_c >> $OUT 2>&1
done
for v in "-DRING_=4" "-DRING_=8"; do
  echo "== exp_chain3 $v" >> $OUT
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I inpaintnet_amd/csrc tools/exp_chain3.hip $v -o build/exp_c3 2>> $OUT && timeout 120 build/exp_c3 >> $OUT 2>&1
done
cat $OUT
