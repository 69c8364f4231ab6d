#!/bin/bash
# Build-time A/B of the second-generation forward chain's A-operand schedule (csrc/gru_chain2.hip contract2): ring depth R (k-steps of
# h pieces requested ahead of the MFMAs) and DEAL (0: the three loads of a k-step as one burst in front of its 27 MFMAs; 1: one load
# in front of every nine).  Run in the build container:  tools/ab_chain2_variants.sh build "3 0" "3 1" "4 1"   -> build/lib_r<R>d<D>.so
# then on the GPU box:                                     tools/ab_chain2_variants.sh run   "3 0" "3 1" "4 1"
mode=$1; shift
if [ "$mode" = build ]; then
  objs=$(ls build/obj/*.o | grep -v gru_chain2.o)
  for v in "$@"; do
    read -r R D <<< "$v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -DINET_CHAIN2_RING=$R -DINET_CHAIN2_DEAL=$D -c inpaintnet_amd/csrc/gru_chain2.hip -o /tmp/gc2_v.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/lib_r${R}d${D}.so $objs /tmp/gc2_v.o || exit 1
  done
  ls -la build/lib_r*.so
else
  for rep in 1 2; do
    for v in default "$@"; do
      if [ "$v" = default ]; then unset INET_LIB_PATH; else read -r R D <<< "$v"; export INET_LIB_PATH=$PWD/build/lib_r${R}d${D}.so; fi
      out=$(timeout 300 python3 bench.py --gpus 1 --steps 300 --warmup 20 --no-cpu-baseline --no-extras --no-parity 2>/dev/null | tail -1)
      python3 -c "
import json,sys
d=json.loads(sys.argv[1]); r=d['roofline']
print(f'{sys.argv[2]:<10} {d[\"ms_per_step\"]:.4f} ms/step   dominant kernel {r[\"avg_launch_us\"]:.1f} us  frac {r[\"frac\"]:.4f}')" "$out" "$v"
    done
  done
fi
