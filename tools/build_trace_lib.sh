#!/bin/bash
# Trace build of the library (phase stamps in the step kernels) for tools/trace_steps.py -- not the product build.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$ROOT/build"
cd "$ROOT/inpaintnet_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-comment -DINET_STEP_TRACE "$@" \
    -o "$ROOT/build/libinet_trace.so" gemm.hip gru.hip pointwise.hip seq.hip vae.hip api.hip prof.hip side.hip lstm.hip
echo "built $ROOT/build/libinet_trace.so"
