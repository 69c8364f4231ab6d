#!/bin/bash
# A second build of the library with extra compiler flags, for A/B runs on one box: tools/build_variant.sh <name> <flags...>
# -> build/lib_<name>.so (select it with INET_LIB_PATH).  Only the sources named in VARIANT_SRCS (default: the granule kernels) are
# recompiled, the other objects are taken from build/obj.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
srcs=${VARIANT_SRCS:-"arnn_gen decode_b1"}
mkdir -p build/obj_$name
cp build/obj/*.o build/obj_$name/
for f in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment "$@" -c inpaintnet_amd/csrc/$f.hip -o build/obj_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic-functions -o build/lib_$name.so build/obj_$name/*.o -ldl
echo build/lib_$name.so
