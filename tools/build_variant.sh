#!/bin/bash
# Build a variant of the GPU library with extra defines, for A/B runs on the GPU box:
#   tools/build_variant.sh w6 -DXE_FAST_WAVES=6     -> xenoeye_amd/libxenoeye_gpu_w6.so
#   tools/build_variant.sh diag -DXE_DIAG           -> the diagnostic build (XE_ABLATE, XE_FAST_R, phase cycles ...)
# then run with XE_GPU_LIB=$PWD/xenoeye_amd/libxenoeye_gpu_w6.so
set -eu
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/xenoeye_amd/csrc/_obj
out=$root/xenoeye_amd/libxenoeye_gpu_$tag.so
for f in xe_kernels.hip xe_walk.hip xe_api.cpp; do
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -g -fPIC -std=c++17 -Wall -DXE_BUILD_ARCH=\"gfx950\" "$@" -c "$root/xenoeye_amd/csrc/$f" -o "$obj/${f}_$tag.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$obj/xe_kernels.hip_$tag.o" "$obj/xe_api.cpp_$tag.o" \
  "$obj/xe_walk.hip_$tag.o" "$obj"/xe_fields.c.o "$obj"/xe_iplist.c.o "$obj"/xe_filter.c.o "$obj"/xe_prepass.c.o "$obj"/xe_conf.c.o "$obj"/xe_pcap.c.o \
  "$obj"/xe_geodb.c.o -lpthread -lm
echo "$out"
