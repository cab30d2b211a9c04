#!/bin/bash
# Build a variant of the GPU library with extra defines, for A/B runs on the GPU box:
#   tools/build_variant.sh w6 -DXE_FAST_WAVES=6     -> xenoeye_amd/libxenoeye_gpu_w6.so
# then run with XE_GPU_LIB=$PWD/xenoeye_amd/libxenoeye_gpu_w6.so
set -eu
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/xenoeye_amd/csrc/_obj
out=$root/xenoeye_amd/libxenoeye_gpu_$tag.so
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -g -fPIC -std=c++17 -Wall -DXE_BUILD_ARCH=\"gfx950\" "$@" -c "$root/xenoeye_amd/csrc/xe_kernels.hip" -o "$obj/xe_kernels_$tag.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$obj/xe_kernels_$tag.o" "$obj/xe_api.cpp.o" \
  "$obj/xe_walk.hip.o" "$obj"/xe_fields.c.o "$obj"/xe_iplist.c.o "$obj"/xe_filter.c.o "$obj"/xe_prepass.c.o "$obj"/xe_conf.c.o "$obj"/xe_pcap.c.o \
  "$obj"/xe_geodb.c.o -lpthread -lm
echo "$out"
