#!/bin/bash
# tools/fwm_time.py on the default build and on library variants.  Usage: tools/ab_fwm_time.sh variant...
for v in default "$@"; do
  if [ "$v" = default ]; then unset XE_GPU_LIB; else export XE_GPU_LIB=$PWD/xenoeye_amd/libxenoeye_gpu_$v.so; fi
  echo "== $v"
  timeout -k 10 200 python -u tools/fwm_time.py 2>&1 | grep rows || exit 1
done
