#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export XE_GPU_LIB=$GRAFT_REPO_ROOT/xenoeye_amd/libxenoeye_gpu_diag.so
for a in 64; do
  XE_ABLATE=$a timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --quick --workload c3 --records 200000 --replicate 16 2>&1 >/dev/null | grep "phase cycles" | tail -2 || exit 1
done
