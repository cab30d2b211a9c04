#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export XE_GPU_LIB=$GRAFT_REPO_ROOT/xenoeye_amd/libxenoeye_gpu_diag.so
for r in 1 2 4; do
  XE_FAST_R=$r timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --quick --workload c4 --records 2097152 2>/dev/null | python -c "
import json,sys;d=json.load(sys.stdin);print('R=$r kernel_ms %.4f' % d['roofline']['kernel_ms'], d['roofline']['kernel'])" || exit 1
done
