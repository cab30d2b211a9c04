#!/bin/bash
# fixed-window kernel: parity subset, per-table times, the bench pair, and the pair for other chunk counts (diagnostic build)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fwc; timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "fwm or fuzz" > gpurun_out/fwc/pytest.log 2>&1 || { tail -40 gpurun_out/fwc/pytest.log; exit 1; }; tail -1 gpurun_out/fwc/pytest.log
timeout -k 10 300 python -u tools/fwm_time.py 4194304 2>&1 | grep rows || exit 1
bash tools/ab_fwm.sh fwc default || exit 1
for c in "$@"; do
  echo "== chunks $c"
  XE_FWM_CHUNKS_RT=$c bash tools/ab_fwm.sh fwc diag || exit 1
done
