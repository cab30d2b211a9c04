#!/bin/bash
# flow cursor and shim: parity tests, then the shim's throughput (tools/shimbench)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/shim
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "cursor or shim or golden or parity" > gpurun_out/shim/pytest.log 2>&1 || { tail -40 gpurun_out/shim/pytest.log; exit 1; }
tail -1 gpurun_out/shim/pytest.log
timeout -k 10 300 python -u tools/shimbench/run.py c2 1048576 3 || exit 1
timeout -k 10 300 python -u tools/shimbench/run.py c1 1048576 3 || exit 1
timeout -k 10 300 python -u tools/shimbench/run.py c3 200000 3 || exit 1
