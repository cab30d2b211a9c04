#!/bin/bash
# the drop-in entry points on 1, 2, 4, 8 capture threads (a handle each) of one GPU
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=${1:-c2}; n=${2:-1048576}
for t in 1 2 4 8; do
  timeout -k 10 300 python -u tools/shimbench/run.py $w $n 3 1 $t || exit 1
done
