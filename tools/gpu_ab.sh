#!/bin/bash
# parity subset, then kernel times of the workloads named.  Usage: tools/gpu_ab.sh <tag> <pytest -k expr> <workloads...>
set -u
tag=$1; kexpr=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$tag
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "$kexpr" > gpurun_out/$tag/pytest.log 2>&1 || { tail -30 gpurun_out/$tag/pytest.log; exit 1; }
tail -1 gpurun_out/$tag/pytest.log
for w in "$@"; do
  args="--workload $w --quick"
  [ "$w" = c3 ] && args="--workload c3 --records 200000 --replicate 16 --quick"
  [ "$w" = c4 ] && args="--workload c4 --records 2097152 --quick"
  timeout -k 10 300 python bench.py $args > gpurun_out/$tag/$w.json 2> gpurun_out/$tag/$w.err || { tail -20 gpurun_out/$tag/$w.err; exit 1; }
  python - gpurun_out/$tag/$w.json $w <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]
print(sys.argv[2], r["kernel"], "%.4f ms frac %.3f" % (r["kernel_ms"], r["frac"]), "walk %.4f path %.4f" % (d["device_path"]["walk_ms"], d["device_path"]["device_path_ms"]))
PY
done
