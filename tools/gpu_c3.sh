#!/bin/bash
# sFlow-related parity tests, then the C3 line.  Usage: tools/gpu_c3.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "sflow or walk or fuzz or golden or shim or fwm" > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
tail -2 "$out/pytest.log"
bash tools/prof_one.sh $tag c3 --walk-reps 4 | awk -F, '{print $1, $2, $4}' | cut -c1-120 | head -8
python - "$out/bench_c3.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; p=d["device_path"]
print("c3 kernel %.4f walk %.4f path %.4f frac %.4f" % (r["kernel_ms"], p["walk_ms"], p["device_path_ms"], p["frac"]))
PY
