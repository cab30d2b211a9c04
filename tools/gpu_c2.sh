#!/bin/bash
# Netflow-walk parity tests, then the C2 and C1 quick lines with walk time.  Usage: tools/gpu_c2.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "walk or fuzz or golden or shim or netflow or scale or fwm" > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
tail -2 "$out/pytest.log"
for w in c2 c1 c5; do
timeout -k 10 300 python bench.py --workload $w --quick > "$out/bench_$w.json" 2> "$out/bench_$w.err" || { tail -20 "$out/bench_$w.err"; exit 1; }
python - "$out/bench_$w.json" $w <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; p=d["device_path"]
print("%s kernel %.4f walk %.4f path %.4f frac %.4f walked %s" % (sys.argv[2], r["kernel_ms"], p["walk_ms"], p["device_path_ms"], p["frac"], d["host_inclusive"]["headers_walked_on_device"]))
PY
done
