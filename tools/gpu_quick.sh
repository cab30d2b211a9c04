#!/bin/bash
# Quick GPU check: parity tests, then the two benchmark lines.  Usage: tools/gpu_quick.sh <tag>
set -u
tag=${1:-quick}
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
timeout -k 10 300 python bench.py --quick > "$out/bench_c2.json" 2> "$out/bench_c2.err" || { tail -20 "$out/bench_c2.err"; exit 1; }
python - "$out/bench_c2.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("C2 %.3f Grec/s kernel %.4f ms frac %.4f" % (d["value"]/1e9, r["kernel_ms"], r["frac"]))
PY
timeout -k 10 300 python bench.py --workload c1 --quick > "$out/bench_c1.json" 2> "$out/bench_c1.err" || { tail -20 "$out/bench_c1.err"; exit 1; }
python - "$out/bench_c1.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("C1 %.3f Grec/s kernel %.4f ms frac %.4f" % (d["value"]/1e9, r["kernel_ms"], r["frac"]))
PY
