#!/bin/bash
# One GPU-box session of round 3: parity tests, then the default benchmark line.
# Usage: tools/gpu_r03.sh <tag> [pytest -k expression]
set -u
tag=${1:-r03}
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -n "${2:-}" ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "$2" > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
else
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
fi
tail -3 "$out/pytest.log"
timeout -k 10 500 python bench.py > "$out/bench_full.json" 2> "$out/bench_full.err" || { tail -20 "$out/bench_full.err"; exit 1; }
python - "$out/bench_full.json" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
def line(name, x):
    r=x["roofline"]; p=x.get("device_path") or {}
    print("%s kernel %.4f ms frac %.4f | walk %s device path %s frac %s" % (name, r["kernel_ms"], r["frac"],
        p.get("walk_ms"), p.get("device_path_ms"), p.get("frac")))
line("c2", d)
for k,v in (d.get("workloads") or {}).items():
    if "roofline" in v: line(k, v)
    else: print(k, v)
print("value %.3f G/s host-incl %s" % (d["value"]/1e9, d["host_inclusive"]["page_locked_input_records_per_s"]))
print("fwm", d.get("fixed_window"))
PY
