#!/bin/bash
# Benchmark lines of the workloads beside the default one.  Usage: tools/gpu_others.sh <tag>
set -u
tag=${1:-others}
out=gpurun_out/$tag
mkdir -p "$out"
run() {
  name=$1; shift
  echo "== $name"
  timeout -k 10 500 python bench.py "$@" > "$out/bench_$name.json" 2> "$out/bench_$name.err" || { tail -20 "$out/bench_$name.err"; exit 1; }
  python - "$out/bench_$name.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; h = d.get("host_inclusive") or {}
print("%.3f G/s kernel %.4f ms frac %.4f cpu %s host-inclusive page-locked %s" % (d["value"] / 1e9, r["kernel_ms"], r["frac"],
      (d.get("cpu_baseline") or {}).get("value"), h.get("page_locked_input_records_per_s")))
PY
}
run c3 --workload c3 --records 200000 --replicate 16 --cpu-sample 2000000 --pipeline 0 --fwm 0
run c5 --workload c5 --cpu-sample 4000000 --pipeline 0
run c4 --workload c4 --records 2097152 --cpu-sample 100000 --pipeline 0 --fwm 0
run c4_256lists --workload c4 --records 500000 --c4-lists 256 --c4-prefixes 30000 --cpu-sample 20000 --pipeline 0 --fwm 0
