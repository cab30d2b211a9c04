#!/bin/bash
# bench.py --quick on several library builds, no tests.  Usage: tools/ab_bench.sh <tag> "<workloads>" variant...
set -u
tag=$1; wls=$2; shift 2
out=gpurun_out/$tag
mkdir -p "$out"
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = default ]; then unset XE_GPU_LIB; else export XE_GPU_LIB=$PWD/xenoeye_amd/libxenoeye_gpu_$v.so; fi
  for w in $wls; do
    timeout -k 10 300 python bench.py --workload $w --quick > "$out/bench_${v}_$w.json" 2> "$out/bench_${v}_$w.err" || { tail -20 "$out/bench_${v}_$w.err"; exit 1; }
    python - "$out/bench_${v}_$w.json" "$v" "$w" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%s %s %.3f G/s kernel %.4f ms frac %.4f" % (sys.argv[2], sys.argv[3], d["value"]/1e9, r["kernel_ms"], r["frac"]))
PY
  done
done
done
