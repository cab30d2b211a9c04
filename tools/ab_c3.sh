#!/bin/bash
# C3 (sFlow) kernel time for library variants. Usage: tools/ab_c3.sh <tag> variant...
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p "$out"
for v in "$@"; do
  if [ "$v" = default ]; then unset XE_GPU_LIB; else export XE_GPU_LIB=$PWD/xenoeye_amd/libxenoeye_gpu_$v.so; fi
  timeout -k 10 300 python bench.py --workload c3 --records 200000 --replicate 16 --quick > "$out/c3_$v.json" 2> "$out/c3_$v.err" || { tail -20 "$out/c3_$v.err"; exit 1; }
  python - "$out/c3_$v.json" "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%s c3 %.3f G/s kernel %.4f ms frac %.4f" % (sys.argv[2], d["value"]/1e9, r["kernel_ms"], r["frac"]))
PY
done
