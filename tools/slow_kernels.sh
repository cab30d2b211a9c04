#!/bin/bash
# The kernels beside the fast one: per-lane Netflow kernel (C2 forced through it),
# C4 over 256 distinct lists, sFlow.  Usage: tools/slow_kernels.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
line() {
  python - "$1" "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%s: %.3f G/s kernel %s %.4f ms frac %.4f" % (sys.argv[2], d["value"]/1e9, r["kernel"], r["kernel_ms"], r["frac"]))
PY
}
XE_FORCE_GENERAL=1 timeout -k 10 300 python bench.py --quick --steps 20 > "$out/general.json" 2> "$out/general.err" || { tail "$out/general.err"; exit 1; }
line "$out/general.json" "C2 per-lane kernel"
timeout -k 10 300 python bench.py --quick --steps 20 --workload c4 --records 500000 --c4-lists 256 --c4-prefixes 30000 > "$out/c4_256.json" 2> "$out/c4_256.err" || { tail "$out/c4_256.err"; exit 1; }
line "$out/c4_256.json" "C4 256 lists"
timeout -k 10 300 python bench.py --quick --steps 20 --workload c3 --records 200000 --replicate 16 > "$out/c3.json" 2> "$out/c3.err" || { tail "$out/c3.err"; exit 1; }
line "$out/c3.json" "C3"
