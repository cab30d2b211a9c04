#!/bin/bash
# A/B of library builds without the test pass: bench.py --quick for c2, c1, c4 on the default
# build and on every xenoeye_amd/libxenoeye_gpu_<variant>.so.  Usage: tools/ab_quick.sh <tag> [variant ...]
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
for v in default "$@"; do
  if [ "$v" = default ]; then unset XE_GPU_LIB; else export XE_GPU_LIB=$PWD/xenoeye_amd/libxenoeye_gpu_$v.so; fi
  for w in c2 c1 c4; do
    timeout -k 10 300 python bench.py --workload $w --quick --steps 100 > "$out/bench_${v}_$w.json" 2> "$out/bench_${v}_$w.err" || { tail -20 "$out/bench_${v}_$w.err"; exit 1; }
    python - "$out/bench_${v}_$w.json" "$v" "$w" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%s %s %.3f G/s kernel %.4f ms frac %.4f" % (sys.argv[2], sys.argv[3], d["value"]/1e9, r["kernel_ms"], r["frac"]))
PY
  done
done
