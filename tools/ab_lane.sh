#!/bin/bash
# per-lane kernels (sFlow, Netflow per-lane) on the default build and a variant.  Usage: tools/ab_lane.sh <tag> <variant>
set -u
out=gpurun_out/$1
mkdir -p "$out"
for v in default "$2"; do
  if [ "$v" = default ]; then unset XE_GPU_LIB; else export XE_GPU_LIB=$PWD/xenoeye_amd/libxenoeye_gpu_$v.so; fi
  timeout -k 10 300 python bench.py --quick --steps 50 --workload c3 --records 200000 --replicate 16 > "$out/${v}_c3.json" 2> "$out/${v}_c3.err" || { tail "$out/${v}_c3.err"; exit 1; }
  XE_FORCE_GENERAL=1 timeout -k 10 300 python bench.py --quick --steps 50 > "$out/${v}_gen.json" 2> "$out/${v}_gen.err" || { tail "$out/${v}_gen.err"; exit 1; }
  python - "$out" "$v" <<'PY'
import json,sys
for w in ("c3", "gen"):
    d=json.load(open("%s/%s_%s.json" % (sys.argv[1], sys.argv[2], w))); r=d["roofline"]
    print("%s %s kernel %s %.4f ms frac %.4f" % (sys.argv[2], w, r["kernel"], r["kernel_ms"], r["frac"]))
PY
done
