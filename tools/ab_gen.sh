#!/bin/bash
# per-lane Netflow kernel (C2 forced through it) on the default build and variants.  Usage: tools/ab_gen.sh <tag> [variant ...]
set -u
out=gpurun_out/$1; shift
mkdir -p "$out"
for v in default "$@"; do
  if [ "$v" = default ]; then unset XE_GPU_LIB; else export XE_GPU_LIB=$PWD/xenoeye_amd/libxenoeye_gpu_$v.so; fi
  XE_FORCE_GENERAL=1 timeout -k 10 300 python bench.py --quick --steps 50 > "$out/${v}_gen.json" 2> "$out/${v}_gen.err" || { tail "$out/${v}_gen.err"; exit 1; }
  python - "$out/${v}_gen.json" "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%s kernel %s %.4f ms frac %.4f" % (sys.argv[2], r["kernel"], r["kernel_ms"], r["frac"]))
PY
done
