#!/bin/bash
# fixed-window table kernel time for library variants. Usage: tools/ab_fwm.sh <tag> variant...
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p "$out"
for v in "$@"; do
  if [ "$v" = default ]; then unset XE_GPU_LIB; else export XE_GPU_LIB=$PWD/xenoeye_amd/libxenoeye_gpu_$v.so; fi
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --cpu-procs 0 --others 0 --sustain-seconds 0 --pipeline 0 > "$out/fwm_$v.json" 2> "$out/fwm_$v.err" || { tail -20 "$out/fwm_$v.err"; exit 1; }
  python - "$out/fwm_$v.json" "$v" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); f=d["fixed_window"]
print("%s fwm %.4f ms (first batch %.4f) rows %s dropped %s" % (sys.argv[2], f["kernel_ms"], f["kernel_ms_first_batch_of_window"], [t["rows"] for t in f["tables"]], [t["dropped"] for t in f["tables"]]))
PY
done
