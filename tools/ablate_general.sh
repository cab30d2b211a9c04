#!/bin/bash
# Where the per-lane Netflow kernel spends its time: C2 forced through it with parts switched off
# (results are wrong with any part off).  Usage: tools/ablate_general.sh <tag>
set -u
out=gpurun_out/$1
mkdir -p "$out"
for ab in 0 256 512 768; do
  XE_ABLATE=$ab XE_FORCE_GENERAL=1 timeout -k 10 200 python bench.py --quick --steps 20 > "$out/gen_$ab.json" 2> "$out/gen_$ab.err" || { tail "$out/gen_$ab.err"; exit 1; }
  python - "$out/gen_$ab.json" $ab <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("ablate %s: kernel %s %.4f ms" % (sys.argv[2], r["kernel"], r["kernel_ms"]))
PY
done
