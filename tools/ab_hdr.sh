#!/bin/bash
# With and without node headers (XE_NO_NODE_HEADERS).  Usage: tools/ab_hdr.sh <tag>
set -u
out=gpurun_out/$1
mkdir -p "$out"
timeout -k 10 700 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
tail -2 "$out/pytest.log"
for v in with without; do
  if [ "$v" = with ]; then unset XE_NO_NODE_HEADERS; else export XE_NO_NODE_HEADERS=1; fi
  for w in c2 c1 c5 c4; do
    args="--workload $w --quick --steps 100"
    [ "$w" = c4 ] && args="--workload c4 --records 2097152 --quick --steps 100"
    timeout -k 10 300 python bench.py $args > "$out/bench_${v}_$w.json" 2> "$out/bench_${v}_$w.err" || { tail -20 "$out/bench_${v}_$w.err"; exit 1; }
    python - "$out/bench_${v}_$w.json" "$v" "$w" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%s headers %s %.3f G/s kernel %.4f ms frac %.4f" % (sys.argv[2], sys.argv[3], d["value"]/1e9, r["kernel_ms"], r["frac"]))
PY
  done
done
