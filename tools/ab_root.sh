#!/bin/bash
# list-table root width: 16 against 24 address bits, on the list-heavy workloads.  Usage: tools/ab_root.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "golden or fuzz or scale" > "$out/pytest.log" 2>&1 || { tail -40 "$out/pytest.log"; exit 1; }
tail -2 "$out/pytest.log"
for w in c4 c2 c5; do
  for rb in 16 24; do
    args="--workload $w --quick --root-bits $rb --walk-reps 0"
    [ "$w" = c4 ] && args="--workload c4 --records 2097152 --quick --root-bits $rb --walk-reps 0"
    timeout -k 10 400 python bench.py $args > "$out/bench_${w}_$rb.json" 2> "$out/bench_${w}_$rb.err" || { tail -20 "$out/bench_${w}_$rb.err"; exit 1; }
    python - "$out/bench_${w}_$rb.json" $w $rb <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print("%s root %s: kernel %.4f ms frac %.4f tables %s MB" % (sys.argv[2], sys.argv[3], r["kernel_ms"], r["frac"], (r.get("list_tables") or {}).get("bytes", 0) // (1 << 20)))
PY
  done
done
