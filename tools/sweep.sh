#!/bin/bash
# Diagnostic: kernel time for records-per-lane settings and phase ablations.
run() {
  env "$@" timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --cpu-sample 0 $BENCH_ARGS 2>/dev/null > /tmp/ab.json || exit 1
  python - "$*" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("%-28s kernel_ms %.4f" % (sys.argv[1], d["roofline"]["kernel_ms"]))
PY
}
for r in 1 2 4; do
  for a in 0 1 15; do
    run XE_FAST_R=$r XE_ABLATE=$a || exit 1
  done
done
run XE_FAST_R=2 XE_ABLATE=2
run XE_FAST_R=2 XE_ABLATE=4
run XE_FAST_R=2 XE_ABLATE=8
