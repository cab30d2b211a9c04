#!/bin/bash
# Diagnostic: time the benchmark kernel with phases switched off (XE_ABLATE bits).
for a in "$@"; do
  XE_ABLATE=$a timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --quick $BENCH_ARGS 2>/dev/null > /tmp/ab.json || exit 1
  python - "$a" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("ablate", sys.argv[1], "kernel_ms %.4f" % d["roofline"]["kernel_ms"])
PY
done
