#!/bin/bash
# Diagnostic: kernel time for records-per-lane settings and phase ablations.
# Usage: tools/sweep.sh "VAR=val VAR=val" ...   (one run per argument)
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --quick $BENCH_ARGS 2>/dev/null > /tmp/ab.json || exit 1
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print("%-36s kernel_ms %.4f" % (sys.argv[1], d["roofline"]["kernel_ms"]))
PY
done
