#!/bin/bash
# Pipelined host-inclusive rate for several thread counts.  Usage: tools/pipe_bench.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
for cfg in "2 1" "3 1" "4 1" "2 4" "4 4" "6 1"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --cpu-sample 0 --cpu-procs 0 --others 0 --sustain-seconds 0 --fwm 0 --steps 20 --pipeline $1 --pipeline-walkers $2 --pipeline-rounds 24 > "$out/pipe_$1_$2.json" 2> "$out/pipe_$1_$2.err" || { tail -20 "$out/pipe_$1_$2.err"; exit 1; }
  python - "$out/pipe_$1_$2.json" $1 $2 <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); h=d["host_inclusive"]
print("threads %s walkers %s: pipelined %.1f M/s (single pinned %.1f M/s)" % (sys.argv[2], sys.argv[3], h["pipelined"]["records_per_s"]/1e6, h["page_locked_input_records_per_s"]/1e6))
PY
done
