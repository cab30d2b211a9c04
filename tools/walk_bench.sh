#!/bin/bash
# Host-inclusive rates with and without the header walk on the device.
# Usage: tools/walk_bench.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
for mode in 3 0; do
  for w in c2 c3 c1; do
    args="--workload $w --quick --pipeline 0 --fwm 0"
    [ "$w" = c3 ] && args="--workload c3 --records 200000 --replicate 16 --quick --pipeline 0 --fwm 0"
    XE_DEVICE_WALK=$mode XE_DEBUG=1 timeout -k 10 300 python bench.py $args > "$out/bench_${w}_walk$mode.json" 2> "$out/bench_${w}_walk$mode.err" || { tail -20 "$out/bench_${w}_walk$mode.err"; exit 1; }
    python - "$out/bench_${w}_walk$mode.json" "$w" "$mode" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; h=d["host_inclusive"]
print("%s walk=%s kernel %.4f ms frac %.4f | host-incl %.1f M/s pinned %.1f M/s pinned-1thread %.1f M/s prepare %.2f ms walked %s" % (
    sys.argv[2], sys.argv[3], r["kernel_ms"], r["frac"], h["records_per_s"]/1e6, h["page_locked_input_records_per_s"]/1e6,
    h["page_locked_input_one_host_thread_records_per_s"]/1e6, h["prepare_s"]*1e3, h["headers_walked_on_device"]))
PY
    grep "^prepare:" "$out/bench_${w}_walk$mode.err" | tail -2
  done
done
