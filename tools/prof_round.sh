#!/bin/bash
# Kernel trace + HBM-side traffic counters for every configuration.
# Usage: tools/prof_round.sh <tag>   -> gpurun_out/<tag>/<workload>_{kernel_stats.csv,pmc.txt}
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in c2 c1 c3 c4; do
  args="--workload $w --quick"
  [ "$w" = c3 ] && args="--workload c3 --records 200000 --replicate 16 --quick"
  [ "$w" = c4 ] && args="--workload c4 --records 2097152 --quick"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$w" -- python3 bench.py --steps 20 --warmup 5 $args > "$out/bench_$w.json" 2> "$out/trace_$w.log" || { tail -20 "$out/trace_$w.log"; exit 1; }
  find "$out/trace_$w" -name "*kernel_stats.csv" -exec cp {} "$out/${w}_kernel_stats.csv" \;
  head -4 "$out/${w}_kernel_stats.csv"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 400 rocprofv3 --pmc $c --output-format csv -d "$out/pmc_${w}_$c" -- python3 bench.py --steps 3 --warmup 1 $args > "$out/pmc_${w}_$c.log" 2>&1 || { tail -5 "$out/pmc_${w}_$c.log"; exit 1; }
  done
  python3 - "$out" "$w" > "$out/${w}_pmc.txt" <<'PY'
import csv, glob, sys, collections
out, w = sys.argv[1], sys.argv[2]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, w, c), recursive=True):
        for row in csv.DictReader(open(f)):
            if ("xe_" not in row["Kernel_Name"] and "xw_" not in row["Kernel_Name"]) or "probe" in row["Kernel_Name"]:
                continue
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    for (kn, cn), (v, n) in sorted(acc.items()):
        print("%-40s %-12s per-launch %.6g (n=%d)" % (kn, cn, v / n, n))
PY
  cat "$out/${w}_pmc.txt"
done
# the fixed-window table kernel on the C2 batch (bench's table leg only)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_fwm" -- python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --cpu-procs 0 --others 0 --sustain-seconds 0 --pipeline 0 --shim-records 0 --overlap-leg 0 > "$out/bench_fwm.json" 2> "$out/trace_fwm.log" || { tail -20 "$out/trace_fwm.log"; exit 1; }
find "$out/trace_fwm" -name "*kernel_stats.csv" -exec cp {} "$out/fwm_kernel_stats.csv" \;
grep -E "xe_fwm|Name" "$out/fwm_kernel_stats.csv" | cut -c1-200
