#!/bin/bash
# Timeline of the header walk of one workload: every kernel and copy of the prepare calls, with gaps.
# Usage: tools/trace_walk.sh <tag> <workload>
set -u
tag=$1; w=$2
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--workload $w --quick --steps 3 --warmup 1 --walk-reps 3"
[ "$w" = c3 ] && args="--workload c3 --records 200000 --replicate 16 --quick --steps 3 --warmup 1 --walk-reps 3"
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out/tl_$w" -- python3 bench.py $args > "$out/tl_$w.json" 2> "$out/tl_$w.log" || { tail -20 "$out/tl_$w.log"; exit 1; }
python3 - "$out/tl_$w" <<'PY'
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# the last stretch that contains a shape / count kernel
idx = [i for i, e in enumerate(ev) if "xw_nf_shape" in e[2] or "xw_sf_count" in e[2]]
if idx:
    i0 = max(0, idx[-1] - 6)
    t0 = ev[i0][0]
    prev = None
    for s, e, n in ev[i0:i0 + 40]:
        print("%9.1f us  +%7.1f  dur %8.1f  %s" % ((s - t0) / 1e3, 0 if prev is None else (s - prev) / 1e3, (e - s) / 1e3, n))
        prev = e
PY
