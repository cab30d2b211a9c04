#!/bin/bash
# Kernel trace of one workload (walk launches included).  Usage: tools/prof_one.sh <tag> <workload> [extra bench args]
set -u
tag=$1; w=$2; shift 2
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
args="--workload $w --quick"
[ "$w" = c3 ] && args="--workload c3 --records 200000 --replicate 16 --quick"
[ "$w" = c4 ] && args="--workload c4 --records 2097152 --quick"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_$w" -- python3 bench.py --steps 20 --warmup 5 $args "$@" > "$out/bench_$w.json" 2> "$out/trace_$w.log" || { tail -20 "$out/trace_$w.log"; exit 1; }
find "$out/trace_$w" -name "*kernel_stats.csv" -exec cp {} "$out/${w}_kernel_stats.csv" \;
cut -d, -f1-5 "$out/${w}_kernel_stats.csv" | head -24
