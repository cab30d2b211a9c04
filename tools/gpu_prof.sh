#!/bin/bash
# Kernel-trace statistics and counter passes only (the profiling half of gpu_round.sh),
# plus a kernel trace of the bench with its fixed-window section.  Usage: tools/gpu_prof.sh <tag>
set -u
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c2" -- python3 bench.py --quick --pipeline 0 --fwm 0 > "$out/trace_c2.log" 2>&1 || { tail -20 "$out/trace_c2.log"; exit 1; }
find "$out/trace_c2" -name "*kernel_stats.csv" -exec cat {} \;
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_fwm" -- python3 bench.py --quick --pipeline 0 --steps 3 --warmup 1 > "$out/trace_fwm.log" 2>&1 || { tail -20 "$out/trace_fwm.log"; exit 1; }
find "$out/trace_fwm" -name "*kernel_stats.csv" -exec cat {} \;
bash tools/prof_pmc.sh "$out/pmc_c2" > "$out/pmc_c2.txt" 2>&1 || { tail -20 "$out/pmc_c2.txt"; exit 1; }
cat "$out/pmc_c2.txt"
