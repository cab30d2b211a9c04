#!/bin/bash
# One GPU-box session: parity tests, benchmark lines, kernel-trace statistics
# and counter passes.  Every GPU step has its own time limit; the first failure
# ends the script.  Usage: tools/gpu_round.sh <tag>
set -u
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== tests"
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$out/pytest.log" 2>&1 || { tail -30 "$out/pytest.log"; exit 1; }
tail -3 "$out/pytest.log"
echo "== bench c2"
timeout -k 10 400 python bench.py > "$out/bench_c2.json" 2> "$out/bench_c2.err" || { tail -20 "$out/bench_c2.err"; exit 1; }
cat "$out/bench_c2.json"
echo "== bench c1"
timeout -k 10 300 python bench.py --workload c1 --cpu-sample 4000000 > "$out/bench_c1.json" 2> "$out/bench_c1.err" || { tail -20 "$out/bench_c1.err"; exit 1; }
cat "$out/bench_c1.json"
echo "== kernel trace"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_c2" -- python3 bench.py --quick --pipeline 0 --fwm 0 > "$out/trace_c2.log" 2>&1 || { tail -20 "$out/trace_c2.log"; exit 1; }
find "$out/trace_c2" -name "*kernel_stats.csv" -exec cat {} \;
echo "== counters"
bash tools/prof_pmc.sh "$out/pmc_c2" > "$out/pmc_c2.txt" 2>&1 || { tail -20 "$out/pmc_c2.txt"; exit 1; }
cat "$out/pmc_c2.txt"
