#!/bin/bash
# Diagnostic: per-phase wave cycles of the fast kernel (XE_ABLATE bit 64) for ablation values.
# Usage: BENCH_ARGS="--workload c2" tools/phase.sh <values...>
for a in "$@"; do
  XE_ABLATE=$((a | 64)) timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --quick $BENCH_ARGS 2>&1 >/dev/null | grep "phase cycles" | tail -1 | sed "s/^/ablate $a: /" || exit 1
done
