#!/bin/bash
# Instruction counts of the benchmark kernel with phases switched off.
# Usage: tools/pmc_phase.sh <tag> <ablate values...>
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for a in "$@"; do
  export XE_ABLATE=$a
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d "$out/a$a" -- python3 bench.py --steps 2 --warmup 1 --quick ${BENCH_ARGS:-} > "$out/a$a.log" 2>&1 || exit 1
  python3 - "$out/a$a" "$a" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "xe_nf" not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
print("ablate %-3s" % sys.argv[2], "  ".join("%s %.1f" % (k.replace("SQ_INSTS_", ""), v / n / 65536) for k, (v, n) in sorted(acc.items())), "(per 64 records)")
PY
done
