#!/bin/bash
# Instruction-mix and wait counters of the benchmark kernel (two --pmc passes).
set -u
out=gpurun_out/${1:-pmc_insts}; shift || true
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d "$out/insts" -- python3 bench.py --steps 3 --warmup 1 --quick "$@" > "$out/insts.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$out/waits" -- python3 bench.py --steps 3 --warmup 1 --quick "$@" > "$out/waits.log" 2>&1 || exit 1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if "xe_" not in row["Kernel_Name"] or "probe" in row["Kernel_Name"]:
            continue
        k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    for (kn, cn), (v, n) in sorted(acc.items()):
        print("%-24s %-28s per-launch %.6g (n=%d)" % (kn, cn, v / n, n))
PY
