#!/bin/bash
# Latency-side counters of the benchmark kernel: cycles memory instructions
# stay in flight, instruction fetch, LDS conflicts.  Usage: tools/pmc_lat.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 2 --warmup 1 --quick ${BENCH_ARGS:-} > "$out/$name.log" 2>&1 || { tail -5 "$out/$name.log"; exit 1; }
}
run l1 SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_IFETCH_LEVEL
run l2 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run l3 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if "xe_nf" not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
    for cn, (v, n) in sorted(acc.items()):
        print("%-32s per-launch %.6g" % (cn, v / n))
PY
