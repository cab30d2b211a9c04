#!/bin/bash
# Instruction mix and wait counters of one kernel of a bench run.
# Usage: KERNEL=xe_sf BENCH_ARGS="--workload c3 --records 200000 --replicate 16" tools/pmc_kernel.sh <tag>
set -u
tag=$1
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 2 --warmup 1 ${QUICK---quick} ${BENCH_ARGS:-} > "$out/$name.log" 2>&1 || { tail -5 "$out/$name.log"; exit 1; }
}
run i1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH
run i2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT
run i3 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE
python3 - "$out" "${KERNEL:-xe_}" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if kern not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
    for cn, (v, n) in sorted(acc.items()):
        print("%-32s per-launch %.6g" % (cn, v / n))
PY
