#!/bin/bash
# Counter passes for the benchmark kernel, one rocprofv3 --pmc run per group
# (counters only, no tracing).  Usage: tools/prof_pmc.sh <outdir> [bench args]
set -u
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 bench.py --steps 3 --warmup 1 --quick --pipeline 0 --fwm 0 $BENCH_ARGS > "$out/$name.log" 2>&1
  echo "$name rc=$?"
}
BENCH_ARGS="$*"
run insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES || exit 1
run waits SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES || exit 1
run fetch TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum || exit 1
run write TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum || exit 1
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
run fetchsize FETCH_SIZE || exit 1
run writesize WRITE_SIZE || exit 1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "xe_" not in row["Kernel_Name"] or "probe" in row["Kernel_Name"]:
                continue
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
        for (kn, cn), (v, n) in sorted(acc.items()):
            print("%-24s %-28s per-launch %.6g (n=%d)" % (kn, cn, v / n, n))
PY
