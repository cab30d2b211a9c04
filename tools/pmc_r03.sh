#!/bin/bash
# Wave-level and memory-pipeline counters of the decode kernels AND the walk kernels, per workload.
# Usage: tools/pmc_r03.sh <tag> <workloads...>   -> gpurun_out/<tag>/<workload>_wave_pmc.txt
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in "$@"; do
  args="--workload $w --quick"
  [ "$w" = c3 ] && args="--workload c3 --records 200000 --replicate 16 --quick"
  [ "$w" = c4 ] && args="--workload c4 --records 2097152 --quick"
  run() {
    name=$1; shift
    timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$out/${w}_$name" -- python3 bench.py --steps 2 --warmup 1 --walk-reps 2 $args > "$out/${w}_$name.log" 2>&1 || { tail -5 "$out/${w}_$name.log"; exit 1; }
  }
  run p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES
  run p2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM
  run p3 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum
  run p4 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
  python3 - "$out" "$w" > "$out/${w}_wave_pmc.txt" <<'PY'
import csv, glob, sys, collections
out, w = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("%s/%s_p*/**/*counter_collection.csv" % (out, w), recursive=True)):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        if ("xe_" not in kn and "xw_" not in kn) or "probe" in kn:
            continue
        k = (kn.split("(")[0].replace("void ", ""), row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (v, n) in sorted(acc.items()):
    print("%-34s %-30s per-launch %.6g (n=%d)" % (kn, cn, v / n, n))
PY
  grep -c . "$out/${w}_wave_pmc.txt"
done
