#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export ROUNDS=12
mkdir -p gpurun_out/ovlprof
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ovlprof -- python3 tools/dbg/overlap_trace.py 1 > gpurun_out/ovlprof/log.txt 2>&1 || { tail -5 gpurun_out/ovlprof/log.txt; exit 1; }
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/ovlprof/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(("copy "+r["Direction"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for f in glob.glob("gpurun_out/ovlprof/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0][-40:]
        rows.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda x:x[1])
big=[r for r in rows if r[2]-r[1] > 100000]
t0=big[-60][1] if len(big)>60 else big[0][1]
for n,s,e in big[-60:]:
    print("%-45s start %9.3f ms  dur %7.3f ms" % (n,(s-t0)/1e6,(e-s)/1e6))
PY
