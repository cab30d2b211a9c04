#!/bin/bash
# sFlow kernel and walk times at several batch sizes (does the batch fit the Infinity Cache?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2 4 8 16; do
  timeout -k 10 200 python bench.py --workload c3 --records 200000 --replicate $rep --quick > gpurun_out/c3s_$rep.json 2> gpurun_out/c3s_$rep.err || { tail -5 gpurun_out/c3s_$rep.err; exit 1; }
  python - gpurun_out/c3s_$rep.json $rep <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]; n=d["config"]["records_per_gpu_per_step"]
print("replicate %s: %d samples, kernel %.4f ms (%.1f ns/k-sample), walk %.4f ms" % (sys.argv[2], n, r["kernel_ms"], r["kernel_ms"]*1e6/n*1e-3*1e3, d["device_path"]["walk_ms"]))
PY
done
