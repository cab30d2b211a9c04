#!/bin/bash
# parity of the device-rule paths, then C2 with and without devices.conf mark filters
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/marks
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "device or tfstr or shim or fuzz or walk" > gpurun_out/marks/pytest.log 2>&1 || { tail -30 gpurun_out/marks/pytest.log; exit 1; }
tail -2 gpurun_out/marks/pytest.log
for w in c2 c2m; do
  timeout -k 10 300 python bench.py --workload $w --quick > gpurun_out/marks/$w.json 2> gpurun_out/marks/$w.err || { tail -20 gpurun_out/marks/$w.err; exit 1; }
  python - gpurun_out/marks/$w.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
r=d["roofline"]
print(d["config"]["workload"][:60], "kernel", r["kernel"], "%.4f ms frac %.3f" % (r["kernel_ms"], r["frac"]), "device path", d["device_path"]["device_path_ms"], "walked on device", d["host_inclusive"]["headers_walked_on_device"])
PY
done
