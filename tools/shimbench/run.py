"""Writes a workload for tools/shimbench/shim_bench.c and runs it.
Usage: python tools/shimbench/run.py <workload c1|c2|c3> <records> <passes> [overlap] [max_datagrams]"""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, ".")
from xenoeye_amd import synth

name, nrec, passes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
extra = sys.argv[4:]
w = getattr(synth, "workload_" + name)(nrec)
d = "/tmp/shimbench_" + name
os.makedirs(d + "/iplists", exist_ok=True)
for n, t in (w.get("iplists") or {}).items():
    open(os.path.join(d, "iplists", n), "w").write(t)
with open(d + "/mos.txt", "w") as f:
    for filt, parent in w["mos"]:
        f.write("%d\t%s\n" % (parent, filt))
data, offs, src, kinds = w["batch"].packed()[:4]
with open(d + "/batch.bin", "wb") as f:
    f.write(np.uint64(len(src)).tobytes())
    f.write(np.asarray(offs, dtype=np.uint64).tobytes())
    f.write(np.asarray(src, dtype=np.uint32).tobytes())
    f.write((np.asarray(kinds, dtype=np.uint8) if kinds is not None else np.zeros(len(src), np.uint8)).tobytes())
    f.write(np.asarray(data, dtype=np.uint8).tobytes())
here = os.path.dirname(os.path.abspath(__file__))
for ov in ([extra[0]] if extra else ["0", "1"]):
    r = subprocess.run([os.path.join(here, "shim_bench"), d, str(passes), ov] + extra[1:], capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr.strip())
