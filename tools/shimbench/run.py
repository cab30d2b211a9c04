"""Writes a workload for tools/shimbench/shim_bench.c and runs it (this process never touches the GPU).
Usage: python tools/shimbench/run.py <workload c1|c2|c3> <records> <passes> [overlap [capture threads [max_datagrams per batch]]]"""
import os, subprocess, sys
sys.path.insert(0, ".")
from xenoeye_amd import synth

name, nrec, passes = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
extra = sys.argv[4:]
w = getattr(synth, "workload_" + name)(nrec)
d = "/tmp/shimbench_" + name
synth.write_workload_dir(w, d)
here = os.path.dirname(os.path.abspath(__file__))
for ov in ([extra[0]] if extra else ["0", "1"]):
    r = subprocess.run([os.path.join(here, "shim_bench"), d, str(passes), ov] + extra[1:], capture_output=True, text=True)
    print(r.stdout.strip())
    if r.stderr.strip():
        print(r.stderr.strip())
