"""Diagnostic: time the fixed-window kernel for single tables on the C2 batch."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import xenoeye_amd
from xenoeye_amd import synth

NREC = int(sys.argv[1]) if len(sys.argv) > 1 else (1 << 21)
w = synth.workload_c2(NREC)
packed = w["batch"].packed()
cases = [("all: proto,dst port", "", ["proto", "dst port", "octets", "packets"]),
         ("all: proto", "", ["proto", "octets"]),
         ("all: no key", "", ["octets"]),
         ("all: src host,dst host", "", ["src host", "dst host", "octets"]),
         ("obj0: src host,dst host", None, ["src host", "dst host", "bits"]),
         ("all: hosts + ports (12 bytes)", "", ["src host", "dst host", "src port", "dst port", "octets"]),
         ("all: src host6,dst host6,proto", "", ["src host6", "dst host6", "proto", "octets", "packets"])]
for name, filt, fields in cases:
    e = xenoeye_amd.Engine(device=0)
    for n, t in w["iplists"].items():
        e.add_iplist(n, t)
    for f, p in w["mos"]:
        e.add_mo("mo", f, p)
    mo = e.add_mo("mo", filt, -1) if filt is not None else 0
    t = e.add_fwm(mo, fields, 1 << 21 if "src host" in fields else 1 << 19)
    e.commit()
    b = e.prepare(packed)
    b.timing(True)
    b.run()
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        e.fwm_clear(t)
        b.run()
        ms.append(b.fwm_ms())
    warm = []
    for _ in range(3):
        b.run()            # keys already in the table
        warm.append(b.fwm_ms())
    rows, dr = e.fwm_read(t)
    print("%-32s rows %7d  empty table %.3f ms  filled table %.3f ms" % (name, len(rows), np.mean(ms), np.mean(warm)))
    e.close()
