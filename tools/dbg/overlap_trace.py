#!/usr/bin/env python3
"""Where does one host thread spend a batch's time in overlap mode?  XE_DEBUG=1 prints prepare's phases."""
import os, sys, time
os.environ["XE_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import xenoeye_amd
from xenoeye_amd import synth

w = synth.workload_c2(1 << 22, seed=1)
eng = xenoeye_amd.Engine(device=0)
for name, text in w["iplists"].items():
    eng.add_iplist(name, text)
for filt, parent in w["mos"]:
    eng.add_mo("mo", filt, parent)
eng.commit()
packed = w["batch"].packed()
ins = [(eng.pinned(packed[0]),) + tuple(packed[1:]) for _ in range(2)]
b = eng.prepare(ins[0]); b.run(); b.counters()
mb = [eng.pinned_array((b.records * b.mask_words,), np.uint64) for _ in range(2)]
b.free()
eng.set_overlap(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
prev = None
T0 = time.perf_counter()
for k in range(int(os.environ.get("ROUNDS", "6"))):
    t0 = time.perf_counter()
    cur = eng.prepare(ins[k & 1])
    t1 = time.perf_counter()
    cur.run()
    t2 = time.perf_counter()
    if prev is not None:
        if not os.environ.get("NOMASKS"):
            prev.masks(out=mb[(k - 1) & 1])
        t3 = time.perf_counter()
        prev.counters()
        t4 = time.perf_counter()
        prev.free()
        t5 = time.perf_counter()
        print("round %d: prepare %.2f run %.2f masks %.2f counters %.2f free %.2f ms" % (k, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3), file=sys.stderr)
    prev = cur
prev.free()
print("avg cycle %.3f ms" % ((time.perf_counter() - T0) * 1e3 / int(os.environ.get("ROUNDS", "6"))), file=sys.stderr)
