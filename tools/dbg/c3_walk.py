"""Diagnostic: the sFlow header walk of C3, rep by rep (device time by HIP events)."""
import sys
sys.path.insert(0, ".")
import torch
import bench
import xenoeye_amd

class A: c4_prefixes = 1000000; c4_lists = 8
w, _ = bench.build_workload("c3", 200000, 0, A.c4_prefixes, 1, A.c4_lists)
bench.replicate(w["batch"], 16)
eng = bench.make_engine(w, 0)
packed = w["batch"].packed()
eng.prepare(packed).free()
b = eng.prepare(packed); b.run(); torch.cuda.synchronize()
eng.set_walk_timing(True)
for k in range(12):
    if k == 6:
        for _ in range(50):
            b.run()
        torch.cuda.synchronize()
    bb = eng.prepare(packed)
    print("rep %d walk %.4f ms" % (k, bb.walk_ms()))
    bb.free()
