#!/usr/bin/env python3
"""Diagnostic: how fast a plain one-address-per-lane walk of the merged list
tables runs on the benchmark's own address columns (time the
xe_trie_probe_kernel rows of `rocprofv3 --kernel-trace --stats`)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import xenoeye_amd  # noqa: E402
from xenoeye_amd import synth  # noqa: E402

w = synth.workload_c2(1 << 22)
eng = xenoeye_amd.Engine(0)
for n, t in w["iplists"].items():
    eng.add_iplist(n, t)
for f, p in w["mos"]:
    eng.add_mo("mo", f, p)
eng.commit()
b = eng.prepare(w["batch"])
b.run()
names = [f["name"] for f in xenoeye_amd.fields()]
fi = {n: i for i, n in enumerate(names)}
npars = None
src4 = b.column(fi["ip4_src_addr"])
dst4 = b.column(fi["ip4_dst_addr"])
src6 = b.column(fi["ip6_src_addr"])
# records of the other family hold zeros in these columns; keep the real ones
m4 = (src4 != 0).any(axis=1)
m6 = (src6 != 0).any(axis=1)
a4 = np.concatenate([src4[m4], dst4[m4]])
a6 = np.concatenate([src6[m6], b.column(fi["ip6_dst_addr"])[m6]])
print("v4 addresses", len(a4), "v6 addresses", len(a6))
for i in range(3):
    r4 = eng.iplist_match("list0", 4, a4)
    r6 = eng.iplist_match("list0", 6, a6)
print("hits", int(r4.sum()), int(r6.sum()))
