import struct, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import xenoeye_amd
from xenoeye_amd import synth
from oracle_binding import Oracle
from helpers import configure

w = synth.workload_c3(3000, seed=77)
rng = synth.Rng(1234)
b = synth.Batch()
for p, s in zip(w["batch"].pkts, w["batch"].src):
    p = bytearray(p)
    k = int(rng.below(1, 6)[0])
    if k == 0:
        p = p[:int(rng.below(1, len(p))[0])]
    elif k == 1:
        for _ in range(3):
            p[int(rng.below(1, len(p))[0])] = int(rng.below(1, 256)[0])
    elif k == 2:
        at = 4 * int(rng.below(1, len(p) // 4)[0])
        p[at:at + 4] = struct.pack(">I", int(rng.below(1, 1 << 32)[0]))
    elif k == 3:
        p += bytes(rng.bytes(int(rng.below(1, 64)[0])))
    b.add(bytes(p), s, kind=1)
orc = Oracle()
res = {}
for mode in (3, 0):
    eng = xenoeye_amd.Engine(device=0)
    eng.set_device_walk(mode)
    configure(eng, orc if mode == 3 else None, w["iplists"], w["mos"]) if mode == 3 else None
    if mode == 0:
        for name, text in w["iplists"].items():
            eng.add_iplist(name, text)
        for filt, parent in w["mos"]:
            eng.add_mo("mo", filt, parent)
        eng.commit()
    pb = eng.prepare(b)
    pb.run()
    dd = pb.dropped()
    dg = pb.record_datagram()
    per = np.bincount(dg[dd != 2], minlength=len(b.pkts))
    allr = np.bincount(dg, minlength=len(b.pkts))
    res[mode] = (per, allr)
    print("mode", mode, "records", pb.records, "delivered", int((dd != 2).sum()))
orc.log_config(keep_flows=False)
oper = []
for p, s in zip(b.pkts, b.src):
    n0 = orc.nrecords()
    orc.sflow(p, s)
    oper.append(orc.nrecords() - n0)
oper = np.array(oper)
print("oracle", oper.sum())
for i in range(len(b.pkts)):
    if not (res[3][0][i] == res[0][0][i] == oper[i]) or res[3][1][i] != res[0][1][i]:
        print("datagram", i, "len", len(b.pkts[i]), "dev", res[3][0][i], res[3][1][i], "host", res[0][0][i], res[0][1][i], "oracle", oper[i])
        print(b.pkts[i][:200].hex())
