import struct, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import golden_io, xenoeye_amd
from xenoeye_amd import synth
texts, p4, p6, want4, want6 = golden_io.iplist_reference()
for rb in (16, 24):
    eng = xenoeye_amd.Engine(device=0)
    eng.set_list_root_bits(rb)
    for i, t in enumerate(texts):
        eng.add_iplist("l%d" % i, t)
    for i in range(len(texts)):
        eng.add_mo("d%d" % i, "dst net6 l%d" % i)
    eng.commit()
    n = len(p6)
    t = [(28, 16), (4, 1)]
    rec = np.zeros((n, 17), dtype=np.uint8)
    rec[:, 0:16] = p6[:n]
    rec[:, 16] = 6
    exp = bytes([192, 0, 2, 9])
    b = synth.Batch()
    b.add(synth.ipfix_template_packet(400, t, 5), exp)
    for i in range(0, n, 60):
        body = rec[i:i + 60].tobytes()
        body = struct.pack(">HH", 400, 4 + len(body)) + body
        b.add(synth.ipfix_header(16 + len(body), 5) + body, exp)
    out = eng.prepare(b); out.run()
    m = out.masks()[:, 0]
    fi = out.flow_info()
    proto = [f for f in xenoeye_amd.fields() if f["name"] == "protocol"][0]
    has_proto = fi[:, ((proto["off"] + proto["max"] + 3) & ~3) + 4]
    sel = np.nonzero(has_proto)[0]
    print("root", rb, "records", len(sel), n)
    for i in range(len(texts)):
        got = ((m[sel] >> np.uint64(i)) & np.uint64(1)).astype(np.uint8)
        bad = np.nonzero(got != want6[i][:n])[0]
        print(" list", i, "bad", len(bad), [(bytes(p6[j]).hex(), int(got[j]), int(want6[i][j])) for j in bad[:6]])
        dev = eng.iplist_match("l%d" % i, 6, p6)
        print("   probe kernel bad", int((dev != want6[i]).sum()))
    eng.close()
