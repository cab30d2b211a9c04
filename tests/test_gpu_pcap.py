"""A capture file replayed in several batches through the device path gives
what the original datagrams give in one batch, and what the oracle gives."""
import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import compare, configure, run_oracle
from oracle_binding import Oracle

pytestmark = pytest.mark.gpu


def test_replay_in_pieces_equals_one_batch_and_the_oracle(tmp_path):
    w = synth.workload_c2(20000, seed=11)
    path = str(tmp_path / "c2.pcap")
    synth.batch_to_pcap(path, w["batch"])
    eng, orc = xenoeye_amd.Engine(device=0), Oracle()
    configure(eng, orc, w["iplists"], w["mos"])
    whole = eng.prepare(w["batch"])
    whole.run()
    run_oracle(orc, w["batch"])
    compare(whole, orc)

    eng2 = xenoeye_amd.Engine(device=0)
    for name, text in w["iplists"].items():   # (a second Oracle() would reset the first: it is one global state)
        eng2.add_iplist(name, text)
    for filt, parent in w["mos"]:
        eng2.add_mo("mo", filt, parent)
    eng2.commit()
    masks, flows, counters = [], [], 0
    reader = xenoeye_amd.PcapReader(path, udp_port=2055, max_datagrams=97)
    for piece in reader:
        b = eng2.prepare(piece)
        b.run()
        masks.append(b.masks().copy())
        flows.append(b.flow_info().copy())
        counters = counters + b.counters()
        b.free()
    assert len(masks) > 3 and reader.stats()["taken"] == len(w["batch"].pkts)
    assert np.array_equal(np.concatenate(masks), whole.masks())
    assert np.array_equal(np.concatenate(flows), whole.flow_info())
    assert np.array_equal(counters, whole.counters())
    assert np.array_equal(counters, orc.counters())


def test_replay_tool_counts_what_the_oracle_counts(tmp_path, capsys):
    """python -m xenoeye_amd.replay over a capture, an object directory and a
    list directory: the printed per-object numbers are the oracle's"""
    import json
    import os
    from xenoeye_amd import replay

    w = synth.workload_c2(30000, seed=5)
    path = str(tmp_path / "c.pcap")
    synth.batch_to_pcap(path, w["batch"], dport=2055)
    lists = tmp_path / "iplists"
    lists.mkdir()
    for name, text in w["iplists"].items():
        (lists / name).write_bytes(text if isinstance(text, bytes) else text.encode())
    root = tmp_path / "mo"
    orc = Oracle()
    for name, text in w["iplists"].items():
        orc.iplist_add(name, text)
    # directory order decides the object order: one object per directory, named so that it sorts
    for i, (filt, parent) in enumerate(w["mos"]):
        assert parent < 0
        orc.mo_add(filt, parent)
        d = root / ("o%03d" % i)
        d.mkdir(parents=True)
        (d / "mo.conf").write_text(json.dumps({"filter": filt, "fwm": [{"name": "p", "fields": ["proto", "packets"]}]}))
    run_oracle(orc, w["batch"])
    want = orc.counters()

    rc = replay.main(["--pcap", path, "--mo-dir", str(root), "--iplists", str(lists), "--port", "2055", "--rows", "2"])
    assert rc == 0
    out = json.loads(capsys.readouterr().out)
    assert out["taken"] == len(w["batch"].pkts) and out["records"] == orc.nrecords()
    got = {o["name"]: (o["records"], o["octets"], o["packets"]) for o in out["objects"]}
    assert len(got) == len(w["mos"])
    # objects are matched by filter text through the engine's own order
    eng = xenoeye_amd.Engine(device=0)
    eng.load_iplists(str(lists))
    eng.load_mos_with_tables(str(root))
    names = [eng.mo_name(i) for i in range(eng.mo_count())]
    dumps = [eng.mo_dump(i) for i in range(eng.mo_count())]
    eng.close()
    for i, (filt, _) in enumerate(w["mos"]):
        k = names.index("o%03d" % i)
        assert got[names[k]] == tuple(int(x) for x in want[i]), (filt, dumps[k])
    assert len(out["fixed_window"]) == len(w["mos"]) and all(t["dropped"] == 0 for t in out["fixed_window"])
    # rows by protocol number ascending ("proto", "packets"), sums as the counters have them
    for t in out["fixed_window"]:
        keys = [r["key"] for r in t["first_rows"]]
        assert keys == sorted(keys) and 1 <= len(keys) <= 2
    assert sum(sum(r["sums"]) for t in out["fixed_window"] for r in t["first_rows"]) > 0
