"""GPU parity of the fixed-window tables (the per-record part of
monit_object_process_nf(), monit-objects.c:923-1023): for every table the set
of rows {key bytes -> sums} equals the oracle's, whatever the order of
records, across batches and across window ends."""
import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import configure, run_oracle, compare

pytestmark = pytest.mark.gpu


@pytest.fixture()
def eng():
    e = xenoeye_amd.Engine(device=0)
    yield e
    e.close()


def setup(eng, orc, w, tables, max_keys=0):
    """tables: [(object index, [field strings])]"""
    for name, text in (w.get("iplists") or {}).items():
        eng.add_iplist(name, text)
        orc.iplist_add(name, text)
    for d in w.get("devices", ()):
        eng.add_device(**d)
        orc.device_add(**d)
    for filt, parent in w["mos"]:
        eng.add_mo("mo", filt, parent)
        orc.mo_add(filt, parent)
    ids = []
    for mo, fields in tables:
        g = eng.add_fwm(mo, fields, max_keys)
        o = orc.fwm_add(mo, fields)
        assert g == o
        ids.append(g)
    eng.commit()
    return ids


def same_tables(eng, orc, ids):
    for t in ids:
        got, dropped = eng.fwm_read(t)
        want = orc.fwm_read(t)
        assert dropped == 0
        assert len(got) == len(want), (t, len(got), len(want))
        assert got == want, t


def test_c2_tables(eng, orc):
    w = synth.workload_c2(30000, prefixes=300)
    w["mos"] = list(w["mos"]) + [("", -1), ("proto 6", 16)]
    tables = [(16, ["proto", "octets", "packets"]),                    # every record, tiny key
              (16, ["src host", "dst host", "bits desc", "packets"]),   # absent for IPv6 records: zeros
              (0, ["dst port", "proto", "octets"]),
              (8, ["src host6", "packets", "bits"]),
              (17, ["src port", "dst port", "tcp-flags", "src as", "octets desc"]),
              (3, ["packets"]),                                         # no key field: one row
              (5, ["dst host6", "src host", "dst ifidx", "src tos", "proto", "octets"])]
    ids = setup(eng, orc, w, tables)
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    compare(b, orc, flows=False)
    same_tables(eng, orc, ids)
    rows, _ = eng.fwm_read(ids[0])
    assert len(rows) == 3 and sum(v[1] for v in rows.values()) > 30000
    # a second batch adds to the same window
    w2 = synth.workload_c2(12000, prefixes=300, seed=99)
    b2 = eng.prepare(w2["batch"])
    b2.run()
    run_oracle(orc, w2["batch"], keep_flows=False)
    same_tables(eng, orc, ids)
    # window end
    eng.fwm_clear(ids[1])
    assert eng.fwm_read(ids[1])[0] == {}
    assert len(eng.fwm_read(ids[2])[0]) > 0


def test_devices_sampling_rate_scales_the_sums(eng, orc):
    w = synth.workload_c1(5000)
    w["devices"] = [dict(ip="192.0.2.1", sampling_rate=1000)]
    ids = setup(eng, orc, w, [(0, ["dst port", "octets", "bits", "packets"]), (0, ["dev-ip", "packets"])])
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    same_tables(eng, orc, ids)
    rows, _ = eng.fwm_read(ids[1])
    assert list(rows) == [bytes([192, 0, 2, 1])]
    assert rows[bytes([192, 0, 2, 1])][0] % 1000 == 0


def test_sflow_and_general_path_records(eng, orc):
    """sFlow samples (left-aligned fields, per-sample rate), cut-short v9
    records and mark-dropped ones: the tables see what the reference's
    consumer sees"""
    # record indices: Netflow records first, then sFlow ones; the batch is built in that order
    w = synth.workload_c1(3000, per_packet=24)
    w["mos"] = [("", -1), ("proto 6", 0)]
    c3 = synth.workload_c3(1500)
    for p, a, k in zip(c3["batch"].pkts, c3["batch"].src, c3["batch"].kind):
        w["batch"].add(p, a, k)
    w["devices"] = [dict(ip="192.0.2.1", sampling_rate=7, skip_unmarked=True, marks=("proto 17",))]
    ids = setup(eng, orc, w, [(0, ["src mac", "src vlan", "proto", "octets", "packets"]),
                              (1, ["src host", "tcp-flags", "bits"]),
                              (0, ["src host6", "dst host6", "packets"]),
                              (0, ["src ifidx", "dst ifidx", "octets"])])
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    compare(b, orc, flows=False)
    assert b.dropped().sum() > 0
    same_tables(eng, orc, ids)


@pytest.mark.parametrize("fields", [["src host", "src port", "packets"],                  # a key of one word
                                    ["src host", "dst host", "src port", "packets"]])     # ten bytes: two words
def test_table_that_is_too_small_counts_what_it_drops(eng, orc, fields):
    w = synth.workload_c1(20000)
    ids = setup(eng, orc, w, [(0, fields)], max_keys=256)
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    got, dropped = eng.fwm_read(ids[0])
    want = orc.fwm_read(ids[0])
    assert len(got) == 1024 and dropped > 0          # 256 keys asked for: 1024 slots
    # a table never frees a slot within a window: keys that got in early are complete, one that
    # got in while the table filled up may miss the part a block was still holding
    assert all(k in want and want[k][0] >= v[0] for k, v in got.items())
    assert sum(want[k] == v for k, v in got.items()) > 900
    assert len(want) > len(got)


def test_unsupported_fields_are_reported(eng):
    eng.add_mo("mo", "", -1)
    for f in (["mfreq(src port, dst port)", "octets"], ["sni", "packets"], ["country(src host)", "packets"],
              ["portstr(src port)", "packets"],      # no port names on this handle
              ["min(src mac, dst mac)", "packets"]):
        with pytest.raises(xenoeye_amd.XeError) as e:
            eng.add_fwm(0, f)
        assert e.value.code == -5
    with pytest.raises(xenoeye_amd.XeError) as e:
        eng.add_fwm(0, ["host", "packets"])
    assert e.value.code == -4


def test_tables_from_mo_conf(tmp_path, eng, orc):
    """the "fwm" entries of mo.conf files become tables (the shape of the
    reference's sample tree: total / proto / saddr ... per object)"""
    import os

    def mk(path, text):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "mo.conf"), "w") as f:
            f.write(text)
    root = str(tmp_path / "mo")
    mk(os.path.join(root, "ingress"), """{
        "filter": "dst net mynet",
        "debug": { "dump-flows": "none" },
        "fwm": [
            { "time": 15, "name": "total", "fields": ["packets", "octets"] },
            { "time": 15, "name": "proto", "fields": ["packets desc", "octets desc", "proto"] },
            { "time": 15, "name": "saddr", "fields": ["packets desc", "octets desc", "src host"], "limit": 5 },
            { "name": "sizes", "fields": ["div(octets, packets)", "packets"] },
            { "name": "strange", "fields": ["mfreq(src port, dst port)", "packets"] }   // stays with the reference
        ],
        "mavg": []
    }""")
    mk(os.path.join(root, "ingress", "tcp"), """{ "filter": "proto 6",
        "fwm": [ { "name": "ports", "fields": ["dst port", "bits desc"] } ] }""")
    mk(os.path.join(root, "plain"), '{ "filter": "proto 17" }')
    w = synth.workload_c1(6000)
    eng.add_iplist("mynet", w["iplists"]["mynet"])
    orc.iplist_add("mynet", w["iplists"]["mynet"])
    added, skipped = eng.load_mos_with_tables(root, 1 << 16)
    assert (added, skipped) == (5, 1)
    names = [eng.mo_name(i) for i in range(eng.mo_count())]
    tabs = eng.fwm_tables()
    assert sorted(n for _, n in tabs) == ["ports", "proto", "saddr", "sizes", "total"]
    eng.commit()
    # the same objects and tables for the oracle, in the library's object order
    filters = {"ingress": "dst net mynet", "ingress_tcp": "proto 6", "plain": "proto 17"}
    parents = {"ingress": -1, "ingress_tcp": names.index("ingress"), "plain": -1}
    for n in names:
        orc.mo_add(filters[n], parents[n])
    fields = {"total": ["packets", "octets"], "proto": ["packets desc", "octets desc", "proto"],
              "saddr": ["packets desc", "octets desc", "src host"], "ports": ["dst port", "bits desc"],
              "sizes": ["div(octets, packets)", "packets"]}
    for mo, n in tabs:
        orc.fwm_add(mo, fields[n])
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    compare(b, orc, flows=False)
    same_tables(eng, orc, list(range(len(tabs))))
    assert len(eng.fwm_read(0)[0]) >= 1
    # "limit" and "time" of the entries, and the export cut by the entry's own limit
    by_name = {n: t for t, (_, n) in enumerate(tabs)}
    assert eng.fwm_window(by_name["saddr"]) == (5, 15)
    assert eng.fwm_window(by_name["proto"]) == (0, 15)
    assert eng.fwm_window(by_name["ports"]) == (0, 0)
    for n, t in by_name.items():
        lim = eng.fwm_window(t)[0]
        assert eng.fwm_export(t) == orc.fwm_export(t, lim), n
    rows, others, hit = eng.fwm_export(by_name["saddr"])
    assert len(rows) == 5 and hit and others[0] > 0
    # most packets first
    assert [r[1][0] for r in rows] == sorted((r[1][0] for r in rows), reverse=True)


def test_export_order_and_others(eng, orc):
    """rows leave in the order of fwm_sort_and_dump() (monit-objects-fwm.c:496-589):
    by the fields as listed, `desc` inverted; the rows behind the limit are
    summed up (fwm_dump(), :405-449)"""
    w = synth.workload_c2(25000, prefixes=200, seed=4)
    w["mos"] = list(w["mos"]) + [("", -1)]
    tables = [(16, ["packets desc", "proto", "dst port"]),
              (16, ["proto desc", "src port", "octets", "packets desc"]),
              (16, ["dst port desc", "bits"]),
              (16, ["src host", "dst host desc", "packets"]),
              (16, ["octets", "packets desc"]),                 # no key field
              (0, ["dst host6 desc", "src as", "octets desc", "bits"]),
              (16, ["proto", "tcp-flags desc"])]                # no sums
    ids = setup(eng, orc, w, tables, max_keys=1 << 17)
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    same_tables(eng, orc, ids)
    for t in ids:
        total = len(eng.fwm_read(t)[0])
        for limit in (0, 1, 5, total - 1, total, total + 1, 1000000):
            if limit < 0:
                continue
            got, want = eng.fwm_export(t, limit), orc.fwm_export(t, limit)
            assert got == want, (t, limit)
            rows, others, hit = got
            assert len(rows) == (min(limit, total) if limit else total)
            assert hit == (limit != 0 and total >= limit)
    # the first table: descending packets, ties by proto then port ascending
    rows, _, _ = eng.fwm_export(ids[0], 0)
    order = [(-r[1][0], r[0]) for r in rows]
    assert order == sorted(order)
    # an empty table exports nothing
    eng.fwm_clear(ids[2])
    assert eng.fwm_export(ids[2], 3) == ([], (0,), False)


def test_many_blocks_insert_the_same_new_keys(eng, orc):
    """a few hundred distinct keys, every one of them met by every block of the table kernel at the
    same moment in the first batch of a window, into the smallest table: each key must end up in
    exactly one row (Engine.fwm_read refuses a table that holds a key twice), with the oracle's sums"""
    rng = synth.Rng(0xF3)
    n = 400000
    fields = [(8, 4), (12, 4), (7, 2), (11, 2), (4, 1), (1, 8), (2, 8)]
    recs = synth.flow_columns(rng, n, fields)
    recs[:, 8:10] = synth._be(rng.below(n, 300), 2)           # 300 source ports
    recs[:, 10:12] = synth._be(rng.below(n, 3) + 1, 2)        # 3 destination ports: 900 keys in the second table
    recs[:, 12] = 6
    b = synth.Batch()
    exp = bytes([192, 0, 2, 1])
    b.add(synth.v9_template_packet(256, fields, 77), exp)
    b.extend(synth.v9_data_packets(256, recs, 48, 77), exp)
    w = dict(batch=b, mos=[("", -1)])
    ids = setup(eng, orc, w, [(0, ["src port", "octets", "packets"]), (0, ["src port", "dst port", "bits"]),
                              (0, ["proto", "packets"])], max_keys=400)
    out = eng.prepare(b)
    for window in range(3):
        out.run()
        if window == 0:
            run_oracle(orc, b, keep_flows=False)
        got, dropped = eng.fwm_read(ids[0])
        assert len(got) == 300 and dropped == 0
        got1, dropped1 = eng.fwm_read(ids[1])
        # 900 keys into a table made for 400 (1024 slots): all of them still fit
        assert len(got1) == 900 and dropped1 == 0
        assert got == orc.fwm_read(ids[0]) and got1 == orc.fwm_read(ids[1])
        assert eng.fwm_read(ids[2])[0] == orc.fwm_read(ids[2])
        for t in ids:
            eng.fwm_clear(t)


def test_long_keys_that_share_their_first_words(eng, orc):
    """keys of several words are claimed word by word (seven key bytes to a word): keys that agree
    in their first words and differ in a later one, keys full of zero bytes and the all-zero key,
    met by every block at once in the first batch of a window, in a table with long probe chains -
    every key in exactly one row, with the oracle's sums"""
    rng = synth.Rng(0x7A)
    n = 300000
    fields = [(8, 4), (12, 4), (7, 2), (11, 2), (4, 1), (1, 8), (2, 8), (27, 16), (28, 16)]
    recs = synth.flow_columns(rng, n, fields)
    recs[:, 0:4] = np.frombuffer(bytes([10, 0, 0, 0]), dtype=np.uint8)      # one source ...
    recs[:, 3] = rng.below(n, 2).astype(np.uint8)                            # ... or its neighbour
    recs[:, 4:7] = 0                                                         # dst host 0.0.0.x: word 0 of few values
    recs[:, 7] = rng.below(n, 40).astype(np.uint8)
    recs[:, 8:10] = synth._be(rng.below(n, 30), 2)                           # ports from 0 on: zero bytes
    recs[:, 10:12] = synth._be(rng.below(n, 3), 2)
    recs[:, 12] = 6
    recs[:, 29:45] = 0                                                       # IPv6 addresses: mostly zero bytes
    recs[:, 44] = rng.below(n, 5).astype(np.uint8)
    recs[:, 45:61] = 0
    recs[:, 52] = rng.below(n, 7).astype(np.uint8)
    for i in range(0, n, 11):                                                # the all-zero keys
        recs[i, 0:13] = 0
        recs[i, 29:61] = 0
    b = synth.Batch()
    exp = bytes([192, 0, 2, 1])
    b.add(synth.v9_template_packet(256, fields, 77), exp)
    b.extend(synth.v9_data_packets(256, recs, 20, 77), exp)
    w = dict(batch=b, mos=[("", -1)])
    ids = setup(eng, orc, w, [(0, ["src host", "dst host", "src port", "dst port", "packets"]),
                              (0, ["src host6", "dst host6", "src port", "proto", "octets", "packets", "bits"]),
                              (0, ["dst host", "src host6", "packets"])], max_keys=4000)
    out = eng.prepare(b)
    for window in range(3):
        out.run()
        if window == 0:
            run_oracle(orc, b, keep_flows=False)
        same_tables(eng, orc, ids)
        got, _ = eng.fwm_read(ids[0])
        assert bytes(12) in got and len(got) > 5000
        assert bytes(35) in eng.fwm_read(ids[1])[0]
        for t in ids:
            eng.fwm_clear(t)


def test_the_key_a_one_word_table_cannot_store_among_the_others(eng, orc):
    """tables whose key is one word keep it xor-ed with a constant, 0 meaning a free slot: the
    one key that would be stored as 0 gets a row of its own, with the oracle's sums like any other"""
    rng = synth.Rng(0x51)
    n = 5000
    fields = [(8, 4), (12, 4), (7, 2), (11, 2), (4, 1), (1, 8), (2, 8)]
    recs = synth.flow_columns(rng, n, fields)
    special = bytes.fromhex("157c4a7fb979379e")       # the constant's bytes in key order: src host, dst host
    for i in range(0, n, 7):
        recs[i, 0:8] = np.frombuffer(special, dtype=np.uint8)
    for i in range(1, n, 7):                          # neighbours that differ in one byte
        recs[i, 0:8] = np.frombuffer(special, dtype=np.uint8)
        recs[i, 7] ^= 1
    b = synth.Batch()
    exp = bytes([192, 0, 2, 1])
    b.add(synth.v9_template_packet(256, fields, 77), exp)
    b.extend(synth.v9_data_packets(256, recs, 48, 77), exp)
    w = dict(batch=b, mos=[("", -1)])
    ids = setup(eng, orc, w, [(0, ["src host", "dst host", "packets", "octets"]), (0, ["packets"])])
    out = eng.prepare(b)
    for window in range(2):
        out.run()
        if window == 0:
            run_oracle(orc, b, keep_flows=False)
        same_tables(eng, orc, ids)
        got, _ = eng.fwm_read(ids[0])
        assert special in got and len(got) > 1400
        for t in ids:
            eng.fwm_clear(t)


def test_function_fields_as_keys(eng, orc):
    """div / div_r / div_l, min, tfstr, portstr, ppstr as key fields
    (monit_object_key_add_fld(), monit-objects.c:878-921): the rows, with the
    texts the reference writes into the key, equal the oracle's"""
    from oracle_binding import port_names
    names = port_names()
    eng.set_port_names(names)
    orc.set_port_names(names)
    w = synth.workload_c2(30000, prefixes=300)
    w["mos"] = list(w["mos"]) + [("", -1), ("proto 6", 16)]
    tables = [(16, ["div(octets, packets)", "packets"]),
              (16, ["div_r(octets, packets, 100) desc", "proto", "octets"]),
              (16, ["div_l(bits, packets, 2)", "octets", "packets"]),
              (16, ["div_l(octets, packets, 10)", "div(packets, octets)", "packets"]),
              (16, ["min(src port, dst port)", "bits"]),
              (17, ["tfstr(tcp-flags)", "packets desc"]),
              (16, ["proto", "portstr(dst port)", "octets"]),
              (16, ["ppstr(src port, dst port)", "packets"]),
              (0, ["src host", "min(src as, dst as)", "tfstr(tcp-flags)", "portstr(src port)", "octets"])]
    ids = setup(eng, orc, w, tables)
    for seed in (None, 77):
        wb = w if seed is None else synth.workload_c2(9000, prefixes=300, seed=seed)
        b = eng.prepare(wb["batch"])
        b.run()
        run_oracle(orc, wb["batch"], keep_flows=False)
        same_tables(eng, orc, ids)
        b.free()
    rows, _ = eng.fwm_read(ids[5])
    assert any(k.rstrip(b"\0") in (b"ACK", b"ACK+SYN", b"SYN", b"ACK+PSH") for k in rows) and all(len(k) == 32 for k in rows)
    rows, _ = eng.fwm_read(ids[6])
    assert any(b" (" in k[1:] for k in rows) and all(len(k) == 41 for k in rows)
    # the export sorts by the texts and sums as the reference does
    for t in (ids[1], ids[5], ids[7]):
        assert eng.fwm_export(t, 5) == orc.fwm_export(t, 5), t


def test_function_fields_on_sflow_and_absent_fields(eng, orc):
    w = synth.workload_c3(4000)
    w["mos"] = [("", -1)]
    ids = setup(eng, orc, w, [(0, ["div(octets, packets)", "tfstr(tcp-flags)", "packets"]),
                              (0, ["min(src port, dst port)", "div_r(bits, packets, 64)", "octets"]),
                              (0, ["min(src as, dst as)", "packets"])])      # sFlow sets no AS: zeros
    b = eng.prepare(w["batch"])
    b.run()
    run_oracle(orc, w["batch"], keep_flows=False)
    same_tables(eng, orc, ids)
