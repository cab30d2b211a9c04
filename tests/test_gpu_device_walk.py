"""Header walk on the device: batches whose templates the handle already knows
are listed by the GPU itself (sFlow always).  Results must equal the oracle's
and those of the host walk, whatever mixture of usual and unusual datagrams a
batch holds, and whatever happens to the templates inside it."""
import struct

import numpy as np
import pytest

import xenoeye_amd
from xenoeye_amd import synth
from helpers import compare, configure, run_oracle

pytestmark = pytest.mark.gpu


def _engines(orc, w, **kw):
    dev = xenoeye_amd.Engine(device=0)
    host = xenoeye_amd.Engine(device=0)
    host.set_device_walk(0)
    configure(dev, orc, w["iplists"], w["mos"], **kw)
    for name, text in w["iplists"].items():
        host.add_iplist(name, text)
    for d in kw.get("devices", ()):
        host.add_device(**d)
    for filt, parent in w["mos"]:
        host.add_mo("mo", filt, parent)
    host.commit()
    return dev, host


def _is_template(p):
    ver = struct.unpack(">H", p[:2])[0]
    if ver == 9:
        return struct.unpack(">H", p[20:22])[0] == 0
    if ver == 10:
        return struct.unpack(">H", p[16:18])[0] == 2
    return False


def _same(a, b):
    assert a.records == b.records
    assert np.array_equal(a.masks(), b.masks())
    assert np.array_equal(a.counters(), b.counters())
    assert np.array_equal(a.dropped(), b.dropped())
    assert np.array_equal(a.record_datagram(), b.record_datagram())
    if a.records:
        assert np.array_equal(a.flow_info(), b.flow_info())


def _round(dev, host, orc, batch, flows=True):
    orc.log_clear()
    a = dev.prepare(batch)
    a.run()
    b = host.prepare(batch)
    b.run()
    # the batch numbers its Netflow records before its sFlow records: the oracle
    # sees the datagrams in that order (neither kind keeps state the other reads)
    ordered = synth.Batch()
    for kind in (0, 1):
        for p, s, k in zip(batch.pkts, batch.src, batch.kind):
            if k == kind:
                ordered.add(p, s, k)
    run_oracle(orc, ordered, keep_flows=flows)
    compare(a, orc, flows=flows)
    _same(a, b)
    walked = a.device_walked()
    a.free()
    b.free()
    return walked


@pytest.mark.parametrize("name", ["c1", "c2", "c5"])
def test_known_templates_are_walked_on_the_device(orc, name):
    w = {"c1": lambda: synth.workload_c1(20000),
         "c2": lambda: synth.workload_c2(30000, prefixes=400),
         "c5": lambda: synth.workload_c5(40000, prefixes=300, n_exporters=64)}[name]()
    dev, host = _engines(orc, w)
    first = _round(dev, host, orc, w["batch"])
    # the first batch brings the templates: its exporters are walked by the host
    assert first[0] == 0
    second = _round(dev, host, orc, w["batch"])
    # the templates come again, unchanged: nothing is left to the host
    assert second[0] == len(w["batch"].pkts), (second, len(w["batch"].pkts))
    assert second[1] == w["nrec"]
    dev.close()
    host.close()


def test_sflow_is_walked_on_the_device(orc):
    w = synth.workload_c3(20000)
    dev, host = _engines(orc, w)
    got = _round(dev, host, orc, w["batch"])
    assert got[2] >= 20000
    dev.close()
    host.close()


def test_sflow_damaged_datagrams(orc):
    """cut, stretched and corrupted sFlow datagrams: the same records, or none, on both walks"""
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
    dev, host = _engines(orc, w)
    _round(dev, host, orc, b)
    dev.close()
    host.close()


def test_mixed_kinds_and_unusual_datagrams(orc):
    """Netflow v5 / v9 / IPFIX and sFlow interleaved; among the Netflow ones
    template refreshes, two sets in a datagram, a count that exceeds the
    records, trailing bytes, an unknown template: those go to the host walk,
    the rest to the device, and every record keeps its place"""
    w1 = synth.workload_c1(3000)
    w2 = synth.workload_c2(4000, prefixes=200)
    w3 = synth.workload_c3(1500)
    rng = synth.Rng(99)
    exp1 = bytes([192, 0, 2, 1])
    mos = w1["mos"] + w2["mos"] + [("proto 6", -1), ("dst port 0-1023", 0)]
    lists = dict(w1["iplists"])
    lists.update(w2["iplists"])
    w = dict(iplists=lists, mos=mos)
    dev, host = _engines(orc, w)

    first = synth.Batch()
    for p, s in zip(w1["batch"].pkts[:3], w1["batch"].src[:3]):
        first.add(p, s)
    for p, s in zip(w2["batch"].pkts[:200], w2["batch"].src[:200]):
        first.add(p, s)
    first.add(synth.v5_packet(rng, 5), bytes([10, 9, 9, 9]))
    _round(dev, host, orc, first)

    second = synth.Batch()
    streams = [list(zip(w1["batch"].pkts[1:], w1["batch"].src[1:], [0] * 10000)),
               list(zip(w2["batch"].pkts[64:], w2["batch"].src[64:], [0] * 10000)),
               list(zip(w3["batch"].pkts, w3["batch"].src, [1] * 10000))]
    odd = []
    data = w1["batch"].pkts[5]
    # template refresh (unchanged), then data
    odd.append((w1["batch"].pkts[0], exp1, 0))
    # a count larger than the records of the set
    p = bytearray(data); p[2:4] = struct.pack(">H", 70); odd.append((bytes(p), exp1, 0))
    # a count smaller than the records
    p = bytearray(data); p[2:4] = struct.pack(">H", 10); odd.append((bytes(p), exp1, 0))
    # trailing bytes behind the set
    odd.append((data + b"\xc0\xa8\x01\x01", exp1, 0))
    # two data sets in one datagram
    odd.append((data + data[20:], exp1, 0))
    # a template nobody sent
    p = bytearray(data); p[20:22] = struct.pack(">H", 999); odd.append((bytes(p), exp1, 0))
    # cut short
    odd.append((data[:500], exp1, 0))
    odd.append((data[:22], exp1, 0))
    # netflow v5, and a v5 datagram with a wrong length
    v5 = synth.v5_packet(rng, 7)
    odd.append((v5, bytes([10, 9, 9, 9]), 0))
    odd.append((v5[:-3], bytes([10, 9, 9, 9]), 0))
    odd.append((b"\x00\x07" + data[2:], exp1, 0))
    j = 0
    while any(streams):
        for st in streams:
            for _ in range(3):
                if st:
                    p, s, k = st.pop(0)
                    second.add(p, s, kind=k)
        if j < len(odd):
            p, s, k = odd[j]
            second.add(p, s, kind=k)
            j += 1
    got = _round(dev, host, orc, second)
    assert got[0] > 100 and got[2] >= 1500
    got2 = _round(dev, host, orc, second)
    assert got2 == got
    dev.close()
    host.close()


def _ipfix(domain, *sets):
    body = b"".join(sets)
    return synth.ipfix_header(16 + len(body), domain) + body


def _set(set_id, body, length=None):
    return struct.pack(">HH", set_id, (4 + len(body)) if length is None else length) + body


def test_ipfix_datagrams_of_several_sets(orc):
    """IPFIX datagrams with up to four data sets, options template sets, sets
    of ids that mean nothing and repeats of known templates between them are
    listed by the device; more sets than that, unknown templates, set lengths
    that leave the datagram or stop short of the set header go to the host walk"""
    w = synth.workload_c2(2000, prefixes=200)
    w["mos"] = w["mos"] + [("proto 6", -1), ("src port 53 or dst port 53", -1)]
    dev, host = _engines(orc, w)
    rng = synth.Rng(4242)
    exps = [bytes([10, 255, 0, e + 1]) for e in range(4)]

    first = synth.Batch()
    for e in range(4):
        first.add(synth.ipfix_template_packet(256, synth.TMPL_IPFIX_V4, e + 1), exps[e])
        first.add(synth.ipfix_template_packet(257, synth.TMPL_IPFIX_V6, e + 1), exps[e])
    _round(dev, host, orc, first)

    def d4(n):
        return _set(256, synth.flow_columns(rng, n, synth.TMPL_IPFIX_V4).tobytes())

    def d6(n):
        return _set(257, synth.flow_columns(rng, n, synth.TMPL_IPFIX_V6).tobytes())

    options = _set(3, struct.pack(">HHH", 300, 2, 1) + struct.pack(">HH", 149, 4) + struct.pack(">HH", 160, 8))
    noise = _set(17, bytes(rng.bytes(12)))
    usual, unusual = [], []
    for j in range(300):
        e = j % 4
        k = j % 6
        if k == 0:
            usual.append((_ipfix(e + 1, d4(7), d6(5)), e))
        elif k == 1:
            usual.append((_ipfix(e + 1, options, d4(9)), e))
        elif k == 2:
            usual.append((_ipfix(e + 1, d6(3), noise, d4(4), d6(6)), e))
        elif k == 3:
            usual.append((_ipfix(e + 1, d4(1), d4(2), d4(3), d6(4)), e))
        elif k == 4:
            usual.append((_ipfix(e + 1, d4(11)), e))
        else:
            usual.append((_ipfix(e + 1, noise, options, d6(2), noise), e))
    # five data sets
    unusual.append((_ipfix(1, d4(1), d4(2), d6(3), d4(4), d6(5)), 0))
    # a template nobody sent behind a good set: the rest of the datagram is dropped
    unusual.append((_ipfix(2, d4(3), _set(999, bytes(rng.bytes(46))), d4(2)), 1))
    # an (unchanged) template between two data sets
    usual.append((_ipfix(3, d4(3), synth.ipfix_template_set(257, synth.TMPL_IPFIX_V6), d6(2)), 2))
    # a set length that leaves the datagram
    unusual.append((_ipfix(4, d4(2), _set(257, synth.flow_columns(rng, 2, synth.TMPL_IPFIX_V6).tobytes(), length=400)), 3))
    # set lengths 0 and 3
    unusual.append((_ipfix(1, d4(2), _set(256, b"", length=0), d4(1)), 0))
    unusual.append((_ipfix(2, d6(2), _set(17, b"", length=3) + bytes(8), d4(1)), 1))
    # a set header cut by the end of the datagram
    unusual.append((_ipfix(3, d4(2)) + b"\x01\x00", 2))
    # only sets that carry no records
    usual.append((_ipfix(4, options, noise), 3))
    # a data set cut inside a record
    p = _ipfix(1, d4(2), d6(3))
    unusual.append((p[:-20], 0))

    second = synth.Batch()
    nusual = 0
    for j, (p, e) in enumerate(usual):
        second.add(p, exps[e])
        nusual += 1
        if (j % 30 == 7) and unusual:
            q, e2 = unusual.pop(0)
            second.add(q, exps[e2])
    for q, e2 in unusual:
        second.add(q, exps[e2])
    got = _round(dev, host, orc, second)
    assert got[0] == nusual, (got, nusual)
    dev.close()
    host.close()


def test_template_refreshes_stay_on_the_device(orc):
    """exporters that send their templates again every 20 datagrams - in
    datagrams of their own, in front of the data set of a v9 datagram, or as a
    set of an IPFIX datagram: as long as the bytes repeat what the handle holds,
    the device lists everything; a refresh that differs in one byte takes its
    exporter to the host walk for that batch"""
    rng = synth.Rng(77)
    e9, e10 = bytes([192, 0, 2, 9]), bytes([192, 0, 2, 10])
    w = synth.workload_c2(64, prefixes=100)
    w["mos"] = w["mos"] + [("proto 6", -1), ("dst port 0-1023", -1)]
    dev, host = _engines(orc, w)
    t9 = synth.v9_template_flowset(300, synth.TMPL_V9_C1)
    first = synth.Batch()
    first.add(synth.v9_template_packet(300, synth.TMPL_V9_C1, 5), e9)
    first.add(synth.ipfix_template_packet(256, synth.TMPL_IPFIX_V4, 6), e10)
    first.add(synth.ipfix_template_packet(257, synth.TMPL_IPFIX_V6, 6), e10)
    _round(dev, host, orc, first)

    def v9_data(n):
        rec = synth.flow_columns(rng, n, synth.TMPL_V9_C1)
        return struct.pack(">HH", 300, 4 + rec.size) + rec.tobytes()

    def d4(n):
        return _set(256, synth.flow_columns(rng, n, synth.TMPL_IPFIX_V4).tobytes())

    def d6(n):
        return _set(257, synth.flow_columns(rng, n, synth.TMPL_IPFIX_V6).tobytes())

    second = synth.Batch()
    for j in range(400):
        if j % 20 == 0:
            # the template again, in a datagram of its own ...
            second.add(synth.v9_template_packet(300, synth.TMPL_V9_C1, 5), e9)
            second.add(synth.ipfix_template_packet(257, synth.TMPL_IPFIX_V6, 6), e10)
        if j % 20 == 10:
            # ... and in front of the data
            second.add(synth.v9_header(12, 5) + t9 + v9_data(12), e9)
            second.add(_ipfix(6, synth.ipfix_template_set(256, synth.TMPL_IPFIX_V4), d4(5), d6(2)), e10)
        second.add(synth.v9_header(30, 5) + v9_data(30), e9)
        second.add(_ipfix(6, d4(9)) if j % 3 else _ipfix(6, d6(4), d4(2)), e10)
    got = _round(dev, host, orc, second)
    assert got[0] == len(second.pkts), (got, len(second.pkts))

    # one byte of the repeated template differs: a changed template, the host's business
    third = synth.Batch()
    for p, s in list(zip(second.pkts, second.src))[:100]:
        third.add(p, s)
    changed = bytearray(synth.v9_header(12, 5) + t9 + v9_data(12))
    changed[20 + 4 + 4 + 3] ^= 1      # the length of the first element
    third.add(bytes(changed), e9)
    for p, s in list(zip(second.pkts, second.src))[100:160]:
        third.add(p, s)
    got = _round(dev, host, orc, third)
    n10 = sum(1 for s in third.src if s == e10)
    assert got[0] == n10, (got, n10)
    dev.close()
    host.close()


def test_template_change_inside_a_batch(orc):
    """an exporter replaces its template half way through a batch whose other
    datagrams the device has already listed with the old one: that exporter is
    walked by the host from its first datagram on, the others stay"""
    fields_a = synth.TMPL_V9_C1
    fields_b = [(12, 4), (8, 4), (11, 2), (7, 2), (4, 1), (1, 4), (2, 4), (10, 4)]
    rng = synth.Rng(5)
    e1, e2 = bytes([192, 0, 2, 1]), bytes([192, 0, 2, 2])
    w = synth.workload_c1(64)
    w["mos"] = [("src net mynet", -1), ("dst net mynet", -1), ("proto 17", -1)]
    dev, host = _engines(orc, w)
    rec_a = synth.flow_columns(rng, 64 * 40, fields_a)
    rec_b = synth.flow_columns(rng, 50 * 20, fields_b)
    pk_a1 = synth.v9_data_packets(256, rec_a, 64, 77)
    pk_a2 = synth.v9_data_packets(256, rec_a, 64, 78)
    pk_b1 = synth.v9_data_packets(256, rec_b, 50, 77)

    first = synth.Batch()
    first.add(synth.v9_template_packet(256, fields_a, 77), e1)
    first.add(synth.v9_template_packet(256, fields_a, 78), e2)
    first.extend(pk_a1[:4], e1)
    first.extend(pk_a2[:4], e2)
    _round(dev, host, orc, first)

    second = synth.Batch()
    for i in range(10):
        second.add(pk_a1[4 + i], e1)
        second.add(pk_a2[4 + i], e2)
    second.add(synth.v9_template_packet(256, fields_b, 77), e1)
    for i in range(10):
        second.add(pk_b1[i], e1)
        second.add(pk_a2[14 + i], e2)
    # the old layout again, now misread through the new template
    second.add(pk_a1[20], e1)
    got = _round(dev, host, orc, second)
    assert got[0] == 20          # exporter 2 only
    third = synth.Batch()
    for i in range(5):
        third.add(pk_b1[10 + i], e1)
        third.add(pk_a2[24 + i], e2)
    got = _round(dev, host, orc, third)
    assert got[0] == 10
    dev.close()
    host.close()


def test_devices_conf_exporters(orc):
    """exporters with a devices.conf entry stay on the device walk, with or without mark filters:
    the runs of an entry with filters name it, and the decode kernel runs them"""
    w = synth.workload_c5(6000, prefixes=100, n_exporters=8)
    devices = [dict(ip="10.5.0.1", sampling_rate=100),
               dict(ip="10.5.0.2", dev_id=2, sampling_rate=10),
               dict(ip="10.5.0.6", sampling_rate=7, marks=["proto 6", "dst port 0-1023"]),
               dict(dev_id=7, skip_unmarked=True, marks=["proto 17"])]
    dev, host = _engines(orc, w, devices=devices)
    _round(dev, host, orc, w["batch"])
    got = _round(dev, host, orc, w["batch"])
    assert got[0] == len(w["batch"].pkts)
    dev.close()
    host.close()


def test_many_devices_conf_entries_first_match_wins(orc):
    """from nine entries on the device walk finds an exporter's entry through a table instead of a
    scan: the answer must stay the FIRST entry that matches, including the entries with address and
    id whose id alone matches in the second test (devices.c:224 is `if`, not `else if`)"""
    w = synth.workload_c5(8000, prefixes=100, n_exporters=8)
    ids = sorted({int.from_bytes(p[16:20] if p[1] == 9 else p[12:16], "big") for p in w["batch"].pkts})
    devices = [dict(ip="10.9.9.9", sampling_rate=11),
               dict(ip="10.9.9.8", dev_id=ids[0], sampling_rate=12),          # wrong address: matches by id alone
               dict(ip="10.5.0.1", sampling_rate=13),                         # never reached by exporters of id ids[0]
               dict(ip="10.5.0.2", dev_id=4242, sampling_rate=14),            # neither
               dict(ip="10.5.0.2", sampling_rate=15, marks=["proto 6"]),
               dict(dev_id=ids[-1], sampling_rate=16, skip_unmarked=True, marks=["proto 17", "dst port 0-1023"]),
               dict(ip="10.5.0.3", dev_id=ids[-1], sampling_rate=17),         # shadowed by the entry before
               dict(ip="10.5.0.4", sampling_rate=18),
               dict(ip="10.5.0.4", sampling_rate=19),                         # shadowed
               dict(ip="10.5.0.5", dev_id=ids[len(ids) // 2], sampling_rate=20),
               dict(dev_id=ids[len(ids) // 2], sampling_rate=21),
               dict(ip="10.5.0.6", sampling_rate=22, marks=["proto 6"]),
               dict(ip="10.5.0.7", sampling_rate=23)]
    w["mos"] = list(w["mos"]) + [("dev-mark 1", -1), ("dev-mark 0", -1)]
    dev, host = _engines(orc, w, devices=devices)
    _round(dev, host, orc, w["batch"])
    got = _round(dev, host, orc, w["batch"])
    assert got[0] == len(w["batch"].pkts)
    dev.close()
    host.close()


def test_batch_beyond_the_fast_kernels_reach(orc, monkeypatch):
    """from 2^28 records on a batch goes to the per-lane kernel, and the device walk hands such a
    batch back to the host walk with the template store as it was (here the limit is lowered so
    that a small batch takes that road; a template change inside the batch must survive it)"""
    fields_a = synth.TMPL_V9_C1
    fields_b = [(12, 4), (8, 4), (11, 2), (7, 2), (4, 1), (1, 4), (2, 4), (10, 4)]
    rng = synth.Rng(6)
    e1 = bytes([192, 0, 2, 1])
    w = synth.workload_c1(64)
    dev, host = _engines(orc, w)
    rec_a = synth.flow_columns(rng, 64 * 30, fields_a)
    rec_b = synth.flow_columns(rng, 50 * 10, fields_b)
    pk_a = synth.v9_data_packets(256, rec_a, 64, 77)
    pk_b = synth.v9_data_packets(256, rec_b, 50, 77)
    first = synth.Batch()
    first.add(synth.v9_template_packet(256, fields_a, 77), e1)
    first.extend(pk_a[:4], e1)
    assert _round(dev, host, orc, first)[0] == 0
    monkeypatch.setenv("XE_FAST_MAX_RECORDS", "1000")
    second = synth.Batch()
    second.extend(pk_a[4:20], e1)
    second.add(synth.v9_template_packet(257, fields_b, 77), e1)      # a new template: logged, rolled back, learnt again
    for p in pk_b:
        second.add(p[:20] + struct.pack(">H", 257) + p[22:], e1)
    second.extend(pk_a[20:], e1)
    got = _round(dev, host, orc, second)
    assert got[0] == 0            # nothing was listed on the device in the end
    monkeypatch.delenv("XE_FAST_MAX_RECORDS")
    got = _round(dev, host, orc, second)
    assert got[0] == len(second.pkts)     # (the template in its middle is known by now)
    dev.close()
    host.close()


def test_offsets_that_decrease_are_refused():
    w = synth.workload_c1(640)
    eng = xenoeye_amd.Engine(device=0)
    for name, text in w["iplists"].items():
        eng.add_iplist(name, text)
    eng.add_mo("mo", "src net mynet", -1)
    eng.commit()
    data, offs, src, kinds = w["batch"].packed()
    bad = offs.copy()
    bad[3], bad[4] = bad[4], bad[3]
    with pytest.raises(xenoeye_amd.XeError):
        eng.prepare((data, bad, src, kinds))
    b = eng.prepare((data, offs, src, kinds))
    b.run()
    assert b.records == 640
    eng.close()
